#!/bin/bash
# round 5, session 7: one-pass attention backward with the two waves of a SIMD taking phase A / tail in opposite order (-DOP_STAGGER,
# VERDICT r4 item 6a in its cheapest form): parity under the variant library, micro and step A/B; + the collective-stream test
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ST=$PWD/profiles/tools/_ab/libmrmt3_stagger.so
MRMT3_TOOL_LIB=$ST timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "onepass or attn_bwd" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/s7_pytest_stagger.log
timeout 600 python3 -m pytest tests/test_train_graph_gpu.py -m gpu -x -q -k "bucketed or collectives" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/s7_pytest.log
for rep in 1 2; do
  echo "--- product"; timeout 300 python3 profiles/tools/attn_micro.py 20 2>&1 | grep -v amdgpu.ids | grep -i "cross\|enc" 
  echo "--- stagger"; MRMT3_TOOL_LIB=$ST timeout 300 python3 profiles/tools/attn_micro.py 20 2>&1 | grep -v amdgpu.ids | grep -i "cross\|enc"
done 2>&1 | tee $O/s7_attn_micro.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0"
for rep in 1 2 3; do
  timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('product ', '%.3f' % d['ms_per_step'])"
  MRMT3_TOOL_LIB=$ST timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger ', '%.3f' % d['ms_per_step'])"
done 2>&1 | tee $O/s7_step_ab.log
