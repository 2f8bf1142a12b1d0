#!/bin/bash
# round 5, session 9: MR-MT3's 320-key cross-attention backward as a split site (one-pass on 256 keys + two-pass tail): parity, step A/B
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/s9_pytest_attn.log
timeout 1500 python3 -m pytest tests/test_bench_shape_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "segmem_v2_with_prev or long_context" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/s9_pytest_model.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0 --variant segmem_v2_with_prev"
for rep in 1 2 3; do
  timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MR-MT3 64 segments, split site (default) ', '%.3f' % d['ms_per_step'])"
  MRMT3_ATTN_ONEPASS_SPLIT=0 timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MR-MT3 64 segments, two-pass on 320 keys  ', '%.3f' % d['ms_per_step'])"
done 2>&1 | tee $O/s9_split_ab.log
timeout 300 python3 bench.py $B --batch 24 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MR-MT3 24 segments, split ', '%.3f' % d['ms_per_step'])" | tee -a $O/s9_split_ab.log
MRMT3_ATTN_ONEPASS_SPLIT=0 timeout 300 python3 bench.py $B --batch 24 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('MR-MT3 24 segments, two-pass', '%.3f' % d['ms_per_step'])" | tee -a $O/s9_split_ab.log
