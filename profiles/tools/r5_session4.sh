#!/bin/bash
# round 5, session 4: the captured-collectives form with the collective stream at high priority (+ side-by-side check), A/B of the
# four exchange forms at world 1, and the fused wi + GEGLU kernel without its spills
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_train_graph_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/s4_pytest.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f' % d['ms_per_step'], 'graphs/step', d['graph_segments'], '|', d['collectives'], '| captured:', d.get('collectives_captured'))"; }
for rep in 1 2 3; do
  timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s4_plain_$rep.json | show "plain (no collectives)      "
  MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s4_forced_torch_$rep.json | show "segments + torch.distributed"
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_NATIVE=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s4_forced_native_$rep.json | show "segments + mrmt3_allreduce  "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=1 timeout 300 python3 bench.py $B 2>$O/s4_twograph_$rep.err | tee $O/s4_forced_twograph_$rep.json | show "two graphs (captured)       "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=inline timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s4_forced_inline_$rep.json | show "one graph, in-line          "
  MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libmrmt3_r4_107.so timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s4_plain_r4lib_$rep.json | show "plain, round-4 library      "
done 2>&1 | tee $O/s4_collectives_ab.log
tail -5 $O/s4_twograph_1.err
