#!/bin/bash
# (MRMT3_DDP_GRAPH_STREAM_PRIO was renamed MRMT3_DDP_STREAM_PRIO after this session)
# round 5, session 6: the two-graph form with the collective stream picked by the side-by-side test, normal vs high priority
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_train_graph_gpu.py -m gpu -x -q -k "flag or collectives" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/s6_pytest.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f' % d['ms_per_step'], 'graphs/step', d['graph_segments'], '|', d['collectives'], '| captured:', d.get('collectives_captured'))"; }
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for rep in 1 2; do
  timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s6_plain_$rep.json | show "plain (no collectives)                     "
  MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s6_forced_torch_$rep.json | show "segments + torch.distributed               "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_NATIVE=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s6_forced_native_$rep.json | show "segments + mrmt3_allreduce                 "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s6_forced_twograph_normal_$rep.json | show "two graphs, collective stream normal (pick)"
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=1 MRMT3_DDP_GRAPH_STREAM_PRIO=-1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s6_forced_twograph_high_$rep.json | show "two graphs, collective stream HIGH         "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=inline timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s6_forced_inline_$rep.json | show "one graph, in-line                         "
done 2>&1 | tee $O/s6_collectives_ab.log
