#!/bin/bash
# round 5, session 12: the host-gated exchange (one compute graph, the host launches each bucket's collective when the graph publishes it):
# parity tests, forced-collective A/B at world 1 (real RCCL), emulated N-GPU cost against the segmented form
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_train_graph_gpu.py -m gpu -x -q -k "collectives or bucketed" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/s12_pytest.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f' % d['ms_per_step'], 'graphs/step', d['graph_segments'], '|', d['collectives'], '| captured:', d.get('collectives_captured'))"; }
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for rep in 1 2; do
  for bb in 64 12; do
  timeout 300 python3 bench.py $B --batch $bb 2>/dev/null | tee $O/s12_plain_${bb}_$rep.json | show "B=$bb plain (no collectives)                "
  MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 300 python3 bench.py $B --batch $bb 2>/dev/null | tee $O/s12_segments_${bb}_$rep.json | show "B=$bb segments + torch.distributed          "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=host timeout 300 python3 bench.py $B --batch $bb 2>/dev/null | tee $O/s12_host_${bb}_$rep.json | show "B=$bb host-gated + torch.distributed        "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=host MRMT3_DDP_NATIVE=1 timeout 300 python3 bench.py $B --batch $bb 2>/dev/null | tee $O/s12_host_native_${bb}_$rep.json | show "B=$bb host-gated + mrmt3_allreduce          "
  done
done 2>&1 | tee $O/s12_collectives_ab.log
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
timeout 1500 python3 profiles/tools/overlap_emulation.py 20 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | tee $O/s12_overlap_emulation.log
