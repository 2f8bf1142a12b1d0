#!/bin/bash
# round 5, session 5: the two-graph form with a relaxed poll in the hand-off wait (+ collective stream priority A/B), the resident-keys
# cross-attention forward (tests, micro, step A/B)
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_train_graph_gpu.py -m gpu -x -q -k "attn or flag or collectives" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/s5_pytest.log
B="--steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline --no-extra-workloads --extra-batch 0"
show() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', 'ms/step %.3f' % d['ms_per_step'], 'graphs/step', d['graph_segments'], '|', d['collectives'], '| captured:', d.get('collectives_captured'))"; }
for rep in 1 2; do
  timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_plain_$rep.json | show "plain, resident cross-attention fwd "
  MRMT3_ATTN_RESIDENT=0 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_plain_stream_$rep.json | show "plain, streaming cross-attention fwd"
  timeout 300 python3 bench.py $B --variant segmem_v2_with_prev 2>/dev/null | tee $O/s5_mrmt3_$rep.json | show "MR-MT3, resident (320 keys)         "
  MRMT3_ATTN_RESIDENT=0 timeout 300 python3 bench.py $B --variant segmem_v2_with_prev 2>/dev/null | tee $O/s5_mrmt3_stream_$rep.json | show "MR-MT3, streaming                   "
done 2>&1 | tee $O/s5_resident_ab.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for rep in 1 2; do
  timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_plain2_$rep.json | show "plain (no collectives)              "
  MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_forced_torch_$rep.json | show "segments + torch.distributed        "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=1 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_forced_twograph_$rep.json | show "two graphs, collective stream HIGH  "
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=1 MRMT3_DDP_STREAM_PRIO=0 timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_forced_twograph_normal_$rep.json | show "two graphs, collective stream normal"
  MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_DDP_GRAPH=inline timeout 300 python3 bench.py $B 2>/dev/null | tee $O/s5_forced_inline_$rep.json | show "one graph, in-line                  "
done 2>&1 | tee $O/s5_collectives_ab.log
