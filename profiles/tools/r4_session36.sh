#!/bin/bash
# round 4: what a bucket boundary costs at one rank (forced collectives): layers per bucket 1 / 2 / 4 / 8
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0"
for i in 1 2; do
for L in 1 2 4 8; do
MRMT3_DDP_LAYERS_PER_BUCKET=$L MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2960$L $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('layers per bucket $L:', round(d['value'],1), round(d['ms_per_step'],3), d.get('collectives'), round(d.get('host_issue_ms_per_step') or 0,3))"
done
timeout 600 python3 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain:', round(d['value'],1), round(d['ms_per_step'],3))"
done | tee $O/r04_bucket_boundary_cost.txt
