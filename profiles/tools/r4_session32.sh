#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 > $O/r04_bench_forced_collectives_world1.json 2> $O/forced.err; tail -3 $O/forced.err
python3 -c "
import json; d=json.load(open('$O/r04_bench_forced_collectives_world1.json')); print(d['value'], d['ms_per_step'], d.get('collectives'), d.get('host_issue_ms_per_step'), d.get('graph'))"
