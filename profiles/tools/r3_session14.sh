#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 > $O/bench_forced_collectives_world1.json 2> $O/bench_forced_collectives_world1.err; python3 -c "
import json;d=json.load(open('$O/bench_forced_collectives_world1.json'));print('forced collectives, world 1:',round(d['ms_per_step'],3),'ms', d['collectives'], 'graph segments', d['graph_segments'], 'host issue', round(d['host_issue_ms_per_step'],2))" || tail -5 $O/bench_forced_collectives_world1.err
MRMT3_DDP_FORCE_COLLECTIVES=1 MRMT3_GRAD_EXCHANGE=bf16 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 --no-roofline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('forced collectives, bf16 exchange:',round(d['ms_per_step'],3),'ms')"
