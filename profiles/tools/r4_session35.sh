#!/bin/bash
# round 4: the RCCL bucket path at world size 1 with forced collectives — torch.distributed vs the library's own communicator (MRMT3_DDP_NATIVE=1), same box
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0"
for i in 1 2; do
MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2953$i $B > $O/forced_torch_$i.json 2> $O/forced_torch.err
MRMT3_DDP_NATIVE=1 MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$i $B > $O/forced_native_$i.json 2> $O/forced_native.err
timeout 600 python3 $B > $O/plain_$i.json 2> $O/plain.err
done
tail -2 $O/forced_native.err
python3 -c "
import json
for n in ('forced_torch_1','forced_native_1','plain_1','forced_torch_2','forced_native_2','plain_2'):
    d=json.load(open('$O/'+n+'.json')); print(n, round(d['value'],1), round(d['ms_per_step'],3), d.get('collectives'), round(d.get('host_issue_ms_per_step') or 0,3))" | tee $O/r04_forced_collectives_native_ab.txt
cp $O/forced_native_2.json $O/r04_bench_forced_collectives_native_world1.json
