#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 profiles/tools/lds_victim.py 15 2>&1 | grep -v amdgpu.ids | tee $O/lds_victim.log
for b in 16 24 32 48 64; do timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 --no-roofline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('B=$b',round(d['value'],1),'seg/s',round(d['ms_per_step'],3),'ms  per segment',round(d['ms_per_step']/$b*1000,1),'us')"; done | tee $O/bench_batch_sweep.txt
