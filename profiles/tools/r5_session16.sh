#!/bin/bash
# round 5, very last tree: GPU suite + smoke + the default bench command with its wall time
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | tail -5; timeout 600 python3 __graft_entry__.py smoke 2>&1 | tail -6; } | grep -v amdgpu.ids | tee $O/r05_pytest_gpu_and_smoke_last.txt
/usr/bin/time -v -o $O/bench_time.txt timeout 1500 python3 bench.py > $O/r05_bench_last.json 2> $O/r05_bench_last.err
grep "Elapsed" $O/bench_time.txt
python3 -c "
import json; d=json.load(open('$O/r05_bench_last.json')); print(d['value'], d['ms_per_step'], d.get('train_b12'))
for k in ('train_mrmt3','train_mrmt3_b12','train_long_context'): print(k, d[k]['ms_per_step'], d[k]['segments_per_s'])
print(d['roofline']['frac'], d['roofline']['traffic'] is not None, d['cpu_baseline']['value'])"
