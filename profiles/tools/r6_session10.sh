#!/bin/bash
# round 6, session 10: the tree as committed last (320-key one-pass backward on, fused wi + GEGLU from 2048 rows): GPU suite in one
# process + smoke, the full default bench line, MR-MT3's step under rocprofv3
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -5; timeout 600 python3 __graft_entry__.py smoke 2>&1 | tail -6; } | grep -v amdgpu.ids | tee $O/r06_pytest_gpu_and_smoke.txt
timeout 1500 python3 bench.py > $O/r06_bench_full.json 2> $O/r06_bench_full.err
python3 -c "
import json; d=json.load(open('$O/r06_bench_full.json')); print(d['value'], d['ms_per_step'], d.get('train_b12'))
for k in ('train_mrmt3','train_mrmt3_b12','train_long_context'): print(k, d[k]['ms_per_step'], d[k]['segments_per_s'], d[k]['model_tflops'], d[k]['top3_families_ms_per_step'])
r=d['roofline']; print(r['achieved'], r['frac'], (r.get('traffic') or {}).get('ratio'), r.get('step'))"
rm -rf $O/prof_mrmt3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mrmt3 -- python3 bench.py --variant segmem_v2_with_prev --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --no-extra-workloads --extra-batch 0 > $O/bench_mrmt3_under_rocprof.json 2> $O/bench_mrmt3_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_mrmt3 > $O/r06_step_breakdown_mrmt3.txt 2>&1; head -14 $O/r06_step_breakdown_mrmt3.txt | cut -c1-150
find $O/prof_mrmt3 -name "*kernel_trace.csv" -delete; find $O/prof_mrmt3 -name "*.db" -delete
