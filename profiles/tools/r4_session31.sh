#!/bin/bash
# round 4, final tree (library 107): GPU suite + smoke, rocprofv3 kernel trace / stats of the bench command, whole-step PMC traffic, full bench
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ timeout 1800 python3 -m pytest tests -q -m gpu 2>&1 | tail -5; timeout 600 python3 __graft_entry__.py smoke 2>&1 | tail -6; } | tee $O/r04_pytest_gpu_and_smoke.txt
rm -rf $O/prof_final
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --extra-batch 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_final > $O/r04_step_breakdown.txt 2>&1; head -8 $O/r04_step_breakdown.txt
cp $(find $O/prof_final -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
find $O/prof_final -name "*kernel_trace.csv" -delete; find $O/prof_final -name "*.db" -delete
bash profiles/tools/pmc_step_traffic.sh 2>&1 | tail -20
timeout 1200 python3 bench.py > $O/r04_bench_full.json 2> $O/r04_bench_full.err
python3 -c "
import json; d=json.load(open('$O/r04_bench_full.json')); print(d['value'], d['ms_per_step'], d.get('train_b12')); r=d['roofline']; print(r['achieved'], r['frac'], r.get('traffic',{}).get('ratio'), r.get('step'))"
