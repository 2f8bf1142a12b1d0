#!/bin/bash
# round 4: rocprofv3 kernel trace + stats of the bench command, one replayed step cut out of it, whole-step PMC traffic
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O/prof_final
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --extra-batch 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_final > $O/r04_step_breakdown.txt 2>&1; head -24 $O/r04_step_breakdown.txt
cp $(find $O/prof_final -name "*kernel_stats.csv" | head -1) $O/r04_bench_kernel_stats.csv
find $O/prof_final -name "*kernel_trace.csv" -delete; find $O/prof_final -name "*.db" -delete
python3 -c "
import json;d=json.load(open('$O/bench_under_rocprof.json'));r=d['roofline'];print('under rocprof:',round(d['ms_per_step'],3),'ms; gemm_nt family avg launch',round(r['avg_launch_ms']*1e3,1),'us, launches',r['launches'],'frac',round(r['frac'],4))"
bash profiles/tools/pmc_step_traffic.sh 2>&1 | tail -30
