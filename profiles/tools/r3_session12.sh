#!/bin/bash
# round-3 GPU session 12: the DDP tests five times over (first-try asserts), the whole GPU suite, the final bench + profiles
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_ddp_gpu.py -x -q 2>&1 | tail -1; done | tee $O/t_ddp_x5.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -4 $O/t_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; head -c 1500 $O/bench_final.json; echo
rm -rf $O/prof_final
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --extra-batch 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_final > $O/r03_step_breakdown.txt 2>&1; head -30 $O/r03_step_breakdown.txt
cp $(find $O/prof_final -name "*kernel_stats.csv" | head -1) $O/r03_bench_kernel_stats.csv
find $O/prof_final -name "*kernel_trace.csv" -delete; find $O/prof_final -name "*.db" -delete
timeout 300 python bench.py --variant segmem_v2_with_prev --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 > $O/bench_segmem_v2_with_prev.json 2>/dev/null; python3 -c "
import json;d=json.load(open('$O/bench_segmem_v2_with_prev.json'));print('MR-MT3 (segmem_v2_with_prev):',round(d['value'],1),'seg/s',round(d['ms_per_step'],3),'ms')"
timeout 300 python profiles/tools/long_context_step.py 12 5 2>&1 | grep -v amdgpu.ids | tee $O/long_context.txt
