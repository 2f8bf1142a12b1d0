#!/bin/bash
# round 6, final tree (library 110): GPU suite + smoke, rocprofv3 kernel trace / stats of the bench command, whole-step PMC traffic, full bench
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -5; timeout 600 python3 __graft_entry__.py smoke 2>&1 | tail -6; } | grep -v amdgpu.ids | tee $O/r06_pytest_gpu_and_smoke.txt
rm -rf $O/prof_final
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_final -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --extra-batch 0 > $O/r06_bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_final > $O/r06_step_breakdown.txt 2>&1; head -8 $O/r06_step_breakdown.txt
cp $(find $O/prof_final -name "*kernel_stats.csv" | head -1) $O/r06_bench_kernel_stats.csv
find $O/prof_final -name "*kernel_trace.csv" -delete; find $O/prof_final -name "*.db" -delete
rm -rf $O/prof_b12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b12 -- python3 bench.py --batch 12 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --no-extra-workloads --extra-batch 0 > $O/bench_b12_under_rocprof.json 2> $O/bench_b12_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_b12 12 > $O/r06_step_breakdown_b12.txt 2>&1; head -6 $O/r06_step_breakdown_b12.txt
find $O/prof_b12 -name "*kernel_trace.csv" -delete; find $O/prof_b12 -name "*.db" -delete
ROUND=6 bash profiles/tools/pmc_step_traffic.sh 2>&1 | tail -20
cp $O/r06_pmc_step_traffic.json profiles/ 2>/dev/null     # (on the box: so that the bench line below can fold the table in)
timeout 1500 python3 bench.py > $O/r06_bench_full.json 2> $O/r06_bench_full.err
python3 -c "
import json; d=json.load(open('$O/r06_bench_full.json')); print(d['value'], d['ms_per_step'], d.get('train_b12'))
for k in ('train_mrmt3','train_mrmt3_b12','train_long_context'): print(k, d[k]['ms_per_step'], d[k]['segments_per_s'], d[k]['model_tflops'], d[k]['top3_families_ms_per_step'])
r=d['roofline']; print(r['achieved'], r['frac'], (r.get('traffic') or {}).get('ratio'), r.get('step'))"
# same-box A/B of this round's kernel edits (pair bf16 conversion in every epilogue and in the attention kernels' P / dS packing, the
# one-compare causal mask, no spills in the attention kernels) against the round-5 kernels under this round's host code
AB=profiles/tools/_ab/libmrmt3_hip_r5kernels.so
if [ -f $AB ]; then
  for i in 1 2 3; do
    for L in "" "$AB"; do
      MRMT3_TOOL_LIB=$L timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --no-roofline > $O/ab.json 2> $O/ab.err
      python3 -c "
import json; d=json.load(open('$O/ab.json')); print('${L:-round-6 library}', 'ms_per_step %.3f' % d['ms_per_step'], 'b12 %.3f' % d['train_b12']['ms_per_step'])"
    done
  done | tee $O/r06_kernel_edits_ab.txt
fi
