#!/bin/bash
# round 6, session 12: the 320-key one-pass backward with the fifth tile of a SIMD pair SHARED (each wave one query half of it): parity,
# site time, train_mrmt3 against the two-pass kernels, in-step time from a kernel trace
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "onepass" 2>&1 | tail -4
timeout 300 python3 profiles/tools/r6_onepass320_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_onepass320_probe_shared.txt
for i in 1 2; do
  for V in 1 0; do
    MRMT3_ATTN_ONEPASS_320=$V timeout 300 python3 bench.py --variant segmem_v2_with_prev --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --no-roofline --extra-batch 0 > $O/ab12.json 2> $O/ab12.err
    python3 -c "
import json; d=json.load(open('$O/ab12.json')); print('MRMT3_ATTN_ONEPASS_320=$V', 'train_mrmt3 ms_per_step %.3f' % d['ms_per_step'], 'loss %.5f' % d['final_loss'])"
  done
done | tee $O/r06_onepass320_step_ab_shared.txt
rm -rf $O/prof_mrmt3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mrmt3 -- python3 bench.py --variant segmem_v2_with_prev --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --no-extra-workloads --extra-batch 0 > $O/bench_mrmt3_under_rocprof.json 2> $O/bench_mrmt3_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_mrmt3 > $O/r06_step_breakdown_mrmt3.txt 2>&1; grep "onepass\|span" $O/r06_step_breakdown_mrmt3.txt | cut -c1-170
find $O/prof_mrmt3 -name "*kernel_trace.csv" -delete; find $O/prof_mrmt3 -name "*.db" -delete
