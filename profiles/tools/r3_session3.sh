#!/bin/bash
# round-3 GPU session 3
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MRMT3_SOAK_TRACE=1 timeout 400 python3 profiles/tools/two_rank_soak.py solo2 45 > $O/soak3_trace.log 2>&1; grep -v amdgpu.ids $O/soak3_trace.log | grep -E "checksum|iterations" | cut -c1-400
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "onepass or attn_bwd" > $O/t_onepass.log 2>&1; tail -8 $O/t_onepass.log
for m in 1 0; do echo "== one-pass $m"; MRMT3_ATTN_ONEPASS=$m timeout 300 python profiles/tools/attn_micro.py 20 2>&1 | grep -v amdgpu.ids | tee $O/attn_micro_onepass$m.log; done
timeout 900 python -m pytest tests/test_bench_shape_gpu.py tests/test_model_gpu.py -x -q -s -k "bench_shape or fp32_trainer" > $O/t_bench_shape.log 2>&1; grep -E "bench shape|passed|failed|Error" $O/t_bench_shape.log | cut -c1-400
timeout 400 python profiles/tools/gemm_ab.py 3 20 12 > $O/gemm_ab_b12.txt 2>&1; grep -v amdgpu.ids $O/gemm_ab_b12.txt
timeout 400 python profiles/tools/gemm_ab.py 2 15 12 cold > $O/gemm_ab_b12_cold.txt 2>&1; grep -v amdgpu.ids $O/gemm_ab_b12_cold.txt
timeout 400 python profiles/tools/gemm_ab.py 2 15 64 cold > $O/gemm_ab_b64_cold.txt 2>&1; grep -v amdgpu.ids $O/gemm_ab_b64_cold.txt
bash profiles/tools/pmc_step_traffic.sh 2>&1 | tail -30
rm -rf $O/prof_b12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b12 -- python3 bench.py --batch 12 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 > $O/bench_b12_rocprof.json 2> $O/bench_b12_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_b12 > $O/r03_step_breakdown_b12.txt 2>&1; head -40 $O/r03_step_breakdown_b12.txt
find $O/prof_b12 -name "*kernel_trace.csv" -delete; find $O/prof_b12 -name "*.db" -delete; du -sh $O/prof_b12
