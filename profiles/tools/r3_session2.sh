#!/bin/bash
# round-3 GPU session 2: attention mask rework A/B, new f32 kernels, bench-shape parity, a short bench
mkdir -p gpurun_out/r3
O=gpurun_out/r3
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn or gemm_tn_f32 or geglu_bwd_and or dropout" > $O/t_kernels.log 2>&1; tail -5 $O/t_kernels.log
for v in base v1 v2; do MRMT3_TOOL_LIB=profiles/tools/_ab/lib$v.so timeout 300 python profiles/tools/attn_micro.py 20 > $O/attn_micro_$v.log 2>&1; echo "== $v"; cat $O/attn_micro_$v.log; done
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "fp32_gradients or fp32_trainer" > $O/t_fp32.log 2>&1; tail -12 $O/t_fp32.log
timeout 1200 python -m pytest tests/test_bench_shape_gpu.py -x -q -s > $O/t_bench_shape.log 2>&1; tail -15 $O/t_bench_shape.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference > $O/bench_a.json 2> $O/bench_a.err; cat $O/bench_a.json | head -c 3000
# the two-process mismatch, localised: gradients compared step by step (eager, then graph replay)
timeout 420 python3 profiles/tools/two_rank_soak.py solo2 50 > $O/soak2_eager.log 2>&1; grep -v amdgpu.ids $O/soak2_eager.log | cut -c1-3000 | tail -12
timeout 420 python3 profiles/tools/two_rank_soak.py solo2 50 graph > $O/soak2_graph.log 2>&1; grep -v amdgpu.ids $O/soak2_graph.log | cut -c1-3000 | tail -12
