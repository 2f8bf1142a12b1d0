#!/bin/bash
# round-3 GPU session 4
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MRMT3_SOAK_TRACE=1 timeout 400 python3 profiles/tools/two_rank_soak.py solo2 45 > $O/soak4_trace.log 2>&1; grep -v amdgpu.ids $O/soak4_trace.log | grep -E "checksum|iterations|DIFFERS" | cut -c1-500 | sort | uniq -c | sort -rn | head -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "onepass or attn_bwd or gemm8_dispatch or gemm_nt8_pingpong or rmsnorm" > $O/t_onepass.log 2>&1; tail -8 $O/t_onepass.log
for m in 1 0; do echo "== one-pass $m"; MRMT3_ATTN_ONEPASS=$m timeout 300 python profiles/tools/attn_micro.py 20 2>&1 | grep -v amdgpu.ids | tee $O/attn_micro_onepass$m.log; done
for v in cur v3 v4; do echo "== lib $v"; MRMT3_TOOL_LIB=profiles/tools/_ab/lib$v.so timeout 300 python profiles/tools/attn_micro.py 20 2>&1 | grep -v amdgpu.ids | grep "p=0.1" | tee $O/attn_micro_lib$v.log; done
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -s -k "fp32_trainer" 2>&1 | grep -E "fp32 trainer|passed|failed|Error" | cut -c1-300
for b in 64 12; do timeout 600 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 > $O/bench_s4_b$b.json 2> $O/bench_s4_b$b.err; python3 -c "
import json;d=json.load(open('$O/bench_s4_b$b.json'));print('B=$b',round(d['value'],1),'seg/s',round(d['ms_per_step'],3),'ms');r=d['roofline'];print({k:round(v,3) for k,v in r['families_ms_per_step'].items()})"; done
MRMT3_ATTN_ONEPASS=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 --no-roofline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('B=64 two-pass attention backward:',round(d['ms_per_step'],3),'ms')"
