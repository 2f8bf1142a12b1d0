#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "onepass" 2>&1 | tail -2
for v in cur new; do echo "== lib $v"; if [ $v = cur ]; then export MRMT3_TOOL_LIB=profiles/tools/_ab/libcur.so; else unset MRMT3_TOOL_LIB; fi; timeout 300 python profiles/tools/attn_micro.py 20 2>&1 | grep -E "cross|enc" ; done
unset MRMT3_TOOL_LIB
for v in cur new; do if [ $v = cur ]; then export MRMT3_TOOL_LIB=profiles/tools/_ab/libcur.so; else unset MRMT3_TOOL_LIB; fi; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 --no-roofline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('lib $v: B=64',round(d['ms_per_step'],3),'ms')"; done
unset MRMT3_TOOL_LIB
MRMT3_ATTN_LO=cross timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 --no-roofline 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('O_lo at cross sites only: B=64',round(d['ms_per_step'],3),'ms')"
MRMT3_ATTN_LO=cross timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_shape_gpu.py -x -q -s -k "bf16_gradients or bench_shape or trajectory" 2>&1 | grep -E "worst|bench shape|passed|failed|Error" | cut -c1-300
