#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_kernels_gpu.py tests/test_gemm_rows_gpu.py tests/test_fuzz_gpu.py tests/test_model_gpu.py tests/test_bench_shape_gpu.py -x -q 2>&1 | tail -4 | tee $O/s29_pytest.log
run() {
  env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('$*', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s29_step_ab.log
}
run A=new_gelu
run MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libprev.so
run A=new_gelu
run MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libprev.so
