#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('$*', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s20_rot_ab.log
}
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_nt" 2>&1 | tail -3
MRMT3_GEMM8_ROT=1 timeout 300 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_nt and not bitwise" 2>&1 | tail -3
run MRMT3_GEMM8_ROT=0
run MRMT3_GEMM8_ROT=1
run MRMT3_GEMM8_ROT=0
run MRMT3_GEMM8_ROT=1
run MRMT3_GEMM8_ROT=1 MRMT3_FUSE_ROWS=0
run MRMT3_GEMM8_ROT=0 MRMT3_FUSE_ROWS=0
