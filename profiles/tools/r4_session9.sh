#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee $O/s9_pytest_rows.log
MRMT3_ROWS_BM=128 timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/s9_pytest_rows.log
MRMT3_ROWS_DBG=4 timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -v amdgpu.ids | sort -u | grep "gemm_rows:"
timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -v amdgpu.ids | tee $O/s9_rows_ab.log
{ timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 384; echo; timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 1024; } 2>&1 | grep -v amdgpu.ids | tee $O/s9_rows_trace.log
for f in 0 7 0 7; do
  MRMT3_FUSE_ROWS=$f timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('FUSE_ROWS=$f', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s9_step_ab.log
done
