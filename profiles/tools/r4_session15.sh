#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee $O/s15_pytest_rows.log
MRMT3_ROWS_BM=128 timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/s15_pytest_rows.log
for fine in 0 5000 10000 15000; do
  echo "== MRMT3_ROWS_SKEW_FINE=$fine"
  MRMT3_ROWS_SKEW_FINE=$fine timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -E "addnorm   o/co|addnorm   wo|normbwd   d_qkv|normbwd   d_cq|geglubwd  d_wo|per step"
done 2>&1 | tee $O/s15_skew_bm128.log
timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 384 2>&1 | grep -v "amdgpu.ids\|^t =" | tee $O/s15_trace.log
