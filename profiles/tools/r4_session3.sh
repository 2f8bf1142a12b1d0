#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee $O/s3_pytest_rows.log
for sk in "0 0" "250 0" "500 0" "1000 0" "1500 0" "500 2000" "500 5000" "0 5000"; do
  set -- $sk
  echo "== MRMT3_ROWS_SKEW=$1 FINE=$2"
  MRMT3_ROWS_SKEW=$1 MRMT3_ROWS_SKEW_FINE=$2 timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -E "addnorm   o/co|addnorm   wo|normbwd   d_qkv|normbwd   d_cq|geglubwd  d_wo|per step"
done 2>&1 | tee $O/s3_rows_skew.log
