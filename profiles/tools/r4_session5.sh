#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee $O/s5_pytest_rows.log
for sk in "0 0" "500 0"; do
  set -- $sk
  echo "== MRMT3_ROWS_SKEW=$1 FINE=$2"
  MRMT3_ROWS_SKEW=$1 MRMT3_ROWS_SKEW_FINE=$2 timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/s5_rows_skew.log
{ MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 384; echo;  MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 1024; } 2>&1 | grep -v amdgpu.ids | tee $O/s5_rows_trace.log
