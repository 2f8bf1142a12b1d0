#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -3 | tee $O/s6_pytest_rows.log
MRMT3_ROWS_SKEW=0 timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -v amdgpu.ids | tee $O/s6_rows_ab.log
{ MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 384; echo;  MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 1024; } 2>&1 | grep -v amdgpu.ids | tee $O/s6_rows_trace.log
