#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 384; echo; MRMT3_ROWS_SKEW=500 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 384; echo;  MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 1024; } 2>&1 | grep -v amdgpu.ids | tee $O/s4_rows_trace.log
