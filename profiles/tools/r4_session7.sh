#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 2 18 34 50 66 130 194 242; do
  MRMT3_ROWS_DBG=$d MRMT3_ROWS_SKEW=0 timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 1024 compact 2>&1 | grep -v amdgpu.ids
done | tee $O/s7_kloop_parts.log
