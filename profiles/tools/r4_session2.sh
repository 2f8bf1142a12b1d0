#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 4 1 2 3; do
  echo "== MRMT3_ROWS_DBG=$d"
  MRMT3_ROWS_DBG=$d timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -v amdgpu.ids | sort -u | grep -E "gemm_rows:|addnorm   o/co|addnorm   wo|normbwd   d_qkv|geglubwd  d_wo"
done 2>&1 | tee $O/s2_rows_dbg.log
