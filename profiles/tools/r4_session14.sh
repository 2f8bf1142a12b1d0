#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fine in 0 3000 6000 10000 15000; do
  for dbg in 0 256; do
  echo "== MRMT3_ROWS_SKEW_FINE=$fine DBG=$dbg"
  MRMT3_ROWS_DBG=$dbg MRMT3_ROWS_SKEW_FINE=$fine timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 8 2>&1 | grep -E "addnorm   o/co|addnorm   wo|normbwd   d_qkv|normbwd   d_cq|geglubwd  d_wo|per step"
  done
done 2>&1 | tee $O/s14_skew_bm128.log
