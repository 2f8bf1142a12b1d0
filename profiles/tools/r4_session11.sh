#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for K in 1024 384; do
for pad in "0 0" "64 0" "64 64" "8 8" "192 192"; do
  set -- $pad
  for d in 2 34; do
  PADW=$1 PADA=$2 MRMT3_ROWS_DBG=$d timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 $K compact 2>&1 | grep "dbg" | sed "s/^/K=$K /"
  done
done; done | tee $O/s11_pad.log
