#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for K in 1024 384; do
for bm in 128 64; do
  for d in 2 258 34 290 226 482; do
  MRMT3_ROWS_BM=$bm MRMT3_ROWS_DBG=$d timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 $K compact 2>&1 | grep "dbg" | sed "s/^/K=$K BM=$bm /; s/; row phase median -[0-9.]*//"
  done
done; done | tee $O/s13_rot.log
