#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for d in 0 2 18 34 50 66 130 194 242; do
  MRMT3_ROWS_DBG=$d timeout 120 python3 profiles/tools/gemm_rows_trace.py 65536 1024 compact 2>&1 | grep "dbg"
done | tee $O/s10_kloop_parts_bm128.log
