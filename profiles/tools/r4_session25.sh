#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -2
{ echo "gemm_rows.hip built with -fno-slp-vectorize (no v_pk_*_f32)"; for d in 0 0 0; do timeout 300 python3 profiles/tools/lds_victims_all.py 10 attn_fwd "gemm_nt_addnorm,gemm_nt_addnorm(p=0)" 2>&1 | grep "victim gemm"; done; } | tee $O/s25_noslp.log
