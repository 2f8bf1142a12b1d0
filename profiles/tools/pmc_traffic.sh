#!/bin/bash
# HBM traffic of every GEMM shape of the training step: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE needs 3 of the
# 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"), then pmc_traffic_parse.py applies the gfx950
# correction (FETCH_SIZE counts 128-byte requests as 64: x2) and writes profiles/r02_pmc_gemm_traffic.{json,txt}.
# Run from the repo root on the GPU box:  bash profiles/tools/pmc_traffic.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 profiles/tools/pmc_gemm_all.py > gpurun_out/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 profiles/tools/pmc_gemm_all.py > gpurun_out/pmc_write.log 2>&1
python3 profiles/tools/pmc_traffic_parse.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_plan.json gpurun_out
