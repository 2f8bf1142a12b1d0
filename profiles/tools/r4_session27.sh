#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1800 python3 -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/s27_pytest_gpu.log
