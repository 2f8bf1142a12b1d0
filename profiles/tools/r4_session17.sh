#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_ddp_gpu.py -x -q 2>&1 | tail -8 | tee $O/s17_pytest_ddp.log
timeout 900 python3 profiles/tools/lds_victim.py --inprocess 25 2>&1 | grep -v amdgpu.ids | tee $O/s17_inprocess_soak.log
