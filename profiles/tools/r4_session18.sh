#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "logmel" 2>&1 | tail -15 | tee $O/s18_pytest_logmel.log
timeout 300 python3 profiles/tools/logmel_micro.py 2>&1 | grep -v amdgpu.ids | tee $O/s18_logmel_micro.log
timeout 600 python3 profiles/tools/lds_victim.py --inprocess 8 attn_fwd 2>&1 | grep -v amdgpu.ids | tee $O/s18_inprocess_soak_wave.log
