#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do timeout 900 python3 -m pytest tests/test_ddp_gpu.py -x -q -k "frontend_inside" 2>&1 | tail -3; done | tee $O/s28_two_rank_audio.log
timeout 900 python3 profiles/tools/lds_victim.py --inprocess 12 2>&1 | grep -v amdgpu.ids | tee $O/s28_inprocess_soak_final.log
