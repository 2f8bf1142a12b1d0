#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 profiles/tools/lds_victims_all.py 8 attn_fwd 2>&1 | grep -v amdgpu.ids | tee $O/s24_victims_all.log
