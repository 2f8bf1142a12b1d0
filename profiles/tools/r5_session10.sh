#!/bin/bash
# round 5, session 10: what the bucketed exchange costs the step at N GPUs, measured on one GPU with emulated collectives
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 profiles/tools/overlap_emulation.py 20 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | tee $O/s10_overlap_emulation.log
