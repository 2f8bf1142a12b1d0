#!/bin/bash
# round 5, session 15: what the exchange's structure costs before any byte moves: segments + per-bucket weight-gradient launches vs the stream plumbing
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python3 profiles/tools/exchange_structure_cost.py 30 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | tee $O/s15_structure_cost.log
