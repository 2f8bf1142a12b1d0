#!/bin/bash
# round-3 GPU session 6: log-mel repeatability beside (a) a second log-mel loop, (b) a training loop
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python3 profiles/tools/logmel_repeat.py 2 400000 2 > $O/logmel_repeat_2proc.log 2>&1; grep -v amdgpu.ids $O/logmel_repeat_2proc.log | cut -c1-400 | tail -8
(timeout 300 python3 profiles/tools/two_rank_soak.py solo1 25 > $O/soak6_beside.log 2>&1 &)
sleep 20
timeout 250 python3 profiles/tools/logmel_repeat.py 1 500000 2 > $O/logmel_repeat_beside_training.log 2>&1; grep -v amdgpu.ids $O/logmel_repeat_beside_training.log | cut -c1-400 | tail -8
sleep 5; grep -v amdgpu.ids $O/soak6_beside.log | cut -c1-300 | tail -4
