#!/bin/bash
# round 5, session 11: hipStreamWaitValue32 as the hand-off (no resident spinning kernel): does it work, what does it cost the chain
# (each mode under its own short timeout: a wait that is never released must not sit there)
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in agent; do
  timeout 150 python3 -u profiles/tools/wait_value_probe.py $mode 2>&1 | grep -v "amdgpu.ids"; echo "[mode $mode: exit $?]"
done | tee $O/s11_wait_value_probe.log
