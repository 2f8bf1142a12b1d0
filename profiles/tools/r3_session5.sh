#!/bin/bash
# round-3 GPU session 5: is the log-mel kernel repeatable beside a second process?  + the full GPU test suite
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python3 profiles/tools/logmel_repeat.py 1 3000 2 > $O/logmel_repeat_1proc.log 2>&1; grep -v amdgpu.ids $O/logmel_repeat_1proc.log | cut -c1-400 | tail -5
timeout 400 python3 profiles/tools/logmel_repeat.py 2 4000 2 > $O/logmel_repeat_2proc.log 2>&1; grep -v amdgpu.ids $O/logmel_repeat_2proc.log | cut -c1-400 | tail -12
timeout 400 python3 profiles/tools/logmel_repeat.py 2 1500 64 > $O/logmel_repeat_2proc_b64.log 2>&1; grep -v amdgpu.ids $O/logmel_repeat_2proc_b64.log | cut -c1-400 | tail -12
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -6 $O/t_all.log
