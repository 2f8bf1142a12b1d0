#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash profiles/tools/r3_session17.sh
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -4 $O/t_all.log
