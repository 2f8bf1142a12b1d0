#!/bin/bash
# round 5: two more full-suite runs with the captured-collectives tests isolated
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  timeout 900 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $O/s24_run$i.log 2>&1; echo "run $i exit $?: $(tail -2 $O/s24_run$i.log | head -1)"
done
grep -n "Fatal Python error" -B3 -A25 $O/s24_run*.log | head -60
