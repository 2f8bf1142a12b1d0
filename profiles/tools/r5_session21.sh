#!/bin/bash
# round 5: four more full GPU-suite runs on the last tree (with the orderly-teardown hook), whole logs kept: how often does a run die?
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do
  timeout 1500 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $O/s21_run$i.log 2>&1
  rc=$?; echo "run $i exit $rc: $(tail -3 $O/s21_run$i.log | grep -o '[0-9]* passed.*')"
  if [ $rc -ne 0 ]; then grep -n "Fatal Python error" -B8 -A60 $O/s21_run$i.log | head -150; break; fi
done | tee $O/s21_summary.log
