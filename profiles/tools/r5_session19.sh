#!/bin/bash
# round 5: hunting the intermittent core dump of session 16 in the FULL GPU suite, whole logs kept
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do
  timeout 1500 python3 -m pytest tests -v -m gpu -p no:cacheprovider > $O/s19_run$i.log 2>&1
  rc=$?; echo "run $i exit $rc: $(grep -c PASSED $O/s19_run$i.log) passed"
  if [ $rc -ne 0 ]; then grep -n "Fatal Python error" -B8 -A60 $O/s19_run$i.log | head -150; break; fi
done
