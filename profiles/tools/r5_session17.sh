#!/bin/bash
# round 5: the GPU suite again with the whole log kept (session 16's run dumped core somewhere; its log was cut to the last lines)
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python3 -m pytest tests -v -m gpu -p no:cacheprovider > $O/s17_pytest_full.log 2>&1; echo "exit $?" >> $O/s17_pytest_full.log
grep -n "Fatal\|PASSED\|FAILED" $O/s17_pytest_full.log | tail -5
grep -n "Fatal Python error" -A40 $O/s17_pytest_full.log | head -80
