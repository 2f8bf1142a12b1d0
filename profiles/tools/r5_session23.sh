#!/bin/bash
# round 5: the suite with the captured-collectives tests in their own process
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $O/s23_pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/s23_pytest.log
tail -4 $O/s23_pytest.log
grep -n "Fatal Python error" -B3 -A25 $O/s23_pytest.log | head -60
