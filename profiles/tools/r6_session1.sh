#!/bin/bash
# round 6, session 1: the new world-2 segment-memory tests; then the failed-capture abort hunted two ways —
# the whole GPU suite IN ONE PROCESS (MRMT3_TEST_CHILD=1 keeps the captured-collectives tests in-process), and a stress loop
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MRMT3_CAPTURE_LOG=$PWD/$O/capture.log
timeout 900 python3 -m pytest tests/test_ddp_gpu.py -m gpu -x -q -p no:cacheprovider > $O/s1_ddp.log 2>&1
echo "ddp tests exit $?"; tail -5 $O/s1_ddp.log
MRMT3_TEST_CHILD=1 timeout 1200 python3 -m pytest tests -m gpu -q -p no:cacheprovider -W always > $O/s1_suite_inproc.log 2>&1
echo "in-process suite exit $?"; grep -v "^  File\|^Extension" $O/s1_suite_inproc.log | tail -30
timeout 500 python3 profiles/tools/r6_capture_stress.py 400 400 early > $O/s1_stress_early.log 2>&1
echo "stress exit $?"; tail -12 $O/s1_stress_early.log
ls -la $O; test -f $O/capture.log && head -80 $O/capture.log
