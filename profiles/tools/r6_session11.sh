#!/bin/bash
# round 6, session 11: the whole GPU suite FIVE times over, each run one process (consecutive green runs with the
# capture-fallback tests in-process), then the long-lived-process stress loop on the final tree
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MRMT3_CAPTURE_LOG=$PWD/$O/capture11.log
for i in 1 2 3 4 5; do
  timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/s11_suite_$i.log 2>&1
  echo "run $i: exit $? :: $(tail -1 $O/s11_suite_$i.log)"
done | tee $O/r06_suite_five_times.txt
timeout 700 python3 profiles/tools/r6_capture_stress.py 300 600 early > $O/s11_stress.log 2>&1
echo "stress exit $? :: $(tail -1 $O/s11_stress.log)" | tee -a $O/r06_suite_five_times.txt
test -f $O/capture11.log && echo "failed captures logged: $(grep -c 'failed capture' $O/capture11.log) (the sabotage tests make 3 per suite run on purpose)" | tee -a $O/r06_suite_five_times.txt
