#!/bin/bash
# round 5: hunting an intermittent core dump of the GPU suite (session 16): the trainer test files over and over, whole logs kept
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3 4; do
  timeout 900 python3 -X faulthandler -m pytest tests/test_model_gpu.py tests/test_train_graph_gpu.py tests/test_train_infer_gpu.py -v -m gpu -p no:cacheprovider > $O/s18_run$i.log 2>&1
  rc=$?; echo "run $i exit $rc: $(grep -c PASSED $O/s18_run$i.log) passed"
  if [ $rc -ne 0 ]; then grep -n "Fatal Python error" -B5 -A45 $O/s18_run$i.log | head -120; break; fi
done
