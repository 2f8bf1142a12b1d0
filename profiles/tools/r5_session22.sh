#!/bin/bash
# round 5: the captured-collectives test over and over until it aborts (one GPU-suite run in ~7 died there), with RCCL / HIP messages on
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export NCCL_DEBUG=WARN
for i in $(seq 1 30); do
  timeout 300 python3 -m pytest tests/test_train_graph_gpu.py -v -m gpu -p no:cacheprovider -k "collectives_captured or native_bucket" > $O/s22_run.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "run $i exit $rc"; cp $O/s22_run.log $O/s22_failed_run.log; grep -v "^  File\|^Extension" $O/s22_run.log | tail -60; break; fi
done
echo "last run: $i (exit $rc)"
