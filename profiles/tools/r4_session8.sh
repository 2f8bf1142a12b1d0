#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in 0 7 1 2 4 0 7; do
  MRMT3_FUSE_ROWS=$f timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('FUSE_ROWS=$f', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s8_step_ab.log
done
