#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  env "$@" timeout 300 python3 bench.py --batch 12 --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('B=12 $*', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s19_b12_ab.log
}
run MRMT3_FUSE_ROWS=0
run MRMT3_FUSE_ROWS=6
run MRMT3_FUSE_ROWS=7
run MRMT3_FUSE_ROWS=7 MRMT3_FUSE_NORMBWD_MAXK=2048
run MRMT3_FUSE_ROWS=0
run MRMT3_FUSE_ROWS=7 MRMT3_FUSE_NORMBWD_MAXK=2048
