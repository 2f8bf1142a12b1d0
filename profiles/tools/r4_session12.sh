#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('$*', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s12_step_ab.log
}
run MRMT3_FUSE_ROWS=0
run MRMT3_FUSE_ROWS=4
run MRMT3_FUSE_ROWS=5
run MRMT3_FUSE_ROWS=6 MRMT3_FUSE_NORMBWD_MAXK=384
run MRMT3_FUSE_ROWS=6 MRMT3_FUSE_NORMBWD_MAXK=1152
run MRMT3_FUSE_ROWS=4 MRMT3_ROWS_BM=64
run MRMT3_FUSE_ROWS=0
run MRMT3_FUSE_ROWS=4
run MRMT3_FUSE_ROWS=7 MRMT3_FUSE_NORMBWD_MAXK=384
