#!/bin/bash
# round 6, session 6: 12 segments per GPU — the fused wi + GEGLU launch from 2048 rows (the encoder's 3072 rows at 12 segments) against the
# default line of 4096 rows; alternated, one process each
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do
  for V in 4096 2048; do
    MRMT3_GEGLU_FUSED_MIN_ROWS=$V timeout 300 python3 bench.py --batch 12 --steps 50 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --no-roofline --extra-batch 0 > $O/ab6.json 2> $O/ab6.err
    python3 -c "
import json; d=json.load(open('$O/ab6.json')); print('MRMT3_GEGLU_FUSED_MIN_ROWS=$V', 'b12 ms_per_step %.3f' % d['ms_per_step'])"
  done
done | tee $O/r06_geglu_fused_min_rows_ab.txt
