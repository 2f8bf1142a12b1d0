#!/bin/bash
# round 6, session 13: phase C of the one-pass attention backward with its transposed reads requested 2 / 3 / 4 K steps ahead of the MFMA
# that consumes them (variant libraries, MRMT3_TOOL_LIB) against the product library: MT3Net, 64 segments
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
  for L in "" profiles/tools/_ab/libmrmt3_hip_pf2.so profiles/tools/_ab/libmrmt3_hip_pf3.so profiles/tools/_ab/libmrmt3_hip_pf4.so; do
    MRMT3_TOOL_LIB=$L timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --no-roofline --extra-batch 0 > $O/ab13.json 2> $O/ab13.err
    python3 -c "
import json; d=json.load(open('$O/ab13.json')); print('${L:-product library}', 'ms_per_step %.3f' % d['ms_per_step'], 'loss %.5f' % d['final_loss'])"
  done
done | tee $O/r06_onepass_prefetch_ab.txt
