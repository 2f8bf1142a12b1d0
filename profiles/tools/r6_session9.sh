#!/bin/bash
# round 6, session 9: the 320-key one-pass backward, who does delta + dQ: every wave its eighth (default) against the light waves all of
# it with their two dQ tiles interleaved (MRMT3_ONEPASS320_DUTIES=1): parity of both, site time, train_mrmt3
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for D in 0 1; do
  echo "== MRMT3_ONEPASS320_DUTIES=$D"
  MRMT3_ONEPASS320_DUTIES=$D timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "onepass" 2>&1 | tail -2
  MRMT3_ONEPASS320_DUTIES=$D timeout 300 python3 profiles/tools/r6_onepass320_probe.py 2>&1 | grep "Lk = 320"
done | tee $O/r06_onepass320_duties.txt
for i in 1 2; do
  for D in 0 1; do
    MRMT3_ONEPASS320_DUTIES=$D timeout 300 python3 bench.py --variant segmem_v2_with_prev --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --no-roofline --extra-batch 0 > $O/ab9.json 2> $O/ab9.err
    python3 -c "
import json; d=json.load(open('$O/ab9.json')); print('MRMT3_ONEPASS320_DUTIES=$D', 'train_mrmt3 ms_per_step %.3f' % d['ms_per_step'])"
  done
done | tee -a $O/r06_onepass320_duties.txt
