#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "onepass" 2>&1 | tail -8 | tee $O/s21_pytest.log
for sp in 0 1 auto; do
  if [ $sp = auto ]; then unset MRMT3_ATTN_ONEPASS_SPLIT; else export MRMT3_ATTN_ONEPASS_SPLIT=$sp; fi
  echo "== MRMT3_ATTN_ONEPASS_SPLIT=$sp"
  timeout 300 python3 profiles/tools/attn_micro.py 10 64 2>&1 | grep -E "dec-cross|enc-self"
done | tee $O/s21_attn_micro.log
unset MRMT3_ATTN_ONEPASS_SPLIT
for sp in 0 auto 0 auto; do
  if [ $sp = auto ]; then unset MRMT3_ATTN_ONEPASS_SPLIT; else export MRMT3_ATTN_ONEPASS_SPLIT=$sp; fi
  timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('SPLIT=$sp', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s21_step_ab.log
done
