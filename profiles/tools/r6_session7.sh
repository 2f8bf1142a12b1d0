#!/bin/bash
# round 6, session 7: the 320-key one-pass cross-attention backward — parity (the parametrised kernel test), the site's time against the
# two-pass kernels (kill line 180 us), and train_mrmt3 with / without it in one process each
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "onepass" 2>&1 | tail -5
timeout 300 python3 profiles/tools/r6_onepass320_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_onepass320_probe.txt
for i in 1 2; do
  for V in 1 0; do
    MRMT3_ATTN_ONEPASS_320=$V timeout 300 python3 bench.py --variant segmem_v2_with_prev --steps 30 --warmup 5 --no-cpu-baseline --no-inference --no-extra-workloads --no-roofline --extra-batch 0 > $O/ab7.json 2> $O/ab7.err
    python3 -c "
import json; d=json.load(open('$O/ab7.json')); print('MRMT3_ATTN_ONEPASS_320=$V', 'train_mrmt3 ms_per_step %.3f' % d['ms_per_step'], 'loss %.5f' % d['final_loss'])"
  done
done | tee $O/r06_onepass320_step_ab.txt
