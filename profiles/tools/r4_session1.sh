#!/bin/bash
# round 4, session 1: the fused projection + row kernels — parity, micro A/B, step A/B
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_gemm_rows_gpu.py -x -q 2>&1 | tail -15 | tee $O/s1_pytest_rows.log
timeout 300 python3 profiles/tools/gemm_rows_ab.py 64 12 2>&1 | grep -v amdgpu.ids | tee $O/gemm_rows_ab_b64.txt
for f in 0 7 0 7; do
  MRMT3_FUSE_ROWS=$f timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('FUSE_ROWS=$f', r['value'], r['ms_per_step'])" | tee -a $O/s1_step_ab.log
done
