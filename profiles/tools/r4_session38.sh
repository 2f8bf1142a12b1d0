#!/bin/bash
# round 4: transposed LDS reads of the flash-attention kernels through inline asm (no compiler-inserted vmcnt(0) per key tile) — parity, then A/B
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_fuzz_gpu.py -q -x -k "attn" 2>&1 | tail -3
{
for i in 1 2; do
echo "== base (builtin transposed reads)"; MRMT3_TOOL_LIB=profiles/tools/_ab/libbase.so python3 profiles/tools/attn_micro.py 20 2>&1 | grep "p=0"
echo "== asm transposed reads"; python3 profiles/tools/attn_micro.py 20 2>&1 | grep "p=0"
done
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0"
one() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  MRMT3_TOOL_LIB=profiles/tools/_ab/libbase.so $B 2>/dev/null | one "step, base"
  $B 2>/dev/null | one "step, asm transposed reads"
done
} | tee $O/r04_attn_tr_asm_ab.txt
