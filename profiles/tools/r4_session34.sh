#!/bin/bash
# round 4: activation helpers with explicit FMAs (fewer vector instructions) — parity first, then same-box step A/B against the
# previous library (libcur.so) and against a packed-f32 build of the new sources (libslp.so: what the build contract costs)
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_gemm_rows_gpu.py tests/test_kernels_gpu.py -q -x -k "geglu or gemm_rows or gemm_nt" 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_model_gpu.py tests/test_bench_shape_gpu.py -q -x -k "bf16 or bench_shape" 2>&1 | tail -4
AB=profiles/tools/_ab
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0"
one() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
{
for i in 1 2 3; do
  $B 2>/dev/null | one "new(explicit fma)"
  MRMT3_TOOL_LIB=$AB/libcur.so $B 2>/dev/null | one "prev(contract off)"
  MRMT3_TOOL_LIB=$AB/libslp.so $B 2>/dev/null | one "new, packed f32 allowed (A/B only)"
done
} | tee $O/r04_activation_fma_ab.txt
{
echo "== in-tree (no packed f32)"; python3 profiles/tools/attn_micro.py 20 2>&1 | grep -v "^$" | tail -8
echo "== packed f32 allowed (A/B only)"; MRMT3_TOOL_LIB=$AB/libslp.so python3 profiles/tools/attn_micro.py 20 2>&1 | grep -v "^$" | tail -8
} | tee $O/r04_attn_micro_packed_ab.txt
