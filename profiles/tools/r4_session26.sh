#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ echo "== library built WITHOUT packed f32 arithmetic (the default build)"; timeout 600 python3 profiles/tools/lds_victims_all.py 8 attn_fwd 2>&1 | grep "victim";
  echo "== library built WITH packed f32 arithmetic (SLP_FLAG= : the round 1-3 build), same tool"; MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libmrmt3_slp.so timeout 600 python3 profiles/tools/lds_victims_all.py 8 attn_fwd "logmel (round-1 kernel),gemm_nt_addnorm,gemm_nt_addnorm(p=0),add_rmsnorm_fwd,gemm_nt_normbwd,gemm_nt_geglubwd" 2>&1 | grep "victim"; } | tee $O/s26_victims_nopk.log
run() {
  env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        r = json.loads(l); print('$*', round(r['value'],1), round(r['ms_per_step'],3))" | tee -a $O/s26_step_ab.log
}
run A=nopk
run MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libmrmt3_slp.so
run A=nopk
run MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libmrmt3_slp.so
run A=nopk MRMT3_FUSE_ROWS=7
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/s26_pytest_gpu.log
