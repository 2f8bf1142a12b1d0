#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for fam in attn_fwd gemm_tn_tile gemm_nt_tile gemm_nt8; do
  (timeout 80 python3 profiles/tools/lds_victim.py --aggressor $fam 40 > /dev/null 2>&1 &)
  sleep 22
  echo "== canary beside $fam"; timeout 30 profiles/tools/lds_canary 8 24 40 2>&1 | grep -v amdgpu.ids | cut -c1-900
  sleep 12
done | tee $O/lds_canary.log
echo "== canary alone"; timeout 30 profiles/tools/lds_canary 5 24 40 | tee -a $O/lds_canary.log
