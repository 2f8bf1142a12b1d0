#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ echo "--- victim at the default addresses"; timeout 200 python3 profiles/tools/lds_victim.py 12 attn_fwd,gemm_tn_tile 0 2>&1 | grep -v amdgpu.ids
  echo "--- victim's buffers shifted by 8 GiB"; timeout 200 python3 profiles/tools/lds_victim.py 12 attn_fwd,gemm_tn_tile 8 2>&1 | grep -v amdgpu.ids; } | tee $O/lds_victim_shifted.log
