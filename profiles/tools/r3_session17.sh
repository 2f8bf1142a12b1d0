#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(timeout 100 python3 profiles/tools/lds_victim.py --aggressor attn_fwd 65 > /dev/null 2>&1 &)
sleep 22
{ for v in 0 1 3 4; do timeout 30 profiles/tools/lds_canary 6 24 40 2 $v 2>&1 | grep -v amdgpu.ids | head -1 | cut -c1-300; done; } | tee $O/lds_fft_variants.log
