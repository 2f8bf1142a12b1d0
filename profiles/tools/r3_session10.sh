#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(timeout 90 python3 profiles/tools/lds_victim.py --aggressor attn_fwd 55 > /dev/null 2>&1 &)
sleep 22
{ for m in 6 7 2; do echo "== mode $m beside attn_fwd"; timeout 30 profiles/tools/lds_canary 7 24 40 $m 2>&1 | grep -v amdgpu.ids | cut -c1-700; done; } | tee $O/lds_alu_canary.log
sleep 25
{ for m in 6 7; do echo "== mode $m alone"; timeout 30 profiles/tools/lds_canary 4 24 40 $m 2>&1 | grep -v amdgpu.ids | cut -c1-300; done; } | tee -a $O/lds_alu_canary.log
