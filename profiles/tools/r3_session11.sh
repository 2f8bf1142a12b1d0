#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
MRMT3_SOAK_MEL=1 MRMT3_SOAK_TRACE=1 timeout 800 python3 profiles/tools/two_rank_soak.py solo2 100 > $O/soak11_mel_trace.log 2>&1; grep -v amdgpu.ids $O/soak11_mel_trace.log | grep -E "checksum|iterations|DIFFERS|differ" | cut -c1-400 | sort | uniq -c | sort -rn | head -12
