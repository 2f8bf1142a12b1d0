#!/bin/bash
# round 4, library 107: co-residency soak once more on the final build (one process / two streams, then two processes)
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
{ echo "library $(python3 -c "import sys; sys.path.insert(0,'mr-mt3_amd'); from mrmt3 import lib; print(lib.load().mrmt3_version())" 2>/dev/null): one process, log-mel on stream A, one kernel family on stream B, 15 s per family"
timeout 900 python3 profiles/tools/lds_victim.py --inprocess 15 2>&1 | grep -v amdgpu.ids
echo; echo "two processes (victim: log-mel loop; aggressor: one family), 15 s per family"
timeout 900 python3 profiles/tools/lds_victim.py 15 attn_fwd,attn_bwd,gemm_nt8,tn_group 2>&1 | grep -v amdgpu.ids
} | tee $O/r04_soak_final_lib.txt
