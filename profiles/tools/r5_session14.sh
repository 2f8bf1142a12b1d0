#!/bin/bash
# round 5, session 14: more exchange parameters under emulated collectives (MT3Net, 64 and 12 segments): layers per bucket, the collective
# stream's priority, the bf16 exchange
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export EMU_CONFIGS="t5:64,t5:12"
run() { echo "=== $1"; env $1 timeout 600 python3 profiles/tools/overlap_emulation.py 20 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp\|^emulated"; }
{ run "MRMT3_DDP_LAYERS_PER_BUCKET=4"; run "MRMT3_DDP_LAYERS_PER_BUCKET=2"; run "MRMT3_DDP_LAYERS_PER_BUCKET=8"; run "MRMT3_DDP_STREAM_PRIO=-1"; run "MRMT3_GRAD_EXCHANGE=bf16"; } | tee $O/s14_exchange_parameters.log
