#!/bin/bash
# round 5, session 13: bucket layout under emulated collectives — the last (exposed) bucket with 4 encoder layers + embeddings (42 MB) or with
# 1 (14 MB, one more boundary); the memory encoder's bucket leaves after its own backward either way; + DDP tests on the new layout
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python3 -m pytest tests/test_train_graph_gpu.py tests/test_ddp_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/s13_pytest.log
for last in 4 1 2; do
  echo "=== MRMT3_DDP_LAST_BUCKET_LAYERS=$last"
  MRMT3_DDP_LAST_BUCKET_LAYERS=$last timeout 900 python3 profiles/tools/overlap_emulation.py 20 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp"
done | tee $O/s13_bucket_layout.log
