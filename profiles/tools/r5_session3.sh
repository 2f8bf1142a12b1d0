#!/bin/bash
# round 5, session 3: do two graphs on two streams replay side by side (the captured-collectives form timed out in session 2)?
# + the weight-gradients-under-attention probe + the tests session 2 did not reach
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python3 profiles/tools/two_graph_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/s3_two_graph_probe.log
timeout 600 python3 profiles/tools/wgrad_under_attention.py 4 2>&1 | grep -v amdgpu.ids | tee $O/s3_wgrad_under_attention.log
timeout 1500 python3 -m pytest tests/test_bench_shape_gpu.py tests/test_ddp_gpu.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/s3_pytest.log
