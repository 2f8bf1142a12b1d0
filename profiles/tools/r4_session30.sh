#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -k "test_bf16_gradients_vs_oracle_autograd" 2>&1 | tail -30
echo "=== with the previous library"
MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libprev.so timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -k "test_bf16_gradients_vs_oracle_autograd" 2>&1 | tail -8
