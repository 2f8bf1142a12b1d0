#!/bin/bash
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py > $O/bench_default.out 2> $O/bench_default.err; echo "stdout lines: $(wc -l < $O/bench_default.out)"; head -c 300 $O/bench_default.out; echo
MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-inference --extra-batch 0 --no-roofline > $O/bench_fc.out 2> $O/bench_fc.err; echo "stdout lines under torchrun with RCCL: $(wc -l < $O/bench_fc.out)"; head -c 200 $O/bench_fc.out; echo
