#!/bin/bash
# round 6, session 4: the trajectory test with its numbers, then the whole GPU suite in one process on the tree with the owned capture
# stream, the spill-free attention kernels and the pair bf16 conversion; bench line; step breakdown under rocprofv3
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MRMT3_CAPTURE_LOG=$PWD/$O/capture4.log
MRMT3_TRAJ_F1=0.0 timeout 900 python3 -m pytest tests/test_trajectory_gpu.py -m gpu -q -s -p no:cacheprovider > $O/s4_traj.log 2>&1
echo "trajectory exit $?"; grep -v "amdgpu.ids\|^$" $O/s4_traj.log | tail -25
timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_trajectory_gpu.py > $O/s4_suite.log 2>&1
echo "suite exit $?"; grep -v "^  File\|^Extension\|Warning\|^  /\|^    \|^$\|Enable trace\|See https" $O/s4_suite.log | tail -25
timeout 300 python3 __graft_entry__.py smoke > $O/s4_smoke.log 2>&1; echo "smoke exit $?"; tail -4 $O/s4_smoke.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > $O/s4_bench.json 2> $O/s4_bench.err; echo "bench exit $?"; python3 -c "
import json; d=json.load(open('$O/s4_bench.json')); print({k: d[k] for k in ('value','ms_per_step','step_graph','graph_segments')}); print({k: (d[k]['ms_per_step'], d[k]['segments_per_s']) for k in d if k.startswith('train_')}); print(d['roofline']['families_ms_per_step'])"
