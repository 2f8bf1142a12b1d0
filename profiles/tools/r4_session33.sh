#!/bin/bash
# round 4: step breakdown at the reference's per-GPU batch (12 segments), current tree
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O/prof_b12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b12 -- python3 bench.py --batch 12 --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-roofline --extra-batch 0 > $O/bench_b12_under_rocprof.json 2> $O/bench_b12_under_rocprof.err
python3 profiles/tools/step_breakdown.py $O/prof_b12 12 > $O/r04_step_breakdown_b12.txt 2>&1; head -50 $O/r04_step_breakdown_b12.txt
find $O/prof_b12 -name "*kernel_trace.csv" -delete; find $O/prof_b12 -name "*.db" -delete
