#!/bin/bash
# round 5, session 1: the whole GPU suite on the knob / diagnostics-free build (+86 tile-height cases), then the step A/B
# of this build against round 4's library (profiles/tools/_ab/libmrmt3_r4_107.so, built from HEAD~ sources)
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/s1_pytest.log
for rep in 1 2; do
  MRMT3_TOOL_LIB=$PWD/profiles/tools/_ab/libmrmt3_r4_107.so timeout 300 python3 bench.py --steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline 2>/dev/null | tee $O/s1_bench_r4lib_$rep.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('r4 lib  ', d['ms_per_step'], d['train_b12']['ms_per_step'])"
  timeout 300 python3 bench.py --steps 20 --warmup 3 --no-inference --no-cpu-baseline --no-roofline 2>/dev/null | tee $O/s1_bench_new_$rep.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('this lib', d['ms_per_step'], d['train_b12']['ms_per_step'])"
done 2>&1 | tee $O/s1_ab.log
