#!/bin/bash
# round 6, session 3: the whole GPU suite in ONE process (nothing is delegated to child processes any more), the stress loop on the new
# trainer, smoke, a short bench line (N = 1) and the N > 1 code path on one GPU (--spawn: preflight lines on stderr)
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MRMT3_CAPTURE_LOG=$PWD/$O/capture3.log
timeout 1500 python3 -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/s3_suite.log 2>&1
echo "suite exit $?"; grep -v "^  File\|^Extension\|Warning\|^  /\|^    \|^$\|Enable trace\|See https" $O/s3_suite.log | tail -25
timeout 400 python3 profiles/tools/r6_capture_stress.py 120 300 early > $O/s3_stress.log 2>&1
echo "stress exit $?"; tail -5 $O/s3_stress.log
timeout 300 python3 __graft_entry__.py smoke > $O/s3_smoke.log 2>&1; echo "smoke exit $?"; tail -4 $O/s3_smoke.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference > $O/s3_bench.json 2> $O/s3_bench.err; echo "bench exit $?"; python3 -c "
import json; d=json.load(open('$O/s3_bench.json')); print({k: d[k] for k in ('value','ms_per_step','step_graph','graph_segments')}); print({k: (d[k]['ms_per_step'], d[k]['segments_per_s']) for k in d if k.startswith('train_')})"
MRMT3_DDP_FORCE_COLLECTIVES=1 timeout 600 python3 bench.py --spawn --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-extra-workloads --extra-batch 0 > $O/s3_bench_spawn.json 2> $O/s3_bench_spawn.err; echo "spawn bench exit $?"; grep "^rank" $O/s3_bench_spawn.err; python3 -c "
import json; d=json.load(open('$O/s3_bench_spawn.json')); print({k: d.get(k) for k in ('value','ms_per_step','collectives','graph_segments','launched_by','error','stage')}); print(json.dumps(d.get('preflight'))[:1500])"
test -f $O/capture3.log && grep -c "failed capture" $O/capture3.log
