#!/bin/bash
# round 5, the tree as committed last: GPU suite (with the orderly-teardown hook) + smoke + the default bench command, timed
mkdir -p gpurun_out/r5
O=gpurun_out/r5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python3 -m pytest tests -q -m gpu -p no:cacheprovider > $O/s20_pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/s20_pytest.log
tail -4 $O/s20_pytest.log
timeout 600 python3 __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/s20_smoke.log
t0=$(date +%s); timeout 1500 python3 bench.py > $O/r05_bench_last.json 2> $O/r05_bench_last.err; echo "bench exit $? after $(( $(date +%s) - t0 )) s" | tee $O/s20_bench_time.log
python3 -c "
import json; d=json.load(open('$O/r05_bench_last.json')); print(d['value'], d['ms_per_step'], d.get('train_b12'))
for k in ('train_mrmt3','train_mrmt3_b12','train_long_context'): print(k, d[k]['ms_per_step'], d[k]['segments_per_s'])
print(d['roofline']['frac'], d['roofline']['traffic'] is not None, d['cpu_baseline']['value'])"
