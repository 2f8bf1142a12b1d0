#!/bin/bash
# round-3 GPU session 14: the whole GPU suite, smoke, the bench line on the final tree
mkdir -p gpurun_out/r3
O=gpurun_out/r3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; tail -4 $O/t_all.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_final3.json 2> $O/bench_final3.err; python3 -c "
import json;d=json.load(open('$O/bench_final3.json'));print('bench:',round(d['value'],1),'seg/s',round(d['ms_per_step'],3),'ms; b12',d['train_b12']['ms_per_step'],'roofline',round(d['roofline']['frac'],4))"
