#!/bin/bash
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python3 bench.py > $O/r04_bench_full.json 2> $O/r04_bench_full.err; tail -c 600 $O/r04_bench_full.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4/r04_bench_full.json"))
print(d["value"], d["ms_per_step"], d.get("train_b12"))
r = d["roofline"]; print("roofline", r["achieved"], r["frac"], r["launches"], r.get("traffic"), r.get("step"))
print({k: round(v, 3) for k, v in r["families_ms_per_step"].items()})
print(json.dumps(d.get("inference"))[:1500])
print(json.dumps(d.get("cpu_baseline"))[:600])
PY
