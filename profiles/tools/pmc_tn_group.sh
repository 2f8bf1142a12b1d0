#!/bin/bash
# HBM traffic of the grouped weight-gradient launch (all TN sites of a step): separate rocprofv3 --pmc passes for
# FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md "rocprofv3 PMC slots"); bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB units,
# gfx950 counts 128-byte fetches as 64).  Run from the repo root on the GPU box: bash profiles/tools/pmc_tn_group.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python3 profiles/tools/pmc_tn_group.py 5 > gpurun_out/pmc_tn_group_timing.txt 2>&1
rm -rf gpurun_out/pmcg_fetch gpurun_out/pmcg_write
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcg_fetch -- python3 profiles/tools/pmc_tn_group.py > gpurun_out/pmcg_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcg_write -- python3 profiles/tools/pmc_tn_group.py > gpurun_out/pmcg_write.log 2>&1
python3 - <<'PY'
import csv, glob, json
def grab(d, name):
    out = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "tn8_group" in r["Kernel_Name"]:
                out[r["Kernel_Name"].split("(")[0]] = float(r["Counter_Value"])
    return out
fe, wr = grab("gpurun_out/pmcg_fetch", "FETCH_SIZE"), grab("gpurun_out/pmcg_write", "WRITE_SIZE")
plan = json.load(open("gpurun_out/pmc_tn_group_plan.json"))
lines = [open("gpurun_out/pmc_tn_group_timing.txt").read().strip().splitlines()[-1]]
tot = 0.0
for k in fe:
    b = 2 * fe[k] * 1024 + wr[k] * 1024
    tot += b
    lines.append("%-28s fetch x2 %9.1f MB  write %9.1f MB  total %9.1f MB" % (k, 2 * fe[k] * 1024 / 1e6, wr[k] * 1024 / 1e6, b / 1e6))
alg = plan["algorithmic_bytes"]
lines.append("family: %.2f GB measured / %.2f GB algorithmic (operands once + dW read and written) = %.2f; partial tiles %.2f GB each way" % (
    tot / 1e9, alg / 1e9, tot / alg, plan["slab_bytes"] / 1e9))
open("gpurun_out/r02_pmc_tn_group.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
