"""Pairs the dispatches of the two PMC passes (FETCH_SIZE, WRITE_SIZE) of profiles/tools/pmc_gemm_all.py with the launch
plan and writes r02_pmc_gemm_traffic.json / .txt:  HBM bytes = 2 x FETCH_SIZE x unit + WRITE_SIZE x unit
(rocprofv3 reports both in KB on this image; FETCH_SIZE tallies 128-byte requests at 64 on gfx950: x2)."""
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, plan_path, out_dir = sys.argv[1:5]
plan = json.load(open(plan_path))
GEMM = ("gemm_nt8_kernel", "gemm_nt_kernel", "gemm_tn8_kernel", "gemm_tn_kernel")


def counters(d, name):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and any(k in r["Kernel_Name"] for k in GEMM):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][:48], float(r["Counter_Value"])))
    rows.sort()
    return rows


fe, wr = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
assert len(fe) == len(wr) == len(plan), (len(fe), len(wr), len(plan))
UNIT = 1024.0            # counter unit: KB
lines, tot = [], {"nt": [0.0, 0.0, 0], "tn": [0.0, 0.0, 0]}
for p, (_, kf, f), (_, kw, w) in zip(plan, fe, wr):
    assert kf == kw
    hbm = 2.0 * f * UNIT + w * UNIT
    alg = p["bytes"] if p["kind"] == "nt" else p["bytes"] - p["flops"] * 0 + p.get("slab_bytes", 0) - 4 * 0
    if p["kind"] == "tn":
        alg = (p["bytes"] - 0) + p["slab_bytes"]          # operands once + the slabs this kernel writes (C itself is the reduce kernel's)
    t = tot[p["kind"]]
    t[0] += hbm * p["per_step"]; t[1] += alg * p["per_step"]; t[2] += p["per_step"]
    lines.append("%-3s %-8s %-28s fetch x2 %8.1f MB  write %8.1f MB  total %8.1f MB  algorithmic %8.1f MB  ratio %.2f" % (
        p["kind"].upper(), p["name"], kf, 2 * f * UNIT / 1e6, w * UNIT / 1e6, hbm / 1e6, alg / 1e6, hbm / alg))
res = {"nt_bytes_per_step": tot["nt"][0], "nt_algorithmic_bytes_per_step": tot["nt"][1], "nt_launches_per_step": tot["nt"][2],
       "tn_bytes_per_step": tot["tn"][0], "tn_algorithmic_bytes_per_step": tot["tn"][1], "tn_launches_per_step": tot["tn"][2],
       "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on profiles/tools/pmc_gemm_all.py; bytes = 2 x FETCH + WRITE"}
json.dump(res, open(os.path.join(out_dir, "r02_pmc_gemm_traffic.json"), "w"), indent=1)
lines.append("per step: NT %.2f GB measured / %.2f GB algorithmic = %.2f   TN %.2f / %.2f = %.2f" % (
    tot["nt"][0] / 1e9, tot["nt"][1] / 1e9, tot["nt"][0] / tot["nt"][1], tot["tn"][0] / 1e9, tot["tn"][1] / 1e9, tot["tn"][0] / tot["tn"][1]))
open(os.path.join(out_dir, "r02_pmc_gemm_traffic.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
