"""HBM bytes of ONE training step per kernel family from two rocprofv3 --pmc passes over profiles/tools/pmc_step.py
(FETCH_SIZE, WRITE_SIZE; separate passes, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
    bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024     (KB units; gfx950 tallies 128-byte fetches at 64: x2)
The last step of each pass is cut out between two adamw_kernel dispatches.
    python3 profiles/tools/pmc_step_parse.py <fetch dir> <write dir> <out.txt> [<out.json> [<lib version> [<batch>]]]
The JSON table is stamped with the library version and the batch it was measured on: bench.py folds it into its line
only when both match the run (a stale table must not ride along)."""
import csv
import glob
import json
import os
import sys

FAMILIES = (("gemm_rows_kernel<0", "gemm_nt_addnorm"), ("gemm_rows_kernel<1", "gemm_nt_normbwd"), ("gemm_rows_kernel<2", "gemm_nt_geglubwd"),
            ("gemm_nt8_kernel<unsigned short, false, 8, 1>", "gemm_nt_geglu"), ("gemm_nt8_kernel<unsigned short, false, 4, 1>", "gemm_nt_geglu"),
            ("gemm_nt8", "gemm_nt"), ("gemm_nt_kernel", "gemm_nt"), ("gemm_tn8_group", "gemm_tn_group"), ("tn8_group_reduce", "gemm_tn_group"),
            ("gemm_tn", "gemm_tn_other"), ("slab_reduce", "gemm_tn_other"), ("attn_fwd", "attn_fwd"), ("attn_bwd", "attn_bwd"),
            ("add_rmsnorm_fwd", "norm_fwd"), ("add_rmsnorm_bwd", "norm_bwd"), ("dw_reduce", "norm_bwd"), ("geglu_bwd", "geglu_bwd"),
            ("geglu_fwd", "geglu_fwd"), ("ce_", "lmhead_ce"), ("adamw", "adamw"), ("transpose", "adamw"), ("logmel", "logmel"),
            ("embed", "embed"), ("eb_", "embed"), ("addpos", "embed"), ("dropmask", "embed"), ("cast", "cast"))


def family(kernel):
    for pat, fam in FAMILIES:
        if pat in kernel:
            return fam
    return "other (torch fill / copy)"


def last_step(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "adamw_kernel" in r[1]]
    assert len(ends) >= 2, "need at least two steps in the trace"
    return rows[ends[-2] + 1:ends[-1] + 1]


fetch_dir, write_dir, out_txt = sys.argv[1:4]
fe, wr = last_step(fetch_dir, "FETCH_SIZE"), last_step(write_dir, "WRITE_SIZE")
assert len(fe) == len(wr), (len(fe), len(wr))
fam = {}
for (_, kf, f), (_, kw, w) in zip(fe, wr):
    assert kf == kw, (kf, kw)
    t = fam.setdefault(family(kf), [0.0, 0.0, 0])
    t[0] += 2.0 * f * 1024.0
    t[1] += w * 1024.0
    t[2] += 1
tot_r, tot_w = sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values())
lines = ["one eager training step (MT3Net, 64 segments, bf16, dropout on), %d kernels; bytes = 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, separate passes)" % len(fe),
         "%-28s %8s %10s %10s %10s" % ("family", "launches", "read GB", "write GB", "total GB")]
for k, v in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    lines.append("%-28s %8d %10.2f %10.2f %10.2f" % (k, v[2], v[0] / 1e9, v[1] / 1e9, (v[0] + v[1]) / 1e9))
lines.append("%-28s %8d %10.2f %10.2f %10.2f" % ("step", len(fe), tot_r / 1e9, tot_w / 1e9, (tot_r + tot_w) / 1e9))
open(out_txt, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
if len(sys.argv) > 4:
    json.dump({"step_bytes": tot_r + tot_w, "read_bytes": tot_r, "write_bytes": tot_w, "kernels": len(fe),
               "families": {k: {"launches": v[2], "read_bytes": v[0], "write_bytes": v[1]} for k, v in fam.items()},
               "lib_version": int(sys.argv[5]) if len(sys.argv) > 5 else None, "batch": int(sys.argv[6]) if len(sys.argv) > 6 else 64,
               "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over profiles/tools/pmc_step.py, last step; bytes = 2 x FETCH + WRITE"},
              open(sys.argv[4], "w"), indent=1)
