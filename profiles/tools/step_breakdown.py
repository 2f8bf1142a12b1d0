"""Cuts one replayed training step (between two adamw_kernel launches) out of a rocprofv3 kernel trace CSV and prints
the per-kernel totals: profiles/r02_step_breakdown.txt.   python3 profiles/tools/step_breakdown.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
a, b = ad[-8], ad[-7]                      # a step of the timed (graph-replayed) region
step = rows[a + 1:b + 1]
t0 = min(int(r["Start_Timestamp"]) for r in step)
t1 = max(int(r["End_Timestamp"]) for r in step)
agg = {}
for r in step:
    k = r["Kernel_Name"].split("(")[0][:70]
    e = agg.setdefault(k, [0, 0])
    e[0] += 1
    e[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(v[1] for v in agg.values())
batch = sys.argv[2] if len(sys.argv) > 2 else "64"          # optional second argument: the --batch the trace was taken at
model = sys.argv[3] if len(sys.argv) > 3 else "MT3Net"       # optional third / fourth: the model and the extra bench.py flags
flags = sys.argv[4] if len(sys.argv) > 4 else ""
print("One replayed training step (B = %s, %s, bf16, dropout on) from the rocprofv3 kernel trace of `python3 bench.py%s%s --steps 20 "
      "--warmup 5 --no-cpu-baseline --no-inference --extra-batch 0`, between two adamw_kernel launches: %d kernels, span %.3f ms, "
      "kernel time %.3f ms\n" % (batch, model, "" if batch == "64" else " --batch " + batch, (" " + flags) if flags else "",
                                len(step), (t1 - t0) / 1e6, busy / 1e6))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %4d launches %8.3f ms %5.1f %%  avg %8.1f us" % (k, v[0], v[1] / 1e6, 100 * v[1] / busy, v[1] / v[0] / 1e3))
