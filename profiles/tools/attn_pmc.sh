#!/bin/bash
# SQ counters of the decoder self-attention kernels (one launch each): who issues, who waits.  Separate --pmc passes, kernel trace only.
mkdir -p gpurun_out/r4
O=gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf $O/apmc_a $O/apmc_b $O/apmc_c
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/apmc_a -- python3 profiles/tools/attn_tiny_causal.py > $O/apmc_a.log 2>&1; tail -1 $O/apmc_a.log
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/apmc_b -- python3 profiles/tools/attn_tiny_causal.py > $O/apmc_b.log 2>&1; tail -1 $O/apmc_b.log
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $O/apmc_c -- python3 profiles/tools/attn_tiny_causal.py > $O/apmc_c.log 2>&1; tail -1 $O/apmc_c.log
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    fs = glob.glob("gpurun_out/r4/apmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "attn_" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        print(tag, k[:60], {c: int(x) for c, x in sorted(v.items())})
PY
