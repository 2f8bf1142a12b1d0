#!/bin/bash
# round 6, session 2: the one-wave-per-SIMD NT product with trickled C stores (profiles/tools/gemm_w4_probe.hip) against
# gemm_nt8_kernel<bf16,8,0>, cold and warm, with its knock-outs; then the SQ counters of both kernels (separate --pmc passes).
mkdir -p gpurun_out/r6
O=gpurun_out/r6
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== abort repro (no abandon)"; timeout 120 python3 profiles/tools/r6_graph_abort_repro.py > $O/s2_abort_repro.txt 2>&1; echo "exit $?" >> $O/s2_abort_repro.txt; grep -v "^python3\|^/lib\|^/usr\|^/tmp\|frame #" $O/s2_abort_repro.txt | head -40
echo "== abort repro (abandon first)"; timeout 120 python3 profiles/tools/r6_graph_abort_repro.py abandon > $O/s2_abort_repro_abandon.txt 2>&1; echo "exit $?" >> $O/s2_abort_repro_abandon.txt; grep -v "^python3\|^/lib\|^/usr\|^/tmp\|frame #" $O/s2_abort_repro_abandon.txt | head -40
echo "== bucket bits"; timeout 300 python3 profiles/tools/r6_bucket_bits.py 2>&1 | grep -v "amdgpu.ids" | tee $O/s2_bucket_bits.txt
P=profiles/tools/gemm_w4_probe
for shape in "65536 512 512" "65536 2048 512" "65536 512 384" "65536 512 1024" "65536 1024 512"; do
  timeout 300 $P $shape 2>&1 | tee -a $O/s2_w4_probe.txt
done
rm -rf $O/w4pmc_a $O/w4pmc_b
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/w4pmc_a -- $P 65536 512 512 > $O/w4pmc_a.log 2>&1; tail -1 $O/w4pmc_a.log
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/w4pmc_b -- $P 65536 512 512 > $O/w4pmc_b.log 2>&1; tail -1 $O/w4pmc_b.log
python3 - <<'PY' | tee gpurun_out/r6/s2_w4_pmc.txt
import csv, glob, collections
for tag in "ab":
    fs = glob.glob("gpurun_out/r6/w4pmc_%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no counter file"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "gemm_" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(tag, k[:70], {c: int(x / max(1, n[(k, c)])) for c, x in sorted(v.items())}, "(per launch)")
PY
