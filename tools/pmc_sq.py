#!/usr/bin/env python3
"""Per-kernel averages of the SQ counters collected by tools/pmc_sq.sh."""
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[name] += 1
cols = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_IDX_ACTIVE",
        "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU_MFMA_MOPS_BF16")
print("rocprofv3 --pmc, bench.py --layers 8 --steps 1 --warmup 1; averages per launch, summed over the chip")
print(f"{'kernel':46s} {'n':>4s} " + " ".join(f"{c.replace('SQ_', '')[:14]:>14s}" for c in cols))
for name, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"] * n[kv[0]]):
    k = max(n[name], 1)
    if c["SQ_BUSY_CYCLES"] / k < 1e5:
        continue
    print(f"{name[:46]:46s} {k:4d} " + " ".join(f"{c[x] / k:14.0f}" for x in cols))
