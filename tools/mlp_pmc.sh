#!/bin/bash
# SQ counters of the fused MLP kernels in isolation (tools/mlp_bench.py), two passes of 8 counters
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/mlp_pmc; mkdir -p gpurun_out/mlp_pmc
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d gpurun_out/mlp_pmc/a -o q -- python3 tools/mlp_bench.py > /dev/null 2>&1 < /dev/null
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/mlp_pmc/b -o q -- python3 tools/mlp_bench.py > /dev/null 2>&1 < /dev/null
python3 - <<'PY' | tee gpurun_out/mlp_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/mlp_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name][r["Counter_Name"]] += 1
for name, c in agg.items():
    if "mlp" not in name: continue
    print(name)
    for k, v in sorted(c.items()):
        print(f"   {k:32s} {v / max(n[name][k], 1):16.0f}")
PY
rm -rf gpurun_out/mlp_pmc
