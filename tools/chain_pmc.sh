#!/bin/bash
# Memory-path counters of the chain kernels in isolation (tools/chain_bench.py), three passes
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/chain_pmc; mkdir -p gpurun_out/chain_pmc
P="timeout 300 rocprofv3 --output-format csv"
$P --pmc TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_THRASHING_STALL TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS -d gpurun_out/chain_pmc/a -o q -- python3 tools/chain_bench.py > /dev/null 2>&1 < /dev/null
$P --pmc TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ -d gpurun_out/chain_pmc/b -o q -- python3 tools/chain_bench.py > /dev/null 2>&1 < /dev/null
$P --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUSY TCC_HIT TCC_MISS TCC_EA0_WRREQ TCC_EA0_RDREQ -d gpurun_out/chain_pmc/c -o q -- python3 tools/chain_bench.py > /dev/null 2>&1 < /dev/null
$P --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d gpurun_out/chain_pmc/d -o q -- python3 tools/chain_bench.py > /dev/null 2>&1 < /dev/null
python3 - <<'PY' | tee gpurun_out/chain_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("gpurun_out/chain_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); n[name][r["Counter_Name"]] += 1
for name, c in agg.items():
    if "chain_a" not in name and "gemm_nt_sw" not in name: continue
    print(name[:110])
    for k, v in sorted(c.items()):
        print(f"   {k:36s} {v / max(n[name][k], 1):18.0f}")
PY
rm -rf gpurun_out/chain_pmc
