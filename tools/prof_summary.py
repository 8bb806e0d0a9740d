#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into profiles-ready summaries (per-kernel time, HBM bytes per launch)."""
import csv, glob, os, sys, collections, json

out, tag = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "train"
summary = {}
stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, f"kernel_stats_{tag}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:60]:
            w.writerow([r.get("Name"), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"), r.get("MinNs"),
                        r.get("MaxNs"), r.get("StdDev")])
    for r in rows[:12]:
        print(f"{float(r['Percentage']):6.2f}%  {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:90]}")
for kind in ("fetch", "write"):
    files = glob.glob(os.path.join(out, f"pmc_{kind}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        name = r.get("Kernel_Name", "")
        agg[name][0] += 1
        agg[name][1] += float(r.get("Counter_Value", 0) or 0)
    summary[kind] = {k: {"launches": v[0], "counter_sum_kb": v[1]} for k, v in agg.items()}
if summary and mode in ("decode", "mar") and "fetch" in summary and "write" in summary:
    # the whole measured unit (warm-up included in the pass: divide by the passes the command ran).  decode: --warmup 2 --steps 1 =
    # 3 rollouts + the B = 1 latency leg's 5 small ones (~1.5 % of a B = 64 rollout each); mar: --warmup 2 --steps 1 = 3 steps.
    # gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM): doubled.
    fetch = sum(v["counter_sum_kb"] for v in summary["fetch"].values()) * 1024.0
    write = sum(v["counter_sum_kb"] for v in summary["write"].values()) * 1024.0
    units = 2.0 if mode == "decode" else 3.0
    depth = 8.0 if mode == "decode" else 1.0   # decode: measured on 4 of the 32 layers
    key = "bytes_per_rollout" if mode == "decode" else "bytes_per_step"
    summary["summary"] = {key: depth * (2.0 * fetch + write) / units, "fetch_raw_bytes": fetch, "write_bytes": write, "units": units,
                          "depth_scale": depth,
                          "note": f"2 x FETCH_SIZE + WRITE_SIZE summed over every kernel of the `bench.py --mode {mode}` PMC passes of "
                                  f"tools/prof_bench.sh (separate rocprofv3 --pmc runs), divided by the {units:.0f} units the command ran"
                                  + (", times 32 / 4 (measured on a 4-layer model)" if mode == "decode" else "")
                                  + "; includes the model's one-time construction copies (< 1 %)"}
    print(mode, key, summary["summary"][key] / 1e9, "GB")
if summary:
    json.dump(summary, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=1)
    for kind, d in summary.items():
        if kind == "summary":
            continue
        top = sorted(d.items(), key=lambda kv: -kv[1]["counter_sum_kb"])[:8]
        for k, v in top:
            print(f"{kind:6s} {v['counter_sum_kb']/max(v['launches'],1)/1024:10.1f} MB/launch (raw counter, KB units) x {v['launches']:5d}  {k[:80]}")
