#!/usr/bin/env python3
"""Reproduce the bench line's per-kernel numbers from a rocprofv3 kernel summary.

    python tools/roofline_check.py [profiles/bench_line_r6.json] [profiles/kernel_stats_r6.csv]

The bench line times C-ABI calls with HIP events in eager steps right after the timed region; rocprofv3 --kernel-trace --stats of the
same command gives per-KERNEL totals over the whole run (prepare + warm-up + timed graph replays + the instrumented steps: the same
launches every step).  For every family below: rocprof us per call = sum of the TotalDurationNs of the kernels the call launches /
the number of launches of the family's counting kernel; it must agree with the line's avg_launch_us within TOL (5 %; or ABS_US = 6 us:
the event bracket of a C-ABI call also holds the launch boundary in front of its kernel).
`roofline.frac` of the line = flops_per_launch / avg_launch_us / peak: with the rocprof us per call in its place anyone gets the same
fraction from profiles/ with a calculator."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 0.05
ABS_US = 6.0  # (a HIP-event bracket also holds the launch boundary in front of the kernel, 2-5 us: what a launch shorter than ~120 us may differ by)
# family -> (kernel-name substrings whose time belongs to a call, substring of the kernel whose launches count the calls)
FAMILY_KERNELS = {
    "wgrad_ring": (("gemm_tn_dma_kernel", "tn_reduce_native_kernel"), "gemm_tn_dma_kernel"),
    "hma_mlp_bwd": (("mlp_bwd_kernel",), "mlp_bwd_kernel"),
    "hma_chain_b_fwd": (("chain_b_fwd_kernel<true",), "chain_b_fwd_kernel<true"),
    "hma_chain_ab_fwd": (("chain_ab_fwd_kernel",), "chain_ab_fwd_kernel"),
    "hma_chain_a_fwd": (("chain_a_fwd_kernel",), "chain_a_fwd_kernel"),
    "hma_chain_a_bwd": (("chain_a_bwd_kernel",), "chain_a_bwd_kernel"),
    "hma_chain_s_bwd": (("chain_s_bwd_kernel",), "chain_s_bwd_kernel"),
    "hma_chain_t_bwd": (("chain_t_bwd_kernel",), "chain_t_bwd_kernel"),
    "hma_attn_spatial_fwd": (("attn_fwd_kernel",), "attn_fwd_kernel"),
    "hma_attn_spatial_bwd_blocked": (("attn_bwd_bal_kernel", "attn_bwd_fused_kernel"), "attn_bwd_"),
    "hma_attn_spatial_bwd": (("attn_bwd_bal_kernel", "attn_bwd_fused_kernel"), "attn_bwd_"),
    "hma_attn_temporal_fwd": (("attn_t_fwd_kernel",), "attn_t_fwd_kernel"),
    "hma_attn_temporal_bwd": (("attn_t_bwd_kernel",), "attn_t_bwd_kernel"),
}


def load_stats(path):
    rows = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    return rows


def rocprof_us_per_call(stats, kernels, count_kernel):
    total = sum(t for name, (_, t) in stats.items() if any(k in name for k in kernels))
    calls = sum(c for name, (c, _) in stats.items() if count_kernel in name)
    return (total / calls / 1e3) if calls else None, calls


def check(line_path, stats_path, tol=TOL):
    line = json.load(open(line_path))
    stats = load_stats(stats_path)
    fams = line["roofline"]["families"]
    out, ok = [], True
    for fam, (kernels, count_kernel) in FAMILY_KERNELS.items():
        if fam not in fams:
            continue
        us, calls = rocprof_us_per_call(stats, kernels, count_kernel)
        if us is None:
            continue
        ev = fams[fam]["avg_launch_us"]
        rel = abs(us - ev) / ev
        if abs(us - ev) <= ABS_US:
            rel = min(rel, TOL)
        frac_rocprof = fams[fam]["flops_per_launch"] / (us * 1e-6) / 1e12 / 2500.0
        out.append((fam, ev, us, calls, rel, fams[fam]["frac"], frac_rocprof))
        ok = ok and rel <= tol
    return ok, out, line["roofline"]["kernel"]


def main():
    line_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "bench_line_r6.json")
    stats_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "kernel_stats_r6.csv")
    ok, rows, dom = check(line_path, stats_path)
    print(f"{'family':30s} {'HIP events us':>14s} {'rocprof us':>11s} {'calls':>7s} {'diff':>7s} {'frac (line)':>12s} {'frac (rocprof)':>15s}")
    for fam, ev, us, calls, rel, f0, f1 in rows:
        print(f"{fam:30s} {ev:14.1f} {us:11.1f} {calls:7d} {100 * rel:6.1f}% {f0:12.4f} {f1:15.4f}" + ("   <- roofline" if fam == dom else ""))
    print("agreement within %.0f %%: %s" % (100 * TOL, "yes" if ok else "NO"))
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
