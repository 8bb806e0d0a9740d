#!/usr/bin/env python3
"""GPU idle time inside one training step from a rocprofv3 kernel trace CSV: the step = the dispatches between two consecutive
adamw kernels (the last full step of the trace); busy = union of the kernels' [start, end] intervals; the gaps between the end of one
kernel and the start of the next while nothing else runs, by size."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
if len(sys.argv) > 2:  # trace_gaps.py trace.csv N: the last N dispatches instead of an optimizer step (decode: no adamw)
    n_last = int(sys.argv[2])
    ev = ev[-n_last:]
    ev.append((ev[-1][1], ev[-1][1], "adamw_kernel end marker"))
    ev.insert(0, (ev[0][0], ev[0][0], "adamw_kernel begin marker"))
marks = [i for i, e in enumerate(ev) if "adamw_kernel" in e[2]]
if len(sys.argv) > 2:
    marks = [marks[0], marks[0], marks[-1], marks[-1]]
# two adamw launches per step (dense + domain ranges): step boundaries = every second mark
ends = marks[1::2] if len(marks) >= 4 else marks
if len(ends) < 2:
    sys.exit("need at least two optimizer steps in the trace")
a, b = ends[-2] + 1, ends[-1] + 1
step = ev[a:b]
t0, t1 = step[0][0], max(e[1] for e in step)
busy, cur_s, cur_e = 0, step[0][0], step[0][1]
gaps = []
for s, e, n in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(((s - cur_e) / 1e3, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = (t1 - t0) / 1e3
ksum = sum(e[1] - e[0] for e in step) / 1e3
print(f"step: {len(step)} dispatches, wall {wall / 1e3:.2f} ms, busy (union) {busy / 1e6:.2f} ms, idle {wall / 1e3 - busy / 1e6:.2f} ms "
      f"({100 * (1 - busy / 1e3 / wall):.1f} %), sum of kernel durations {ksum / 1e3:.2f} ms")
g = sorted(x for x, _ in gaps)
if g:
    import statistics
    print(f"gaps: n = {len(g)}, median {statistics.median(g):.1f} us, mean {sum(g) / len(g):.1f} us, p90 {g[int(0.9 * len(g))]:.1f} us, max {g[-1]:.1f} us")
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 1e9)):
        sel = [x for x in g if lo <= x < hi]
        print(f"  {lo:>3} .. {hi if hi < 1e9 else 'inf':>4} us: {len(sel):5d} gaps, {sum(sel) / 1e3:6.2f} ms")
    big = sorted(gaps, key=lambda t: -t[0])[:8]
    for x, n in big:
        print(f"  {x:8.1f} us before {n[:90]}")
