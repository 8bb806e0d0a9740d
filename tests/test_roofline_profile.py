"""The bench line's per-kernel numbers are reproducible from the committed rocprofv3 kernel summary: for the
committed pair profiles/bench_line_r6.json (the line a `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5
--mode train --no-cpu-baseline` run printed) and profiles/kernel_stats_r6.csv (that run's kernel summary), the HIP-event microseconds
per C-ABI call of every kernel family agree with the rocprof microseconds of the kernels the call launches to 5 %, the dominant family
-- the weight-gradient ring kernel WITH its reduction -- included.  CPU-only: reads the two committed files."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hip_event_times_agree_with_the_rocprof_summary():
    import roofline_check as rc

    line = os.path.join(ROOT, "profiles", "bench_line_r6.json")
    stats = os.path.join(ROOT, "profiles", "kernel_stats_r6.csv")
    if not (os.path.exists(line) and os.path.exists(stats)):
        pytest.skip("profiles/bench_line_r6.json / kernel_stats_r6.csv not committed yet")
    ok, rows, dom = rc.check(line, stats)
    assert rows and any(r[0] == dom for r in rows), (dom, [r[0] for r in rows])
    bad = [(fam, round(ev, 1), round(us, 1)) for fam, ev, us, _, rel, _, _ in rows if rel > rc.TOL]
    assert ok and not bad, f"HIP-event us per call vs rocprof us per call differ by more than 5 %: {bad}"
