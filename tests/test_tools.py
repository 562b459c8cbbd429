"""Host-side tools that shape the committed evidence (no GPU)."""
import csv
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for name, t0, t1 in rows:
            w.writerow(["KERNEL_DISPATCH", name, t0, t1])


def test_kernel_trace_steps_keeps_the_timed_region_only(tmp_path):
    """tools/kernel_trace_steps.py: per-kernel statistics of the dispatches BETWEEN the two marker launches bench.py brackets its timed
    region with -- the bank pass / warm-up before and the instrumented legs behind are left out; rows may come in any order."""
    rows = [("conv_a", 100, 110), ("conv_a", 120, 150),                         # bank pass / warm-up
            ("void at::native::spin_kernel(long)", 200, 201),
            ("conv_a", 300, 400), ("conv_b", 410, 430), ("conv_a", 500, 700),   # the timed steps
            ("void at::native::spin_kernel(long)", 800, 801),
            ("conv_a", 900, 5000)]                                              # an instrumented leg
    p = tmp_path / "b_kernel_trace.csv"
    _trace(p, list(reversed(rows)))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_trace_steps.py"), str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = {row["Name"]: row for row in csv.DictReader(io.StringIO(r.stdout))}
    assert set(out) == {"conv_a", "conv_b"}
    assert int(out["conv_a"]["Calls"]) == 2 and float(out["conv_a"]["AverageNs"]) == 150.0 and int(out["conv_a"]["MaxNs"]) == 200
    assert int(out["conv_b"]["Calls"]) == 1 and float(out["conv_b"]["TotalDurationNs"]) == 20.0
    assert "3 dispatches of 8" in r.stderr


def test_kernel_trace_steps_needs_both_markers(tmp_path):
    p = tmp_path / "t.csv"
    _trace(p, [("conv_a", 1, 2), ("void at::native::spin_kernel(long)", 3, 4), ("conv_a", 5, 6)])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_trace_steps.py"), str(p)], capture_output=True, text=True)
    assert r.returncode != 0 and "marker" in r.stderr
