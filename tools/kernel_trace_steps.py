#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED STEPS of a bench.py run from rocprofv3's kernel trace.

bench.py launches ATen's ``spin_kernel`` once right before and once right after its timed region (``bench.trace_marker``); this tool
keeps the dispatches between the two and prints them in the layout of rocprofv3's ``*_kernel_stats.csv`` -- the whole-run stats file
also counts the bank pass (the same kernels on 256x256 shop images), the warm-up and the instrumented legs behind the timed region.

usage: kernel_trace_steps.py <..._kernel_trace.csv>  > stats_of_the_timed_steps.csv     (summary on stderr)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit(f"kernel_trace_steps.py: {len(marks)} marker launch(es) in {sys.argv[1]} -- need the two around the timed region")
lo, hi = marks[0], marks[1]
sel = rows[lo + 1:hi]
agg = collections.OrderedDict()
for r in sel:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg.setdefault(r["Kernel_Name"], []).append(d)
total = sum(sum(v) for v in agg.values())
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 2), min(v), max(v)])
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["End_Timestamp"])
print(f"timed region: {len(sel)} dispatches of {len(rows)} in the trace, {span / 1e6:.3f} ms between the markers, "
      f"{total / 1e6:.3f} ms of kernel time (two streams overlap)", file=sys.stderr)
