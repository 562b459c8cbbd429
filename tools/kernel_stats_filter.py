#!/usr/bin/env python3
"""Drop kernels by name from a rocprofv3 `*_kernel_stats.csv` and renormalise the Percentage column (e.g. the one-off weight-packing
kernels of a profiled process: `kernel_stats_filter.py in.csv pack > out.csv`).  Prints what was dropped on stderr."""
import csv, sys
src, pats = sys.argv[1], [p.lower() for p in sys.argv[2:]] or ["pack"]
rows = list(csv.DictReader(open(src)))
keep = [r for r in rows if not any(p in r["Name"].lower() for p in pats)]
drop = [r for r in rows if r not in keep]
tot = sum(float(r["TotalDurationNs"]) for r in keep) or 1.0
for r in keep:
    r["Percentage"] = f"{100.0 * float(r['TotalDurationNs']) / tot:.6f}"
w = csv.DictWriter(sys.stdout, fieldnames=rows[0].keys(), quoting=csv.QUOTE_ALL)
w.writeheader()
w.writerows(keep)
print(f"dropped {len(drop)} kernels ({sum(float(r['TotalDurationNs']) for r in drop) / 1e6:.2f} ms): " + ", ".join(r['Name'][:40] for r in drop), file=sys.stderr)
