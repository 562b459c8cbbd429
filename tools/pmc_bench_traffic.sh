#!/bin/bash
# HBM traffic of the conv kernels during one bench run (GPU box): two separate --pmc passes
# (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -- they do not fit together), per the guide.
# Output: gpurun_out/pmc_traffic_<tag>.json  (per-launch averages, KB as reported by rocprofv3)
TAG=${1:-r01}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>/tmp/pmc_$c.err || echo "pass $c failed"
done
python3 - <<PY
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"/tmp/pmc_{c}/p_counter_collection.csv")):
        name = r["Kernel_Name"]
        key = None
        for t in ("float", "_Float16"):
            for bm in (128, 64):
                for bn in (128, 64):
                    if f"conv_igemm<{t}, {bm}, {bn}>" in name:
                        key = f"conv_igemm<{t},{bm},{bn}>"
        if key and r["Counter_Name"] == c:
            agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": v[0], "avg_per_launch": v[1] / v[0], "total": v[1]} for k, v in agg.items()}
json.dump(out, open("$R/gpurun_out/pmc_traffic_$TAG.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
