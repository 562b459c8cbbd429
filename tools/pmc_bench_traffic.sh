#!/bin/bash
# HBM traffic + matrix-pipe utilisation of the conv kernels during one bench run (GPU box): separate --pmc passes
# (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -- they do not fit together), per the guide; never mixed with trace domains
# other than --kernel-trace.  usage: tools/pmc_bench_traffic.sh <tag> [bench flags...]
# Output: gpurun_out/pmc_traffic_<tag>.json  (per-launch averages; FETCH/WRITE in KiB as reported by rocprofv3)
TAG=${1:-r01}; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16")
i=0
for c in "${PASSES[@]}"; do
  rm -rf /tmp/pmc_$i
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$i -o p -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --single-stream "$@" > /dev/null 2>/tmp/pmc_$i.err || echo "pass $c failed"
  i=$((i+1))
done
python3 - <<PY
import csv, collections, json, re, os
out = {}
pat = re.compile(r"(conv_igemm(?:_bx3)?|conv3x3_wino)<([^>]*)>")
for i in range($i):
    f = f"/tmp/pmc_{i}/p_counter_collection.csv"
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for r in csv.DictReader(open(f)):
        if "conv3x3_wino24" in r["Kernel_Name"]:
            mm = re.search(r"conv3x3_wino24<(\d+)>", r["Kernel_Name"])
            key = "conv3x3_wino24pc" if "conv3x3_wino24pc" in r["Kernel_Name"] else f"conv3x3_wino24<{mm.group(1)}>" if mm else "conv3x3_wino24"
            a = agg[r["Counter_Name"]][key]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            continue
        if "conv1x1_pc" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]["conv1x1_pc"]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            continue
        if "conv3x3_f16pc" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]["conv3x3_f16pc"]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            continue
        mh = re.search(r"pw_swh_kernel<(\d+), *(\d+)", r["Kernel_Name"])
        if mh:
            a = agg[r["Counter_Name"]][f"conv1x1_swh<{mh.group(1)},{mh.group(2)}>"]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            continue
        ms = re.search(r"pw_sw_kernel<(\d+), *(\d+)", r["Kernel_Name"])
        if ms:
            a = agg[r["Counter_Name"]][f"conv1x1_sw<{ms.group(1)},{ms.group(2)}>"]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            continue
        mg = re.search(r"conv_igemmIDF16_Li(\d+)ELi(\d+)ELi(\d+)", r["Kernel_Name"])     # rocprofv3 leaves _Float16 instantiations mangled
        if mg:
            a = agg[r["Counter_Name"]][f"conv_igemm<_Float16,{mg.group(1)},{mg.group(2)}>"]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            continue
        m = pat.search(r["Kernel_Name"])
        if not m:
            continue
        args = [a.strip() for a in m.group(2).split(",")]
        key = (f"{m.group(1)}<{args[0]}>" if m.group(1).endswith("wino") else
               f"{m.group(1)}<{args[0]},{args[1]}>" if m.group(1).endswith("bx3") else f"{m.group(1)}<{args[0]},{args[1]},{args[2]}>")
        a = agg[r["Counter_Name"]][key]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for c, d in agg.items():
        out[c] = {k: {"launches": v[0], "avg_per_launch": v[1] / v[0], "total": v[1], "avg_us": v[2] / v[0]} for k, v in d.items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "SQ_BUSY_CU_CYCLES" in out:
    # matrix-pipe busy fraction while the CU is busy: MFMA busy cycles are counted per SIMD (4 per CU)
    out["mfma_busy_frac"] = {k: round(out["SQ_VALU_MFMA_BUSY_CYCLES"][k]["total"] / (4.0 * out["SQ_BUSY_CU_CYCLES"][k]["total"]), 4)
                             for k in out["SQ_VALU_MFMA_BUSY_CYCLES"] if k in out["SQ_BUSY_CU_CYCLES"]}
if "GRBM_GUI_ACTIVE" in out:
    # shader clock while the kernel ran: GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles
    out["clock_ghz"] = {k: round(v["total"] / 8.0 / (v["avg_us"] * v["launches"]) / 1e3, 3) for k, v in out["GRBM_GUI_ACTIVE"].items()}
import subprocess, datetime, sys
sys.path.insert(0, "$R")
import bench
out["_meta"] = {"tag": "$TAG", "csrc_digest": bench.csrc_digest(), "collected": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
                "command": "rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --single-stream $*"}
json.dump(out, open("$R/gpurun_out/pmc_traffic_$TAG.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k in ("mfma_busy_frac",)}, indent=1))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    print(c, {k: round(v["avg_per_launch"]) for k, v in out.get(c, {}).items()})
PY
