#!/bin/bash
# usage: tools/pmc_wino.sh <shape N,H,W,C,K,pad> <counters...>   (GPU box; one --pmc pass over tools/wino_bench.py; prints the
# per-launch mean of each counter for the implicit-GEMM and the Winograd kernel)
S=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pmcw
timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcw -o p -- python3 $R/tools/wino_bench.py $S > /dev/null 2>/tmp/pmcw.err
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pmcw/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    k="igemm" if "conv_igemm" in n else "wino24" if "conv3x3_wino24" in n else "wino" if "conv3x3_wino" in n else None
    if k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in agg:
    print("$S", k, "dur_us %.1f" % (sum(dur[k])/len(dur[k])/1e3), {c: round(sum(v)/len(v)) for c,v in agg[k].items()})
PY
