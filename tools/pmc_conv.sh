#!/bin/bash
# usage: tools/pmc_conv.sh "<conv_bench flags>" <shape> <counters...>   (GPU box; prints mean counter values of the conv kernel)
F=$1; S=$2; shift 2
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pmcx
timeout 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcx -o p -- python3 $R/tools/conv_bench.py $F $S > /dev/null 2>/tmp/pmcx.err
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pmcx/p_counter_collection.csv")))
agg=collections.defaultdict(list); dur=[]
for r in rows:
    if "conv_igemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("flags [$F] dur_us %.1f" % (sum(dur)/len(dur)/1e3), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
