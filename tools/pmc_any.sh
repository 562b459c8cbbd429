#!/bin/bash
# usage: tools/pmc_any.sh "<script.py args>" <kernel-name-substring> <counters...>   (GPU box; one --pmc pass, mean per launch)
CMD=$1; K=$2; shift 2
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pmcy
timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcy -o p -- python3 $R/$CMD > /dev/null 2>/tmp/pmcy.err
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("/tmp/pmcy/p_counter_collection.csv")))
agg=collections.defaultdict(list); dur=[]
for r in rows:
    if "$K" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
print("kernel [$K] launches %d dur_us %.1f" % (len(dur) // max(1, len(agg)), sum(dur)/max(1,len(dur))/1e3), {k: round(sum(v)/len(v)) for k,v in agg.items()})
PY
