#!/bin/bash
# usage: tools/pmc_f16pc.sh <shape N,H,W,C,K,pad> <counters...>   (GPU box; one --pmc pass over tools/f16pc_ab.py)
S=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pmcf
timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcf -o p -- python3 $R/tools/f16pc_ab.py $S > /dev/null 2>/tmp/pmcf.err
python3 - <<PY
import csv,collections,glob
f=glob.glob("/tmp/pmcf/**/p_counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    k="f16pc" if "f16pc" in n and "pack" not in n else "igemm_f16" if "conv_igemmIDF16" in n else None
    if k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in agg:
    d=sum(dur[k])/len(dur[k])/1e3
    c={c: sum(v)/len(v) for c,v in agg[k].items()}
    extra=""
    if "GRBM_GUI_ACTIVE" in c: extra+=" clock_GHz %.3f" % (c["GRBM_GUI_ACTIVE"]/8/d/1e3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c: extra+=" mfma_busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"]/(4*c["SQ_BUSY_CU_CYCLES"]))
    print("$S", k, "${SEAM_LIB_PATH##*/}", "dur_us %.1f" % d, extra, {a: round(b) for a,b in c.items()})
PY
