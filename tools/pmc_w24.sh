#!/bin/bash
# usage: tools/pmc_w24.sh <shape N,H,W,C,K,pad> <counters...>   (GPU box; one --pmc pass over tools/w24_ab.py; per-launch mean of
# each counter for the F(2x4,3x3) kernel the launcher picked; the library / kernel form follow SEAM_LIB_PATH / SEAM_W24_PC)
S=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp; rm -rf /tmp/pmcw
timeout 180 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcw -o p -- python3 $R/tools/w24_ab.py $S > /dev/null 2>/tmp/pmcw.err
python3 - <<PY
import csv,collections,glob
f=glob.glob("/tmp/pmcw/**/p_counter_collection.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if "conv3x3_wino24" in n:
        k="wino24pc" if "wino24pc" in n else "wino24"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in agg:
    d=sum(dur[k])/len(dur[k])/1e3
    c={c: sum(v)/len(v) for c,v in agg[k].items()}
    extra=""
    if "GRBM_GUI_ACTIVE" in c: extra+=" clock_GHz %.3f" % (c["GRBM_GUI_ACTIVE"]/8/d/1e3)      # (the counter sums the 8 XCDs)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CU_CYCLES" in c: extra+=" mfma_busy %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"]/(4*c["SQ_BUSY_CU_CYCLES"]))
    print("$S", k, "${SEAM_LIB_PATH##*/}", "PC=${SEAM_W24_PC:-1}", "dur_us %.1f" % d, extra, {a: round(b) for a,b in c.items()})
PY
