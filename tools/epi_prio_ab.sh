#!/bin/bash
# Same-box A/B of the igemm epilogue at raised wave priority (SEAM_EPI_PRIO=0 / 1): per-layer conv_bench + the bench step (GPU box)
R=$GRAFT_REPO_ROOT
SHAPES="80,50,50,256,1024,1,1,0,1 80,50,50,1024,256,1,1,0,0 80,100,100,128,512,1,1,0,1 80,200,200,64,256,1,1,0,1 80,200,200,256,64,1,1,0,0 80,100,100,512,128,1,1,0,0 80,25,25,512,2048,1,1,0,1 80,25,25,2048,512,1,1,0,0 80,200,200,256,256,1,1,0,0 80,100,100,512,256,1,1,0,0 80,200,200,256,128,1,1,0,0 80,50,50,1024,512,1,1,0,0 80,100,100,256,256,3,2,1,0 2560,14,14,256,1024,1,1,0,0 80,400,400,12,64,4,1,2,0"
for rep in 1 2; do
for P in 0 1; do echo "=== SEAM_EPI_PRIO=$P (rep $rep)"; SEAM_EPI_PRIO=$P python3 $R/tools/conv_bench.py $SHAPES 2>/dev/null | awk 'NR>1{printf "%s %s | ", $1, $2} END{print ""}'; done
done
for rep in 1 2 3; do
for P in 0 1; do echo "bench SEAM_EPI_PRIO=$P: $(SEAM_EPI_PRIO=$P python3 $R/bench.py --no-cpu-baseline --no-roofline --no-extras --steps 10 --warmup 3 2>/dev/null | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"; done
done
