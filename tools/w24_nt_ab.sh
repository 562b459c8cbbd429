#!/bin/bash
# NT = 1 vs NT = 2 of conv3x3_wino24 per layer shape (same box, interleaved)
SH="80,200,200,64,64,1 80,100,100,128,128,1 80,50,50,256,256,1 80,25,25,512,512,1 80,25,25,256,256,1 2560,14,14,256,256,1 2560,12,12,256,256,0 2560,10,10,256,256,0 80,13,13,256,256,1"
for rep in 1 2; do
  for nt in 1 2; do
    echo "#### NT=$nt rep $rep"
    SEAM_W24_NT=$nt python tools/wino_bench.py $SH 2>&1 | grep -v amdgpu.ids | awk '{print $1, $9, $10}'
  done
done
