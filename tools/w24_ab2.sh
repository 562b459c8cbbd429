#!/bin/bash
# A/B: baseline vs new wino24 on the main NT=2 shapes, interleaved twice
mkdir -p gpurun_out/r04
SH="80,200,200,256,256,1 80,100,100,256,256,1 80,50,50,256,256,1 2560,14,14,256,256,1 80,25,25,512,512,1"
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export SEAM_LIB_PATH=$PWD/seam-match-rcnn_amd/lib/libseam_hip_base.so; else unset SEAM_LIB_PATH; fi
    echo "#### $lib rep $rep"
    python tools/wino_bench.py $SH 2>&1 | grep -v amdgpu.ids | awk '{print $1, $9, $10, $12}'
  done
done
