#!/bin/bash
# same-box A/B of two builds of the library on the pointwise layers: lib/libseam_hip_base.so vs lib/libseam_hip.so
for rep in 1 2; do
  for lib in base new; do
    if [ $lib = base ]; then export SEAM_LIB_PATH=$PWD/seam-match-rcnn_amd/lib/libseam_hip_base.so; else unset SEAM_LIB_PATH; fi
    echo "#### $lib rep $rep"
    python tools/pw_bench.py "$@" 2>&1 | grep -v amdgpu.ids | awk '{print $1, $4, $5, $7}'
  done
done
