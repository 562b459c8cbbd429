#!/bin/bash
# Ablation builds of the F(2x4,3x3) Winograd kernel (SEAM_W24_ABL bits: 1 no patch loads / LDS stores, 2 no weight loads,
# 4 no barrier, 8 no transforms, 16 no epilogue; operands keep real data).  `build [bits...]` in the dev container, `run [shapes...]` on the GPU box.
cd "$(dirname "$0")/../.." || exit 1
C=seam-match-rcnn_amd/csrc; L=tools/experiments/_lib     # travels with gpurun (*.so is git-ignored); delete after the experiment
if [ "$1" = build ]; then
  shift
  mkdir -p $L
  for a in ${@:-1 2 3 4 8 15}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DSEAM_W24_ABL=$a -c $C/seam_wino24.hip -o /tmp/w24_abl$a.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v seam_wino24.o) /tmp/w24_abl$a.o -o $L/libseam_abl$a.so
  done
else
  shift
  python tools/w24_ab.py "$@" 2>/dev/null | tail -n +3
  for f in $(ls $L/libseam_abl*.so | sort -V); do echo "== $f"; SEAM_LIB_PATH=$PWD/$f python tools/w24_ab.py "$@" 2>/dev/null | tail -n +3; done
fi
