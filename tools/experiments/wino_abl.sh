#!/bin/bash
# Ablation builds of the Winograd kernel (SEAM_WINO_ABL bits: 1 no patch loads, 2 no weight loads, 4 no barrier, 8 no LDS
# reads, 16 no LDS stores).  Run `build` in the dev container (hipcc cross-compiles), `run` on the GPU box.
cd "$(dirname "$0")/../.." || exit 1
C=seam-match-rcnn_amd/csrc; L=seam-match-rcnn_amd/lib/abl
if [ "$1" = build ]; then
  mkdir -p $L
  for a in ${2:-1 2 3 4 8 16 31}; do
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DSEAM_WINO_ABL=$a -c $C/seam_wino.hip -o /tmp/wino_abl$a.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v seam_wino.o) /tmp/wino_abl$a.o -o $L/libseam_abl$a.so
  done
else
  shift
  python tools/wino_bench.py "$@" | tail -n +2
  for f in $L/libseam_abl*.so; do echo "== $f"; SEAM_LIB_PATH=$PWD/$f python tools/wino_bench.py "$@" 2>/dev/null | tail -n +2; done
fi
