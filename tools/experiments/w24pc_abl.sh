#!/bin/bash
# Variant builds of conv3x3_wino24pc.  `build name:flags ...` in the dev container (flags: comma-separated -D macros, e.g.
# abl1:SEAM_W24PC_ABL=1  ring6:SEAM_W24PC_RING=6), `run [shapes...]` on the GPU box (base library first, then every variant).
cd "$(dirname "$0")/../.." || exit 1
C=seam-match-rcnn_amd/csrc; L=tools/experiments/_lib     # travels with gpurun (*.so is git-ignored); delete after the experiment
if [ "$1" = build ]; then
  shift
  rm -rf $L; mkdir -p $L
  for v in "$@"; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | sed 's/,/ -D/g; s/^/-D/')
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DSEAM_DEV_BUILD $flags -c $C/seam_wino24.hip -o /tmp/w24pc_$name.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "wino24pc" | grep "VGPRs:\|ScratchSize\|Occupancy" | tr '\n' ' '; echo " <- $name"
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v seam_wino24.o) /tmp/w24pc_$name.o -o $L/libseam_$name.so || exit 1
  done
else
  shift
  echo "== shipped"; python3 tools/w24_ab.py "$@" 2>/dev/null | tail -n +3
  for f in $(ls $L/libseam_*.so | sort -V); do echo "== $f"; SEAM_LIB_PATH=$PWD/$f python3 tools/w24_ab.py "$@" 2>/dev/null | tail -n +3; done
  echo "== shipped (again)"; python3 tools/w24_ab.py "$@" 2>/dev/null | tail -n +3
fi
