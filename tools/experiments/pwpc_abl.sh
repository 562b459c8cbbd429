#!/bin/bash
# Variant builds of conv1x1_pc: `build name:flags ...` in the dev container (e.g. trace:SEAM_PWPC_TRACE=8 abl4:SEAM_PWPC_TRACE=8,SEAM_PWPC_ABL=4),
# `run [shapes...]` on the GPU box (tools/pwpc_ab.py under each variant; trace builds print their stamps on stderr as "TR wave ...").
cd "$(dirname "$0")/../.." || exit 1
C=seam-match-rcnn_amd/csrc; L=tools/experiments/_libp
if [ "$1" = build ]; then
  shift
  rm -rf $L; mkdir -p $L
  for v in "$@"; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | sed 's/,/ -D/g; s/^/-D/')
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DSEAM_DEV_BUILD $flags -c $C/seam_pwpc.hip -o /tmp/pwpc_$name.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v seam_pwpc.o) /tmp/pwpc_$name.o -o $L/libseam_$name.so || exit 1
  done
else
  shift
  for f in $(ls $L/libseam_*.so | sort -V); do echo "== $f"; SEAM_LIB_PATH=$PWD/$f python3 tools/pwpc_ab.py "$@" 2>&1 | grep -v amdgpu.ids; done
fi
