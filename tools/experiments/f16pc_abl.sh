#!/bin/bash
# Variant builds of conv3x3_f16pc: `build name:flags ...` in the dev container, `run [shapes...]` on the GPU box.
cd "$(dirname "$0")/../.." || exit 1
C=seam-match-rcnn_amd/csrc; L=tools/experiments/_libf
if [ "$1" = build ]; then
  shift
  rm -rf $L; mkdir -p $L
  for v in "$@"; do
    name=${v%%:*}; flags=$(echo "${v#*:}" | sed 's/,/ -D/g; s/^/-D/')
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -DSEAM_DEV_BUILD $flags -c $C/seam_f16pc.hip -o /tmp/f16pc_$name.o || exit 1
    hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v seam_f16pc.o) /tmp/f16pc_$name.o -o $L/libseam_$name.so || exit 1
  done
else
  shift
  echo "== shipped"; python3 tools/f16pc_ab.py "$@" 2>/dev/null | tail -n +2
  for f in $(ls $L/libseam_*.so | sort -V); do echo "== $f"; SEAM_LIB_PATH=$PWD/$f python3 tools/f16pc_ab.py "$@" 2>/dev/null | tail -n +2; done
fi
