#!/bin/bash
# Same-box A/B of ONE source file against an earlier version of it.
#   build <file.hip> [git-ref]   (dev container)  -> tools/experiments/_lib/libseam_base.so = the current objects with <file.hip>
#                                                    taken from <git-ref> (default HEAD)
#   run <tool.py> [args...]      (GPU box)        -> base, new, base, new (interleaved; boxes of the pool differ by 2-6 %)
# _lib/ travels with gpurun (*.so is git-ignored); delete it after the experiment.
cd "$(dirname "$0")/../.." || exit 1
C=seam-match-rcnn_amd/csrc; L=tools/experiments/_lib
if [ "$1" = build ]; then
  f=$2; ref=${3:-HEAD}; mkdir -p $L
  git show $ref:$C/$f > /tmp/ab_base_$f || exit 1
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C -c /tmp/ab_base_$f -o /tmp/ab_base_${f%.hip}.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/${f%.hip}.o") /tmp/ab_base_${f%.hip}.o -o $L/libseam_base.so || exit 1
  echo "built $L/libseam_base.so ($f @ $ref)"
else
  shift; tool=$1; shift
  for rep in 1 2; do
    echo "== base"; SEAM_LIB_PATH=$PWD/$L/libseam_base.so python3 $tool "$@" 2>&1 | grep -v amdgpu.ids
    echo "== new";  python3 $tool "$@" 2>&1 | grep -v amdgpu.ids
  done
fi
