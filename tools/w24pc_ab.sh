#!/bin/bash
# conv3x3_wino24pc (producer / consumer, round 5) vs conv3x3_wino24<2> (round 4) on the bench's layer shapes, same box, interleaved.
# usage (GPU box): tools/w24pc_ab.sh <tag> [shapes...]   -> gpurun_out/<tag>_w24pc_ab.txt
TAG=${1:-r05}; shift
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out; mkdir -p $O
{
for rep in 1 2; do
  SEAM_W24_PC=0 timeout 600 python3 $R/tools/w24_ab.py "$@" 2>&1 | grep -v amdgpu.ids
  SEAM_W24_PERSIST=0 timeout 600 python3 $R/tools/w24_ab.py "$@" 2>&1 | grep -v amdgpu.ids | sed 's/wino24pc/wino24pc-1tile/'
  timeout 600 python3 $R/tools/w24_ab.py "$@" 2>&1 | grep -v amdgpu.ids
done
} > $O/${TAG}_w24pc_ab.txt
cat $O/${TAG}_w24pc_ab.txt
