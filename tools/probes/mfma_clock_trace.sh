#!/bin/bash
# Clock / power of the chip while every SIMD streams bare fp32 (and, second half, fp16) MFMAs (substantiate or drop the "0.85 power ceiling").
# usage (GPU box): tools/probes/mfma_clock_trace.sh <tag>   -> gpurun_out/<tag>_mfma_clock_trace.txt
TAG=${1:-r05}; R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out; mkdir -p $O
hipcc -O3 --offload-arch=gfx950 $R/tools/probes/pc_probe.hip -o /tmp/pc_probe 2>/dev/null || exit 1
{
for kind in long long16; do
for mode in "" zero; do
  echo "#### $kind (long = fp32 32x32x2, long16 = fp16 32x32x16), operands: ${mode:-random}"
  /tmp/pc_probe $kind $mode > /tmp/pc_long.txt &
  P=$!
  sleep 0.3
  for i in $(seq 1 12); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Average Graphics Package Power\|Current Socket Graphics Package Power\|power" | tr -s ' ' | tr '\n' ';'; echo
    sleep 0.15
  done
  wait $P
  cat /tmp/pc_long.txt
done
done
echo "#### idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr -s ' ' | tr '\n' ';'; echo
} > $O/${TAG}_mfma_clock_trace.txt 2>&1
cat $O/${TAG}_mfma_clock_trace.txt
