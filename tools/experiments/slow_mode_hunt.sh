#!/bin/bash
# Hunt the two-stream slow mode of the fp16 path: repeat the config-5 line with two body streams under rocprofv3's kernel
# trace until one run is slow (> 180 ms) and one is fast; keep the timed-step kernel statistics of both.   (GPU box)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
have_fast=0; have_slow=0
for i in $(seq 1 ${1:-12}); do
  rm -rf /tmp/hunt
  rocprofv3 --kernel-trace --output-format csv -d /tmp/hunt -o b -- python3 $R/bench.py --dtype f16 --workload c5 --two-streams --no-cpu-baseline --no-roofline --no-extras --steps 5 --warmup ${WARMUP:-1} > /tmp/hunt.json 2>/dev/null
  ms=$(python3 -c "import json; print(json.loads(open('/tmp/hunt.json').read().strip().splitlines()[-1])['ms_per_step'])")
  echo "run $i: $ms ms  device allocs in the timed region: $(python3 -c "import json; print(json.loads(open('/tmp/hunt.json').read().strip().splitlines()[-1])['hbm_gb']['device_allocs_in_timed_region'])")"
  tr=$(ls /tmp/hunt/b_kernel_trace.csv /tmp/hunt/*/b_kernel_trace.csv 2>/dev/null | head -1)
  if python3 -c "import sys; sys.exit(0 if $ms > 180 else 1)"; then
    if [ $have_slow = 0 ]; then python3 $R/tools/kernel_trace_steps.py $tr > $O/hunt_slow_kernel_stats.csv 2> $O/hunt_slow_region.txt; cp $tr $O/hunt_slow_trace.csv; echo "$ms" > $O/hunt_slow_ms.txt; have_slow=1; fi
  else
    if [ $have_fast = 0 ]; then python3 $R/tools/kernel_trace_steps.py $tr > $O/hunt_fast_kernel_stats.csv 2> $O/hunt_fast_region.txt; echo "$ms" > $O/hunt_fast_ms.txt; have_fast=1; fi
  fi
  if [ $have_fast = 1 ] && [ $have_slow = 1 ]; then break; fi
done
echo "fast $have_fast slow $have_slow"
