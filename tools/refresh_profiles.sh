#!/bin/bash
# Regenerate the round's measurement artefacts on the GPU box -> gpurun_out/<tag>_*  (copy the summaries into profiles/).
# usage: tools/refresh_profiles.sh <tag>      (ONE generation per round, on the final tree; the frozen bf16x3 experiment is not refreshed)
TAG=${1:-r01}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp
# counters first: bench.py attaches traffic / pipe-busy / clock from profiles/<tag>[_f16]_pmc_traffic.json when its csrc digest
# matches the tree -- so the lines below carry the counters of THIS generation (copy the two JSONs into profiles/ afterwards)
bash $R/tools/pmc_bench_traffic.sh $TAG > $O/${TAG}_pmc.log 2>&1
cp $O/pmc_traffic_$TAG.json $R/profiles/${TAG}_pmc_traffic.json
bash $R/tools/pmc_bench_traffic.sh ${TAG}_f16 --workload c5 --dtype f16 > $O/${TAG}_f16_pmc.log 2>&1
cp $O/pmc_traffic_${TAG}_f16.json $R/profiles/${TAG}_f16_pmc_traffic.json
python3 $R/bench.py > $O/${TAG}_bench.json 2>$O/${TAG}_bench.err
rm -rf /tmp/prof_$TAG
# kernel stats of the 8-clip steps only: --no-extras keeps the one-clip / full-forward legs (same kernels, other sizes) out of the averages
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o b -- python3 $R/bench.py --no-cpu-baseline --no-extras --single-stream > $O/${TAG}_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof_$TAG/b_kernel_stats.csv $O/${TAG}_bench_kernel_stats_whole_run.csv 2>/dev/null || cp /tmp/prof_$TAG/*/b_kernel_stats.csv $O/${TAG}_bench_kernel_stats_whole_run.csv
# the TIMED STEPS only (bench.py brackets them with two spin_kernel launches): the whole-run file above also counts the bank pass
# (the same kernels on 256x256 shop images), the warm-up and the instrumented roofline / parity legs
python3 $R/tools/kernel_trace_steps.py $(ls /tmp/prof_$TAG/b_kernel_trace.csv /tmp/prof_$TAG/*/b_kernel_trace.csv 2>/dev/null | head -1) > $O/${TAG}_bench_kernel_stats.csv 2> $O/${TAG}_bench_kernel_stats_region.txt
# the same for the config-5 (fp16) step
rm -rf /tmp/prof5_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5_$TAG -o b -- python3 $R/bench.py --dtype f16 --workload c5 --no-cpu-baseline --no-extras --steps 5 --warmup 3 > /dev/null 2>&1
python3 $R/tools/kernel_trace_steps.py $(ls /tmp/prof5_$TAG/b_kernel_trace.csv /tmp/prof5_$TAG/*/b_kernel_trace.csv 2>/dev/null | head -1) > $O/${TAG}_c5_kernel_stats.csv 2> $O/${TAG}_c5_kernel_stats_region.txt
python3 $R/bench.py --workload c3 --no-roofline --cpu-runs 1 > $O/${TAG}_bench_c3.json 2>/dev/null
python3 $R/bench.py --workload c4 --no-roofline --no-cpu-baseline > $O/${TAG}_bench_c4.json 2>/dev/null
python3 $R/bench.py --dtype f16 --no-cpu-baseline > $O/${TAG}_bench_f16.json 2>/dev/null
python3 $R/bench.py --dtype f16 --workload c5 --no-cpu-baseline --steps 5 --warmup 3 > $O/${TAG}_bench_c5_f16.json 2>/dev/null
python3 $R/bench.py --graph --clips 1 --no-cpu-baseline --no-extras > $O/${TAG}_bench_graph.json 2>/dev/null
python3 $R/bench.py --clips 1 --no-cpu-baseline --no-extras > $O/${TAG}_bench_clips1.json 2>/dev/null
python3 $R/tools/heads_bench.py > $O/${TAG}_heads_bench.txt 2>/dev/null
for G in 20000 50000; do
  rm -rf /tmp/pmf_$TAG
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmf_$TAG -o p -- python3 $R/tools/pairmf_bench.py 256 $G > $O/${TAG}_pairmf_${G}.txt 2>/dev/null
  (cat /tmp/pmf_$TAG/p_kernel_stats.csv 2>/dev/null || cat /tmp/pmf_$TAG/*/p_kernel_stats.csv) > $O/${TAG}_pairmf_${G}_kernel_stats.csv
done
python3 $R/tools/train_bench.py 8 10 > $O/${TAG}_train_bench.txt 2>/dev/null
python3 $R/tools/eval_bench.py 200 2000 10 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_eval_bench.txt
python3 $R/tools/eval_bench.py 200 20000 10 2>/dev/null | grep -v amdgpu.ids >> $O/${TAG}_eval_bench.txt
python3 $R/tools/conv_breakdown.py 8 f32 c2 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_breakdown_f32.txt
python3 $R/tools/conv_breakdown.py 8 f16 c5 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_breakdown_c5_f16.txt
python3 $R/bench.py --workload c4 --force-collective --no-roofline --no-cpu-baseline --no-extras > $O/${TAG}_bench_c4_one_rank_rccl.json 2>/dev/null
python3 $R/tools/full_forward_timing.py > $O/${TAG}_full_forward.txt 2>/dev/null
python3 $R/tools/full_forward_timing.py --batch 80 --skip-pack 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_full_forward80.txt
# kernel stats of the drop-in model(images) forward on the 80-frame batch bench.py reports as full_forward_clips_per_s -- rocprofv3
# directly on the script; the one-off weight-packing kernels are dropped from the table and the percentages renormalised
rm -rf /tmp/ff_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ff_$TAG -o f -- python3 $R/tools/full_forward_timing.py --batch 80 --skip-pack > /dev/null 2>&1
(cat /tmp/ff_$TAG/f_kernel_stats.csv 2>/dev/null || cat /tmp/ff_$TAG/*/f_kernel_stats.csv) > /tmp/ff_$TAG.csv
python3 $R/tools/kernel_stats_filter.py /tmp/ff_$TAG.csv pack > $O/${TAG}_full_forward80_kernel_stats.csv 2> $O/${TAG}_full_forward80_dropped.txt
# the probes behind DESIGN section 3: partner-wave experiments and the clock / power trace of bare fp32 MFMAs
hipcc -O3 --offload-arch=gfx950 $R/tools/probes/pc_probe.hip -o /tmp/pc_probe 2>/dev/null && /tmp/pc_probe > $O/${TAG}_pc_probe.txt 2>&1
bash $R/tools/probes/mfma_clock_trace.sh $TAG > /dev/null 2>&1
bash $R/tools/w24pc_ab.sh $TAG > /dev/null 2>&1
# conv3x3_f16pc vs the fp16 implicit GEMM on the config-5 3x3 shapes (whole-batch reference), the big layer on post-ReLU / zero inputs
# (operand statistics move the power-capped clock), and its counters
{ python3 $R/tools/f16pc_ab.py --full; python3 $R/tools/f16pc_ab.py --relu-input 48,192,336,256,256,1; python3 $R/tools/f16pc_ab.py --zeros 48,192,336,256,256,1; } 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_f16pc_ab.txt
bash $R/tools/pmc_f16pc.sh 48,192,336,256,256,1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES > $O/${TAG}_f16pc_pmc.txt 2>&1
python3 $R/tools/pwh_ab.py 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_pwh_ab.txt
python3 $R/tools/pwhpc_ab.py 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_pwhpc_ab.txt
python3 $R/tools/wino_nsplit_ab.py 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_wino_nsplit_ab.txt
python3 $R/tools/f44_noise_probe.py 2>/dev/null | grep -v amdgpu.ids > $O/${TAG}_f44_noise_probe.txt
ls -la $O | grep $TAG
