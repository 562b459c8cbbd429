#!/usr/bin/env python3
"""bench.py -- video-clips/sec of the SEAM Match-RCNN forward hot path on MI355X.

Metric (BASELINE.json): video-clips/sec, clip = 10 frames x 800x800, 32 fixed ROIs/frame,
1000-product gallery (configs[1]).  One *step* = one clip per rank through the whole path
with fixed ROIs (SURVEY.md 8d "full pipeline with fixed ROIs", 3.07 TFLOP/clip):

  frames[10,3,800,800] (resident in HBM) -> normalise/pad -> ResNet-50 body -> FPN -> RPN head
  -> RoIAlign 14x14 on 320 ROIs -> mask head (+sigmoid/label select) -> match_predictor trunk
  (match_features) -> temporal_aggregator Mode A: trunk + 32 sequences x 10 through the non-local
  block + attention pooling -> pairwise match logits vs the 1000-product bank -> top-20.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); every rank runs its own
clip per step (weak scaling) and *owns* 1000/N rows of the product bank, which are all-gathered
on a side stream each step before the match.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N == 1 only): `roofline` for the dominant kernel (conv_igemm<float,128,128>,
bound = fp32 MFMA) from HIP events bracketing every launch of one instrumented step, and
`cpu_baseline` = the CPU oracle timed on the host cores for one clip.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2500.0          # dense fp16/bf16 MFMA peak (not the 2:1-sparse marketing figure)
T, R, G, TOPK = 10, 32, 1000, 20
H = W = 800
# algorithmic FLOP per clip: 10 frames x 240.0 G + 320 ROIs x (2 trunks + mask head)
FLOP_PER_CLIP = 10 * 240.0e9 + 320 * (2 * 0.5338e9 + 1.033e9)


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=T, help="frames of the clip the CPU baseline times")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) and time graph replays")
    ap.add_argument("--workload", choices=("c2", "c5"), default="c2",
                    help="c2 (default, the BASELINE metric's config): 10 x 800x800 frames, 32 ROI/frame, fp32; "
                         "c5: configs[4] per-GPU shape -- 30 x 1080x1920 frames, 64 ROI/frame (use with --dtype f16)")
    ap.add_argument("--clips", type=int, default=8,
                    help="clips batched per step on each GPU: their 10 x clips frames go through the extractor as one batch "
                         "(BASELINE configs[2] batches 8 clips the same way); value counts clips/s.  --clips 1 = one clip "
                         "per step (latency-oriented; 11 % lower throughput: the small pyramid levels cannot fill 256 CUs)")
    ap.add_argument("--dtype", choices=("f32", "f16", "bf16x3"), default="f32",
                    help="f32 (default, the headline: exact fp32 MFMA) | f16 (config-5 style fp16 MFMA, fp32 accumulate; "
                         "extractor + trunks in fp16, descriptors / NLB / match logits fp32)")
    return ap.parse_args()


def build_model(dev):
    import numpy as np
    import torch
    import seam_match_rcnn_amd.synth as synth
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.video_matchrcnn_state(5).items()}
    model = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    model.load_state_dict(sd)
    return model.to(dev).eval(), sd


def main():
    args = parse()
    global T, R, H, W, FLOP_PER_CLIP
    if args.workload == "c5":
        T, R, H, W = 30, 64, 1080, 1920
        # 1080p -> 749x1333 -> padded 768x1344: 387.1 GFLOP/frame
        FLOP_PER_CLIP = T * 387.1e9 + T * R * (2 * 0.5338e9 + 1.033e9)
    import numpy as np
    import torch
    import torch.distributed as dist
    import seam_match_rcnn_amd.synth as synth
    from seam_match_rcnn_amd import _native, ops, retrieval

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _native.lib()                                  # fail loudly if the HIP extension is missing
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    log("building synthetic weights")
    model, sd = build_model(dev)
    if args.dtype == "f16":
        model.set_compute_dtype(torch.float16)
    elif args.dtype == "bf16x3":      # fp32 activations, split-bf16 operands (3 bf16 MFMAs per product), fp32 accumulate
        model.set_compute_dtype(ops.BX3)
    log("weights on device; generating frames")
    ta = model.roi_heads.temporal_aggregator
    B = args.clips
    frames = torch.cat([torch.from_numpy(synth.frames(rank * B + c, T, H, W)) for c in range(B)]).to(dev)   # this rank's clips, resident
    frame_list = list(frames.unbind(0))
    from seam_match_rcnn_amd.models.detection import resized_size
    rh, rw, _ = resized_size(H, W)                                           # ROIs live in the resized frame
    rois_np = synth.fixed_rois(R, rh, rw)
    rois = [torch.from_numpy(rois_np).to(dev) for _ in range(T * B)]
    types = torch.zeros(B * T * R, dtype=torch.int32)                        # all street ROIs (CPU, as the ref passes)
    ids = torch.cat([c * R + torch.arange(R, dtype=torch.int64).repeat(T) for c in range(B)])   # sequence id = (clip, ROI slot)
    lo, hi = retrieval.shard_range(G, rank, world)
    bank_shard = torch.from_numpy(synth.gallery(7, G)[lo:hi]).to(dev)        # this rank's product descriptors
    side = torch.cuda.Stream(device=dev) if world > 1 else None

    def step():
        pending = retrieval.gather_product_bank(bank_shard, G, side_stream=side)   # overlaps the extractor
        res, feats, rpn = model.forward_fixed_rois(frame_list, rois, run_rpn_head=True)
        roi_features = torch.cat([r["roi_features"] for r in res])          # [320,256,14,14] (reference layout)
        out = ta(roi_features, types, ids)                                    # Mode A: trunk + NLB + attention pool
        x5, idx, score = retrieval.match_sequences(ta, out[0], pending.wait(), TOPK)
        return res, out, x5, idx, score

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run = step
    if args.graph:
        # the whole step (about 130 launches of our kernels + torch glue) as ONE hipGraph: no allocation, no
        # host sync and -- with the cached packing plan -- no H2D copy happens inside a steady-state step
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(cap):
            step(); step()
        torch.cuda.current_stream().wait_stream(cap)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            graph_out = step()
        run = graph.replay
    log("warmup")
    with torch.no_grad():
        for _ in range(args.warmup):
            run()
            torch.cuda.synchronize()
            log("warmup step done")
        sync_all()
        log("timing")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        sync_all()
        elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ms_per_step = 1e3 * elapsed / args.steps
    log(f"timed: {ms_per_step:.2f} ms/step")
    value = world * B * args.steps / elapsed                                  # whole-job clips/s

    roofline = None
    if rank == 0 and world == 1 and not args.no_roofline:
        with torch.no_grad():
            ops.CONV_TRACE = []
            step()
            torch.cuda.synchronize()
            trace, ops.CONV_TRACE = ops.CONV_TRACE, None
        per = {}
        for variant, flops, e0, e1, _shape, nbytes in trace:
            a = per.setdefault(variant, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += nbytes
        dom = max(per, key=lambda k: per[k][2])
        n, fl, sec, alg_bytes = per[dom]
        achieved = fl / sec / 1e12
        peak = FP32_MFMA_PEAK_TFLOPS if args.dtype == "f32" else F16_MFMA_PEAK_TFLOPS
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    **({"mfma_passes_per_product": 3, "issued_frac": round(3 * achieved / peak, 4)} if args.dtype == "bf16x3" else {}),
                    **({"algorithm": "Winograd F(2x4,3x3): 24 MFMA multiplies per 2x4 output tile and channel pair instead of 72, "
                                     "so `achieved` (ALGORITHMIC FLOP/s, 2*M*K*9*C per launch) can exceed the MFMA peak",
                        "mfma_issued_frac": round(achieved / 3.0 / peak, 4)} if dom == "conv3x3_wino24" else
                       {"algorithm": "Winograd F(2x2,3x3): 16 MFMA multiplies per 2x2 output tile and channel pair instead of 36, "
                                     "so `achieved` (ALGORITHMIC FLOP/s, 2*M*K*9*C per launch) can exceed the MFMA peak",
                        "mfma_issued_frac": round(achieved / 2.25 / peak, 4)} if dom.startswith("conv3x3_wino") else {}),
                    "traffic": pmc_traffic(dom) if args.dtype == "f32" else None,
                    "mfma_busy_frac_pmc": pmc_mfma_busy(dom) if args.dtype == "f32" else None,
                    "algorithmic_bytes_per_launch": round(alg_bytes / n),
                    "launches_per_step": n, "avg_launch_us": round(1e6 * sec / n, 2),
                    "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
                    "conv_ms_per_step": round(1e3 * sum(v[2] for v in per.values()), 3),
                    "other_variants": {k: {"launches": v[0], "TFLOP/s": round(v[1] / v[2] / 1e12, 2)}
                                       for k, v in per.items() if k != dom}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "c2":
        log("cpu baseline")
        cpu = cpu_baseline(sd, frames[:T].cpu(), rois_np, args.cpu_frames)
        log("cpu baseline done")

    if rank == 0:
        metric = ("video-clips/sec (10f x 800^2, 32 ROI/f, 1k gallery)" if args.workload == "c2"
                  else "video-clips/sec (30f x 1080p, 64 ROI/f, 1k gallery)")
        line = {"metric": metric, "value": round(value, 4),
                "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": {"workload": ("configs[1]" if args.workload == "c2" else "configs[4] per-GPU shape") +
                                       f" full pipeline, fixed ROIs: {T} frames {H}x{W} -> ResNet-50-FPN + RPN head "
                                       f"-> RoIAlign 14x14 ({R} ROI/frame) -> mask head -> match trunk x2 -> SEAM NLB + "
                                       f"attention pool ({R} seq x {T}) -> pair logits vs {G}-product bank -> top-{TOPK}",
                           "clips_per_step_per_gpu": B, "hip_graph": bool(args.graph), "frames": T, "rois_per_frame": R, "gallery": G, "topk": TOPK,
                           "algorithmic_tflop_per_clip": round(FLOP_PER_CLIP / 1e12, 3),
                           "parallelism": f"dp{world} (clips sharded; product bank all-gathered over RCCL)"
                           if world > 1 else "single GPU"},
                "pipeline_tflops": round(FLOP_PER_CLIP * value / 1e12, 2)}
        if roofline is not None:
            line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json,
    produced by tools/pmc_bench_traffic.sh on this same bench command): FETCH_SIZE and WRITE_SIZE are
    collected in separate passes, reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide
    coalesced reads, so the read side is doubled (MI355X_MICROARCH.md, HBM section).  None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        return round((2.0 * d["FETCH_SIZE"][kernel]["avg_per_launch"] + d["WRITE_SIZE"][kernel]["avg_per_launch"]) * 1024)
    except (KeyError, ValueError, OSError):
        return None


def pmc_mfma_busy(kernel):
    """Matrix-pipe busy fraction of `kernel` from the same committed PMC passes:
    SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES).  None when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    try:
        return json.load(open(files[-1]))["mfma_busy_frac"][kernel]
    except (IndexError, KeyError, ValueError, OSError):
        return None


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sd, frames_cpu, rois_np, n_frames):
    """The CPU oracle (kind = "port": torch-CPU restatement, the same ATen kernels the reference
    dispatches) timed on the host cores for ONE clip (bounded sample), frame by frame."""
    import torch
    from oracle import detection as OD
    from oracle import heads as OH
    from oracle import model as OM
    import seam_match_rcnn_amd.synth as synth

    cores = usable_cores()
    torch.set_num_threads(cores)
    rois = torch.from_numpy(rois_np)
    n_frames = max(1, min(n_frames, frames_cpu.shape[0]))
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    tap = OM.sub(sd, "roi_heads.temporal_aggregator.")
    bank = torch.from_numpy(synth.gallery(7, G))
    with torch.no_grad():
        t0 = time.perf_counter()
        roi_feats = []
        for f in range(n_frames):
            batch, sizes = OD.transform([frames_cpu[f]])
            feats = OD.fpn(OD.resnet50_body(batch, sd), sd)
            OD.rpn_head(list(feats.values()), sd)
            rf = OD.multiscale_roi_align([feats[k] for k in "0123"], [rois], sizes, 14)
            OD.maskrcnn_inference(OD.mask_head(rf, sd), [torch.ones(len(rois), dtype=torch.int64)])
            OH.match_trunk(rf, mp)
            roi_feats.append(rf)
        x = torch.cat(roi_feats)
        ids = torch.arange(R, dtype=torch.int64).repeat(n_frames)
        out = OH.temporal_aggregation_forward(x, torch.zeros(len(ids), dtype=torch.int32), ids, tap)
        x5 = OH.pair_logits(out[0], bank, tap["last.weight"], tap["last.bias"])
        OH.rank_topk(x5, TOPK)
        sec = time.perf_counter() - t0
    clip_sec = sec * (T / n_frames)      # per-frame work dominates; scale when fewer frames were timed
    return {"value": round(1.0 / clip_sec, 5), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"1 clip, {n_frames} of {T} frames timed ({sec:.1f} s), same stages/inputs as the GPU step; "
                      f"torch {torch.__version__} CPU fp32, {cores} threads"}


if __name__ == "__main__":
    main()
