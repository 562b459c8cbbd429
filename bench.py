#!/usr/bin/env python3
"""bench.py -- video-clips/sec of the SEAM Match-RCNN forward hot path on MI355X.

Metric (BASELINE.json): video-clips/sec, clip = 10 frames x 800x800, 32 fixed ROIs/frame,
1000-product gallery (configs[1]).  One *step* = ``--clips`` (8) clips per rank through the whole path
with fixed ROIs (SURVEY.md 8d "full pipeline with fixed ROIs", 3.07 TFLOP/clip):

  frames[10,3,800,800] (resident in HBM) -> normalise/pad -> ResNet-50 body -> FPN -> RPN head
  -> RoIAlign 14x14 on 320 ROIs -> mask head (+sigmoid/label select) -> match_predictor trunk
  (match_features) -> temporal_aggregator Mode A: trunk + 32 sequences x 10 through the non-local
  block + attention pooling -> pairwise match logits vs the 1000-product bank -> top-20.

Workloads (``--workload``): c2 = configs[1] (the metric's config, default) | c3 = configs[2] (8 clips per step, 20 000-product
gallery, ranking without the [S,G,2] logits in HBM) | c4 = configs[3] per-rank view (8 clips per rank per step, 50 000-product bank
whose shards are all-gathered over RCCL every step) | c5 = configs[4] per-GPU shape (30 x 1080p frames, 64 ROI/frame; --dtype f16).

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); every rank runs its own clips per step (weak scaling)
and *owns* G/N rows of the product bank, which are all-gathered on a side stream each step before the match (the achieved
all-gather rate per xGMI link is reported under ``allgather``).  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N == 1 only):
  roofline      the dominant kernel of one instrumented step (HIP events bracketing every conv launch on the launch stream);
                ``achieved`` = MFMA FLOP/s the kernel ISSUES for its algorithmic work (algorithmic / 3 for Winograd F(2x4,3x3),
                / 2.25 for F(2x2,3x3)), ``frac`` = achieved / the dense MFMA peak of the dtype (always <= 1); the
                direct-convolution-equivalent rate is kept as ``algorithmic_tflops``.
  cpu_baseline  the CPU oracle (torch-CPU restatement) on the host cores for one clip: 1 warm-up, min of ``--cpu-runs`` (3).
  parity        the timed GPU step's outputs for clip 0 (roi_features, x3_1b, match logits, top-20) against the outputs the
                cpu_baseline leg computed from the same inputs; the run FAILS when they disagree by more than 1e-3.
  extras        ``value_clips1`` (one clip per step) and ``full_forward_ms_per_clip`` (the drop-in ``model(images)`` forward
                with RPN proposals, box head, NMS and mask paste on the same 10 frames).  ``--no-extras`` skips them (used for
                the rocprofv3 kernel-stats run, so that the per-kernel averages cover the 8-clip steps only).
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
XGMI_LINK_GBS = 153.0                  # per-link, per-direction (MI355X_MICROARCH.md / BASELINE.md section 2)
TOPK = 20
WORKLOADS = {
    # name: frames, ROI/frame, H, W, gallery, BASELINE.json config it stands for, how the match is ranked
    "c2": dict(T=10, R=32, H=800, W=800, G=1000, cfg="configs[1]", rank="logits"),
    "c3": dict(T=10, R=32, H=800, W=800, G=20000, cfg="configs[2]", rank="topk"),
    "c4": dict(T=10, R=32, H=800, W=800, G=50000, cfg="configs[3] per-rank view (64 clips over 8 GPUs = 8 clips per rank per step)",
               rank="topk"),
    "c5": dict(T=30, R=64, H=1080, W=1920, G=1000, cfg="configs[4] per-GPU shape", rank="logits"),
}
PARITY_TOL = 1e-3                      # north_star: within 1e-3 relative fp32 (16-bit paths: reported, looser gate)

_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the value_clips1 / full_forward_ms_per_clip legs")
    ap.add_argument("--force-collective", action="store_true",
                    help="test hook: initialise the process group and issue the bank all-gather (side stream, timed) even with one rank, "
                         "so that the N > 1 code path can be exercised on a single GPU")
    ap.add_argument("--stub-step", action="store_true",
                    help="test hook (CPU, gloo): replace the GPU step by the bank all-gather alone, so that the launch / barrier / "
                         "rank-reduction / one-line logic of the N > 1 path can run on a box without GPUs; the line says data=stub "
                         "and is not a measurement")
    ap.add_argument("--single-stream", action="store_true",
                    help="every launch on one HIP stream (the default runs the ResNet body on two streams and the small pyramid levels on "
                         "a side stream: bit-identical results, ~1.4 %% faster, but concurrent kernels stretch each other's durations -- "
                         "per-kernel profiles (rocprofv3 --stats, the PMC passes) are taken with this flag; the roofline leg always "
                         "instruments a single-stream step)")
    ap.add_argument("--two-streams", action="store_true",
                    help="fp16 path only (--dtype f16): run the body on two HIP streams like the fp32 path (1.5 %% faster; 91-106 GB of "
                         "allocator pools instead of 48).  The fp16 default is ONE stream.  The 205-220 ms lines once seen with two streams "
                         "were hipMalloc stalls INSIDE the timed region (side-stream tensors were record_stream-ed, an unpaced host ran the "
                         "whole timed region ahead, the caching allocator grew step by step; 100+ ms per call on a box whose HBM had not been "
                         "touched since boot) -- the model no longer records streams and the timed loop keeps the host two steps ahead at "
                         "most (DESIGN 3.4); two streams still want --warmup >= 3 (their own pools)")
    ap.add_argument("--cpu-frames", type=int, default=None, help="frames of the clip the CPU baseline times (default: all)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed CPU runs after one warm-up; the minimum is reported")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step into a HIP graph (torch.cuda.CUDAGraph) and time graph replays")
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="c2",
                    help="c2 (default, the BASELINE metric's config): 10 x 800x800 frames, 32 ROI/frame, 1000 products, fp32; "
                         "c3: configs[2], 20 000 products; c4: configs[3] per-rank view, 50 000 products all-gathered per step; "
                         "c5: configs[4] per-GPU shape -- 30 x 1080x1920 frames, 64 ROI/frame (use with --dtype f16)")
    ap.add_argument("--clips", type=int, default=8,
                    help="clips batched per step on each GPU: their frames go through the extractor as one batch "
                         "(BASELINE configs[2] batches 8 clips the same way); value counts clips/s.  --clips 1 = one clip "
                         "per step (latency-oriented; ~11 %% lower throughput: the small pyramid levels cannot fill 256 CUs)")
    ap.add_argument("--dtype", choices=("f32", "f16", "bf16x3"), default="f32",
                    help="f32 (default, the headline: exact fp32 MFMA) | f16 (config-5 style fp16 MFMA, fp32 accumulate; "
                         "extractor + trunks in fp16, descriptors / NLB / match logits fp32) | bf16x3 (split-bf16, opt-in)")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_command(argv, n, port):
    """The child command of the self-launch: N ranks of this file under torch.distributed.run on the loopback (the same
    command line the driver uses for N > 1; the reference's own launch idiom is an external `python -m torch.distributed.launch
    --nproc_per_node=N train_movingfashion.py ...`, ref README.md:98-110)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_self_launch(gpus, env):
    """`python bench.py --gpus N` with N > 1 outside a launcher starts its own ranks.  SEAM_BENCH_SELF_LAUNCH=1 takes the same
    branch with N == 1 (so that a 1-GPU box can test it)."""
    if "WORLD_SIZE" in env or "RANK" in env:
        return False
    return gpus > 1 or env.get("SEAM_BENCH_SELF_LAUNCH") == "1"


def self_launch(args, argv=None, popen=None):
    """Start N ranks as a CHILD `python -m torch.distributed.run ... bench.py <same flags>`, relay rank 0's single JSON line on this
    process's stdout, everything else on stderr, and return the child's exit code.  ALWAYS a subprocess, never an exec: counting the
    GPUs below may already have initialised HIP in this parent (on ROCm builds without amdsmi torch.cuda.device_count() falls back to
    hipGetDeviceCount), and replacing a process that has touched the GPU takes the machine down on this pool.  `popen` is a test seam."""
    import subprocess
    argv = sys.argv[1:] if argv is None else argv
    if popen is None:
        import torch
        have = args.gpus if args.stub_step else torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible on this node", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // max(1, args.gpus))))
    env.pop("SEAM_BENCH_SELF_LAUNCH", None)
    cmd = launch_command(argv, args.gpus, free_port())
    log("self-launch: " + " ".join(cmd))
    p = (popen or subprocess.Popen)(cmd, stdout=subprocess.PIPE, env=env, text=True, cwd=ROOT)
    lines = 0
    for raw in p.stdout:
        s = raw.strip()
        is_line = False
        if s.startswith("{") and s.endswith("}"):
            try:
                is_line = "metric" in json.loads(s)
            except ValueError:
                is_line = False
        if is_line and lines == 0:
            lines += 1
            sys.stdout.write(s + "\n")
            sys.stdout.flush()
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = p.wait()
    if rc == 0 and lines != 1:
        print("bench.py: the launched ranks printed no result line", file=sys.stderr)
        return 1
    return rc


def build_model(dev):
    import numpy as np
    import torch
    import seam_match_rcnn_amd.synth as synth
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.video_matchrcnn_state(5).items()}
    model = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    model.load_state_dict(sd)
    return model.to(dev).eval(), sd


SHOP_HW = 256          # side of the synthetic shop images (one product each); the extractor runs them at this size
SHOP_BATCH = 128       # shop images per extractor call; batches are aligned on GLOBAL product ids, so a row never depends on the rank count


def shop_images(batch_index, dev):
    """The synthetic shop images of products [batch_index * SHOP_BATCH, +SHOP_BATCH): a function of the batch index alone
    (device Philox stream), so every rank that computes a product computes it from the same pixels."""
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EA3B00C + batch_index)
    # a smooth random colour field per product (8x8 control points, bilinear) + pixel noise: products differ at every scale
    coarse = torch.rand((SHOP_BATCH, 3, 8, 8), device=dev, generator=g)
    noise = torch.rand((SHOP_BATCH, 3, SHOP_HW, SHOP_HW), device=dev, generator=g)
    img = torch.nn.functional.interpolate(coarse, size=(SHOP_HW, SHOP_HW), mode="bilinear", align_corners=False)
    return (0.8 * img + 0.2 * noise).contiguous()


def shop_box():
    import torch
    return torch.tensor([[0.1 * SHOP_HW, 0.1 * SHOP_HW, 0.9 * SHOP_HW, 0.9 * SHOP_HW]], dtype=torch.float32)


def compute_bank_rows(model, ta, lo, hi, dev):
    """Rows [lo, hi) of the product bank through the SHOP-SIDE PATH of the reference's evaluation
    (/root/reference/evaluate_movingfashion.py:31-47: every product's shop image goes through the extractor once, the product
    box's RoI features through the aggregator as a shop item, `temporal_aggregator(roi_features[best][None], [1], [0])[1]`):
    synthetic shop images -> forward_fixed_rois (ResNet-50-FPN + RoIAlign on the product box) -> the aggregator's shop
    descriptor (types == 1: the match trunk's 256-vector).  Runs once, before anything is timed."""
    import torch
    saved = model.transform.min_size, model.transform.max_size
    model.transform.min_size, model.transform.max_size = SHOP_HW, SHOP_HW
    rows = []
    try:
        with torch.no_grad():
            box = shop_box().to(dev)
            for bi in range(lo // SHOP_BATCH, (hi + SHOP_BATCH - 1) // SHOP_BATCH):
                imgs = shop_images(bi, dev)
                res, _, _ = model.forward_fixed_rois(list(imgs.unbind(0)), [box] * SHOP_BATCH, run_rpn_head=False)
                rf = torch.cat([r["roi_features"] for r in res])
                n = rf.shape[0]
                desc = ta(rf, torch.ones(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int64))[1]      # x3_2: shop descriptors
                p0 = bi * SHOP_BATCH
                rows.append(desc[max(lo, p0) - p0:min(hi, p0 + SHOP_BATCH) - p0].clone())
    finally:
        model.transform.min_size, model.transform.max_size = saved
    return torch.cat(rows) if rows else torch.empty((0, 256), device=dev)


def flop_per_clip(wl):
    """Algorithmic FLOP per clip: frames x (body + FPN + RPN head) + ROIs x (2 trunks + mask head)."""
    per_frame = 240.0e9 if (wl["H"], wl["W"]) == (800, 800) else 387.1e9      # 1080p -> 749x1333 -> padded 768x1344
    return wl["T"] * per_frame + wl["T"] * wl["R"] * (2 * 0.5338e9 + 1.033e9)


def main():
    args = parse()
    wl = WORKLOADS[args.workload]
    T, R, H, W, G = wl["T"], wl["R"], wl["H"], wl["W"], wl["G"]
    FLOP_PER_CLIP = flop_per_clip(wl)
    if needs_self_launch(args.gpus, os.environ):       # before any HIP call in this process
        raise SystemExit(self_launch(args))
    import torch
    import torch.distributed as dist
    import seam_match_rcnn_amd.synth as synth
    from seam_match_rcnn_amd import _native, ops, retrieval

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    stub = args.stub_step
    if stub:
        dev, dsync = torch.device("cpu"), (lambda: None)
        args.no_roofline = args.no_cpu_baseline = args.no_extras = True
    else:
        torch.cuda.set_device(local_rank)
        dev, dsync = torch.device("cuda", local_rank), torch.cuda.synchronize
        _native.lib()                              # fail loudly if the HIP extension is missing
    multi = world > 1 or args.force_collective          # the data-parallel code path (collective, barriers, rank reductions)
    out = sys.stdout
    if multi:
        # RCCL prints a version / host banner on the process's stdout (fd 1): keep the contract's ONE JSON line alone there by
        # pointing fd 1 at stderr for everything else and writing the line to a duplicate of the real stdout
        sys.stdout.flush()
        out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_ADDR" not in os.environ:             # --force-collective outside torchrun: a one-rank group on the loopback
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", WORLD_SIZE="1")
        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    B = args.clips
    lo, hi = retrieval.shard_range(G, rank, world)
    # this rank's rows of the product-descriptor bank: extractor output (below, once the model exists); the CPU-only stub flow has
    # no extractor and stands in a fixed table
    bank_shard = torch.from_numpy(synth.gallery(7, G)[lo:hi]).to(dev) if stub else None
    bank_info = {"source": "stub table (synth.gallery; --stub-step has no extractor)"} if stub else None
    # test hook (stub runs only): this rank ends up with a DIFFERENT bank than its peers -- exercises the pre-timing check
    corrupt = stub and os.environ.get("SEAM_BENCH_TEST_CORRUPT_RANK") == str(rank)
    gathers = []                                                             # timed all-gathers of the measured steps
    model = ta = frame_list = rois = types = ids = side = None
    if not stub:
        import seam_match_rcnn_amd.models.detection as det
        if args.dtype == "f16" and not args.two_streams:
            args.single_stream = True
        if args.single_stream:
            det.BODY_STREAMS, det.LEVEL_STREAMS = 1, False
        log("building synthetic weights")
        model, sd = build_model(dev)
        if args.dtype == "f16":
            model.set_compute_dtype(torch.float16)
        elif args.dtype == "bf16x3":      # fp32 activations, split-bf16 operands (3 bf16 MFMAs per product), fp32 accumulate
            model.set_compute_dtype(ops.BX3)
        log("weights on device; generating frames")
        ta = model.roi_heads.temporal_aggregator
        # the step hands 'roi_features' straight to the aggregator: take the NHWC tile as a channels_last view instead of the
        # drop-in default (a contiguous NCHW copy that the aggregator would only transpose back)
        model.roi_heads.roi_features_contiguous = False
        frames = torch.cat([torch.from_numpy(synth.frames(rank * B + c, T, H, W)) for c in range(B)]).to(dev)   # this rank's clips, resident
        frame_list = list(frames.unbind(0))
        from seam_match_rcnn_amd.models.detection import resized_size
        rh, rw, _ = resized_size(H, W)                                           # ROIs live in the resized frame
        rois_np = synth.fixed_rois(R, rh, rw)
        rois = [torch.from_numpy(rois_np).to(dev) for _ in range(T * B)]
        types = torch.zeros(B * T * R, dtype=torch.int32)                        # all street ROIs (CPU, as the ref passes)
        ids = torch.cat([c * R + torch.arange(R, dtype=torch.int64).repeat(T) for c in range(B)])   # sequence id = (clip, ROI slot)
        side = torch.cuda.Stream(device=dev) if multi else None
        # the bank shard: each product's shop image goes through the extractor ONCE, on the rank that owns its row
        log(f"bank rows [{lo}, {hi}) through the shop-side path ({SHOP_HW}x{SHOP_HW} shop images)")
        tb0 = time.perf_counter()
        bank_shard = compute_bank_rows(model, ta, lo, hi, dev)
        dsync()
        bank_info = {"source": "extractor output: synthetic shop image -> forward_fixed_rois -> aggregator shop descriptor "
                               "(ref evaluate_movingfashion.py:31-47), computed once before the timed region by the rank that owns the row",
                     "rows_this_rank": int(hi - lo), "shop_image": f"{SHOP_HW}x{SHOP_HW}", "seconds": round(time.perf_counter() - tb0, 2)}
        log(f"bank shard ready in {bank_info['seconds']} s")

    def run_step(flist, rlist, ty, sid, timed_gather=False):
        if stub:                                                                 # test hook: the exchange step alone, on CPU tensors
            pending = retrieval.gather_product_bank(bank_shard, G, force=multi)
            time.sleep(0.002 * (1 + rank))
            got = pending.wait()
            if corrupt:
                got = got.clone()
                got[0, 0] += 1.0
            return None, None, None, None, None, got
        pending = retrieval.gather_product_bank(bank_shard, G, side_stream=side, timed=timed_gather, force=multi)   # overlaps the extractor
        if timed_gather and pending.events is not None:
            gathers.append(pending)
        res, feats, rpn = model.forward_fixed_rois(flist, rlist, run_rpn_head=True)
        roi_features = ops.cat_rows([r["roi_features"] for r in res])       # [K,256,14,14] (the per-image views, re-joined)
        out = ta(roi_features, ty, sid)                                      # Mode A: trunk + NLB + attention pool
        bank = pending.wait()
        if wl["rank"] == "logits":
            x5, idx, score = retrieval.match_sequences(ta, out[0], bank, TOPK)
        else:                                                                # configs 3/4: no [S,G,2] logits in HBM
            x5 = None
            idx, score = retrieval.match_sequences_topk(ta, out[0], bank, TOPK)
        return res, out, x5, idx, score, bank

    state = {"timed": False}

    def step():
        return run_step(frame_list, rois, types, ids, timed_gather=state["timed"])

    def sync_all():
        dsync()
        if multi:
            dist.barrier()
            dsync()

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
    def bank_identical(bank):
        """the gathered bank must be the same [G,256] bits on every rank: min == max over ranks of a row-weighted integer checksum"""
        wts = torch.arange(1, bank.shape[0] + 1, device=dev, dtype=torch.int64)
        cs = (bank.contiguous().view(torch.int32).to(torch.int64).sum(1) * wts).sum().reshape(1)   # exact: order-independent integers
        lo_cs, hi_cs = cs.clone(), cs.clone()
        dist.all_reduce(lo_cs, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_cs, op=dist.ReduceOp.MAX)
        return bool(lo_cs.item() == hi_cs.item()) and tuple(bank.shape) == (G, 256)

    log("warmup")
    last = None
    with torch.no_grad():
        # (the warm-up runs at the timed loop's queue depth -- not synchronised step by step -- so that the caching allocator meets the
        #  same number of in-flight steps there as in the timed region and has reached its plateau when timing starts)
        wpace = [] if (dev.type == "cuda" and not args.graph) else None
        for _ in range(args.warmup):
            if wpace is not None and len(wpace) >= QUEUE_DEPTH:
                wpace.pop(0).synchronize()
            last = run()
            if wpace is not None:
                ev = torch.cuda.Event()
                ev.record()
                wpace.append(ev)
            else:
                dsync()
            log("warmup step launched")
        sync_all()
        if multi and last is not None and not args.graph:
            # fail fast, BEFORE anything is timed: a rank whose all-gathered bank differs would time a different problem
            # (every rank takes part in the two reductions and reaches the same verdict, so all of them exit together)
            if not bank_identical(last[5]):
                raise SystemExit(f"bench.py: rank {rank}: the all-gathered product bank differs between ranks after the warm-up "
                                 f"steps -- refusing to time (check the shard ranges / the collective)")
            log("bank identical on all ranks (pre-timing check)")
        trace_marker()                  # (outside the timed region: the barrier + synchronize below come after it)
        sync_all()
        log("timing")
        state["timed"] = multi and not args.graph
        dev_allocs0 = int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0)) if dev.type == "cuda" else 0
        reserved0 = torch.cuda.memory_reserved(dev) if dev.type == "cuda" else 0
        t0 = time.perf_counter()
        # The host stays at most QUEUE_DEPTH steps ahead of the device (it waits for the event of step i - QUEUE_DEPTH before it
        # launches step i; the device always has >= 2 steps queued, so it never idles).  History: with `record_stream`-ed
        # side-stream tensors in the model an unpaced host made the caching allocator grow for the whole timed region -- such a block
        # is reusable only once the recorded work has RUN -- 2.4 GB and a hipMalloc per step (47.7 GB over 20 steps), each call a
        # device-wide stall of 100+ ms on a box whose HBM has not been touched since boot.  The model no longer records streams
        # (fork / join lifetimes: 0 allocations even unpaced); the pacing stays as the sane shape of a throughput loop.
        pace = [] if (dev.type == "cuda" and not args.graph) else None
        for _ in range(args.steps):
            if pace is not None and len(pace) >= QUEUE_DEPTH:
                pace.pop(0).synchronize()
            last = run()
            if pace is not None:
                ev = torch.cuda.Event()
                ev.record()
                pace.append(ev)
        sync_all()
        elapsed = time.perf_counter() - t0
        # hipMalloc calls of the caching allocator INSIDE the timed region: each one stalls the device for ~100 ms.  With the body on
        # two streams the allocator keeps a pool per stream, and a step whose launches interleave differently from the warm-up's
        # can still need fresh blocks -- the "slow mode" of the two-stream config-5 lines with --warmup 1
        dev_allocs_timed = (int(torch.cuda.memory_stats(dev).get("num_device_alloc", 0)) - dev_allocs0) if dev.type == "cuda" else 0
        reserved_growth = (torch.cuda.memory_reserved(dev) - reserved0) if dev.type == "cuda" else 0
        state["timed"] = False
        trace_marker()
    if args.graph:
        last = graph_out
    per_rank_ms = None
    bank_agree = None
    if multi:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = torch.empty(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(every, mine)
        per_rank_ms = [round(1e3 * float(e) / args.steps, 3) for e in every.tolist()]
        elapsed = float(every.max().item())                                     # the contract's MAX over ranks
        bank_agree = bank_identical(last[5])            # ... and again on the last timed step's bank

    ms_per_step = 1e3 * elapsed / args.steps
    log(f"timed: {ms_per_step:.2f} ms/step")
    value = world * B * args.steps / elapsed                                  # whole-job clips/s

    allgather = None
    med = -1.0
    if multi:
        # every rank takes part in the reduction below whether or not its own event timing worked (a rank-local failure must not
        # leave the others waiting in a collective): -1 marks "no measurement on this rank"
        try:
            us = sorted(g.elapsed_us() for g in gathers)
            med = us[len(us) // 2] if us else -1.0
        except Exception as e:          # noqa: BLE001 -- diagnostics only, the headline number does not depend on it
            log(f"all-gather timing unavailable: {e}")
            med = -1.0
        tt = torch.tensor([med], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)                              # slowest rank's median
        med = float(tt.item())
    if multi and med > 0:
        shard_bytes = (retrieval.shard_range(G, 0, world)[1]) * 256 * 4       # what a rank sends to EACH peer
        allgather = {"collective": "all_gather_into_tensor (RCCL over xGMI), side stream, overlapped with the extractor",
                     "bytes_sent_per_peer": shard_bytes, "bytes_received_per_rank": (world - 1) * shard_bytes,
                     "median_us_slowest_rank": round(med, 1),
                     "GB/s_per_link_if_direct": round(shard_bytes / med / 1e3, 2),
                     "GB/s_received_per_rank": round((world - 1) * shard_bytes / med / 1e3, 2),
                     "xgmi_link_peak_GB/s": XGMI_LINK_GBS,
                     "bound_us_direct_full_mesh": round(shard_bytes / XGMI_LINK_GBS / 1e3, 1),
                     "bound_us_ring": round((world - 1) * shard_bytes / XGMI_LINK_GBS / 1e3, 1),
                     "note": "device time of the collective between HIP events on its stream; it runs concurrently with the "
                             "extractor's kernels, so this is an upper bound on its isolated duration"}

    roofline = None
    if not args.no_roofline and (multi or rank == 0):
        # N > 1: the instrumented step contains the bank all-gather, so every rank runs it; rank 0's trace is reported
        roofline = roofline_leg(step, args.dtype)
        if rank != 0:
            roofline = None
        if multi:
            sync_all()

    cpu = cpu_out = None
    n_cpu = args.cpu_frames if args.cpu_frames is not None else T
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and (args.workload != "c5" or args.cpu_frames is not None)
    if want_cpu:
        log("cpu baseline")
        cpu, cpu_out = cpu_baseline(sd, frames[:T].cpu(), rois_np, n_cpu, wl, max(1, args.cpu_runs), last[5].cpu(),
                                    shop_images(0, dev)[:2].cpu())
        log("cpu baseline done")

    parity = None
    if cpu_out is not None:
        parity = parity_leg(last, cpu_out, wl, n_cpu, ta, args.dtype)
        log(f"parity: {parity}")

    extras = {}
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "c2" and not args.graph:
        log("extras: one clip per step, full drop-in forward")
        with torch.no_grad():
            n1 = T * R
            ids1 = torch.arange(R, dtype=torch.int64).repeat(T)
            one = lambda: run_step(frame_list[:T], rois[:T], types[:n1], ids1)   # noqa: E731
            for _ in range(2):
                one()
            torch.cuda.synchronize()
            k1 = max(5, args.steps)
            t0 = time.perf_counter()
            for _ in range(k1):
                one()
            torch.cuda.synchronize()
            extras["value_clips1"] = round(k1 / (time.perf_counter() - t0), 4)
            ts = []
            for i in range(4):                                                # drop-in model(images): RPN + box head + NMS + paste
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                det = model(frame_list[:T])
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            extras["full_forward_ms_per_clip"] = round(1e3 * sorted(ts[1:])[1], 3)
            extras["full_forward_detections_per_frame"] = [int(len(d["scores"])) for d in det]
            del det
            tb = []
            for i in range(3):                                                # the same forward on the step's whole batch (B clips)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                detb = model(frame_list)
                torch.cuda.synchronize()
                tb.append(time.perf_counter() - t0)
            extras["full_forward_clips_per_s"] = round(B / min(tb[1:]), 3)
            extras["full_forward_batched_ms_per_clip"] = round(1e3 * min(tb[1:]) / B, 3)
            extras["full_forward_batched_detections"] = int(sum(len(d["scores"]) for d in detb))
            del detb
            # host -> device of one step's frames (the boundary hands over host tensors in a real pipeline; `value` is measured with
            # the frames resident, as the contract says): pinned source, best of 3, NOT overlapped with compute
            host = frames.cpu().pin_memory()
            host_u8 = (frames.clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().cpu().pin_memory()
            dst, dst8 = torch.empty_like(frames), torch.empty(host_u8.shape, dtype=torch.uint8, device=dev)
            for name, src, d in (("h2d_ms_per_step", host, dst), ("h2d_ms_per_step_u8", host_u8, dst8)):
                best = None
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    d.copy_(src, non_blocking=True)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                extras[name] = round(1e3 * best, 3)
            extras["h2d_GBps"] = round(host.numel() * 4 / (extras["h2d_ms_per_step"] * 1e-3) / 1e9, 1)
            extras["value_pcie_inclusive"] = round(B / (ms_per_step * 1e-3 + extras["h2d_ms_per_step"] * 1e-3), 4)
            extras["value_pcie_inclusive_u8"] = round(B / (ms_per_step * 1e-3 + extras["h2d_ms_per_step_u8"] * 1e-3), 4)
            del host, host_u8, dst, dst8
            extras["extras_note"] = ("h2d_ms_per_step / value_pcie_inclusive: one step's frames copied from pinned host memory (fp32 CHW as "
                                     "the reference hands them over; _u8: HWC bytes) and the clips/s if that copy were NOT overlapped "
                                     "with the step -- a bound, never `value`; "
                                     "value_clips1: the same step with ONE clip per step (latency-oriented); "
                                     "full_forward_ms_per_clip: median of 3 of model(10 frames) with RPN proposals, box head, "
                                     "per-class NMS, mask + match branches and mask paste (not part of `value`); "
                                     "full_forward_clips_per_s: the same drop-in forward on the step's batch of clips "
                                     "(clips_per_step_per_gpu x frames images in ONE model(images) call), best of 2 after a warm-up")

    match_stage = None
    if rank == 0 and not stub and wl["rank"] == "topk" and not args.graph:
        # configs[2]/[3]: the similarity + top-k stage on its own (HIP events on the launch stream), this rank's sequences vs the bank
        with torch.no_grad():
            desc, bank = last[1][0].contiguous(), last[5]
            stats = torch.zeros(4, dtype=torch.int32, device=dev)
            for _ in range(3):
                ops.pair_topk(desc, bank, ta.last.weight, ta.last.bias, TOPK, stats=stats)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                ops.pair_topk(desc, bank, ta.last.weight, ta.last.bias, TOPK, stats=stats)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            q, g = desc.shape[0], bank.shape[0]
            gemm = 2.0 * q * g * 256 / us / 1e6
            match_stage = {"kernel": "seam_pair_topk_mfma_f32 (prep + sample GEMM + thresholds + filter GEMM + exact re-score / proof)",
                           "queries": q, "gallery": g, "topk": TOPK, "us_per_call": round(us, 1),
                           "gemm_tflops": round(gemm, 1), "frac_of_fp32_mfma_peak": round(gemm / FP32_MFMA_PEAK_TFLOPS, 4),
                           "gemm_flops_are": "2*Q*G*256 of the logit-difference GEMM (the direct form the oracle evaluates is 1536 FLOP per pair)",
                           "direct_form_equivalent_tflops": round(1536.0 * q * g / us / 1e6, 1),
                           "queries_redone_with_the_direct_form": int(stats[0]), "largest_candidate_list": int(stats[1]),
                           "bit_identical_to": "seam_pair_logits_f32 + seam_rank_topk_f32 (tests/test_gpu_pairmf.py)"}

    failed = None
    if rank == 0:
        what = (f"{wl['cfg']} full pipeline, fixed ROIs: {T} frames {H}x{W} -> ResNet-50-FPN + RPN head "
                f"-> RoIAlign 14x14 ({R} ROI/frame) -> mask head -> match trunk x2 -> SEAM NLB + "
                f"attention pool ({R} seq x {T}) -> " +
                (f"pair logits vs {G}-product bank -> top-{TOPK}" if wl["rank"] == "logits" else
                 f"MFMA similarity + fused top-{TOPK} vs {G}-product bank (no [S,G,2] tensor in HBM)"))
        gal = {1000: "1k", 20000: "20k", 50000: "50k"}[G]
        metric = (f"video-clips/sec (10f x 800^2, 32 ROI/f, {gal} gallery)" if args.workload != "c5"
                  else "video-clips/sec (30f x 1080p, 64 ROI/f, 1k gallery)")
        line = {"metric": metric, "value": round(value, 4),
                "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.dtype,
                "data": "synthetic" if not stub else "stub (--stub-step test hook: bank all-gather only, NOT a measurement)",
                "config": {"workload": what,
                           "clips_per_step_per_gpu": B, "hip_graph": bool(args.graph), "hip_streams": 1 if args.single_stream else 2, "frames": T, "rois_per_frame": R, "gallery": G, "topk": TOPK,
                           "algorithmic_tflop_per_clip": round(FLOP_PER_CLIP / 1e12, 3),
                           "parallelism": f"dp{world} (clips sharded; product bank all-gathered over RCCL)"
                           if world > 1 else "single GPU"},
                "pipeline_tflops": round(FLOP_PER_CLIP * value / 1e12, 2)}
        if not stub:
            free_b, total_b = torch.cuda.mem_get_info(dev)
            line["hbm_gb"] = {"allocator_peak_reserved": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1),
                              "allocator_peak_allocated": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                              "device_total": round(total_b / 2 ** 30, 1), "device_free_now": round(free_b / 2 ** 30, 1),
                              "allocator_retries": int(torch.cuda.memory_stats(dev).get("num_alloc_retries", 0)),
                              "device_allocs_in_timed_region": dev_allocs_timed,
                              "reserved_growth_in_timed_region": round(reserved_growth / 2 ** 30, 2)}
        line.update(extras)
        if per_rank_ms is not None:
            line["ms_per_step_per_rank"] = per_rank_ms
            line["bank_identical_on_all_ranks"] = bank_agree
            if not bank_agree:
                failed = "the all-gathered product bank differs between ranks"
        if bank_info is not None:
            line["bank"] = bank_info
        if allgather is not None:
            line["allgather"] = allgather
        if match_stage is not None:
            line["match_stage"] = match_stage
        if roofline is not None:
            # end to end: MFMA work issued for the step's convolutions / the TIMED step (all streams, every non-conv kernel and
            # launch gap included) / peak -- the fraction of the fp32-MFMA roof the whole step sustains
            ws = roofline["issued_tflop_per_step"] / (ms_per_step * 1e-3) / roofline["peak"]
            roofline["whole_step"] = {"frac": round(ws, 4), "issued_tflop_per_step": roofline["issued_tflop_per_step"],
                                      "ms_per_step": round(ms_per_step, 3),
                                      "is": "sum over conv launches of algorithmic FLOP x mfma_issue_ratio, / ms_per_step of the timed "
                                            "steps / peak"}
            line["roofline"] = roofline
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if parity is not None:
            line["parity"] = parity
            if not parity["ok"]:
                failed = f"parity outside tolerance: {parity}"
        out.write(json.dumps(line) + "\n")
        out.flush()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: " + failed)


# ------------------------------------------------------------------------------------------------ roofline
def mfma_issue_ratio(variant):
    """MFMA multiplies issued per algorithmic multiply-accumulate of the direct convolution, and the algorithm's name."""
    if variant.startswith("conv3x3_wino24"):
        return 1.0 / 3.0, "Winograd F(2x4,3x3): 24 MFMA multiplies per 2x4 output tile and channel pair instead of 72"
    if variant.startswith("conv3x3_wino"):
        return 1.0 / 2.25, "Winograd F(2x2,3x3): 16 MFMA multiplies per 2x2 output tile and channel pair instead of 36"
    if variant.startswith("conv_igemm_bx3"):
        return 3.0, "split-bf16: 3 bf16 MFMAs per fp32 product (hi*hi + hi*lo + lo*hi)"
    if variant.startswith("conv3x3_f16pc"):
        return 1.0, "direct 3x3 convolution, input patch staged in LDS: one MFMA multiply per algorithmic multiply"
    if variant.startswith("conv1x1_pc"):
        return 1.0, "producer / consumer pointwise GEMM (long reductions): one MFMA multiply per algorithmic multiply"
    if variant.startswith("conv1x1_sw"):
        return 1.0, "weights-stationary pointwise GEMM: one MFMA multiply per algorithmic multiply"
    return 1.0, "implicit GEMM: one MFMA multiply per algorithmic multiply"


def roofline_leg(step, dtype):
    """One instrumented step: every conv launch bracketed by HIP events on its launch stream (ops.CONV_TRACE).  The step runs on
    ONE stream (the timed steps overlap two batch halves on two streams, which stretches every kernel's own duration: a per-kernel
    roofline needs the kernel alone on the chip) -- the same configuration as `bench.py --single-stream`, under which the committed
    rocprofv3 kernel stats are taken."""
    import torch
    from seam_match_rcnn_amd import ops
    import seam_match_rcnn_amd.models.detection as det
    saved = det.BODY_STREAMS, det.LEVEL_STREAMS
    det.BODY_STREAMS, det.LEVEL_STREAMS = 1, False
    try:
        with torch.no_grad():
            step()                                  # untimed: allocator pools of the single-stream walk
            torch.cuda.synchronize()
            ops.CONV_TRACE = []
            step()
            torch.cuda.synchronize()
            trace, ops.CONV_TRACE = ops.CONV_TRACE, None
    finally:
        det.BODY_STREAMS, det.LEVEL_STREAMS = saved
    per = {}
    for variant, flops, e0, e1, _shape, nbytes in trace:
        a = per.setdefault(variant, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += nbytes
    dom = max(per, key=lambda k: per[k][2])
    n, fl, sec, alg_bytes = per[dom]
    algorithmic = fl / sec / 1e12
    peak = FP32_MFMA_PEAK_TFLOPS if dtype == "f32" else F16_MFMA_PEAK_TFLOPS
    issue, algo = mfma_issue_ratio(dom)
    achieved = algorithmic * issue
    src, traffic, busy, clock = pmc_fields(dom, dtype)

    def rate(v):
        return round(v[1] / v[2] / 1e12, 2)

    extra = {}
    if dtype == "f16":
        # frac stays against the nominal dense peak; what the chip sustains on this instruction is measured separately
        extra["sustained_mfma_tflops_measured"] = {
            "dense_random_operands": 1760.0, "zero_operands": 2492.0, "measured_in_this_run": False,
            "source": "profiles/r05_mfma_clock_trace.txt (a committed probe trace of an earlier round, quoted -- not re-measured here): bare v_mfma_f32_32x32x16_f16 on every SIMD, operands in registers -- "
                      "the 1300 W package limit holds the clock at 1.72 GHz on dense random operands (2.40 GHz / 842 W on zeros)"}
    return {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), **extra,
            "achieved_is": "MFMA FLOP/s issued for the algorithmic work = algorithmic_tflops x mfma_issue_ratio "
                           "(unfilled tile slots are not counted as work)",
            "measured_on": "one instrumented step on a single stream (= bench.py --single-stream; the timed steps run two streams)",
            "algorithm": algo, "mfma_issue_ratio": round(issue, 4),
            "algorithmic_tflops": round(algorithmic, 2),
            "algorithmic_flops_are": "2*M*K*R*S*C of the direct convolution per launch (SURVEY.md 8d)",
            "traffic": traffic, "mfma_busy_frac_pmc": busy, "clock_ghz_pmc": clock,
            "pmc_source": src,
            "algorithmic_bytes_per_launch": round(alg_bytes / n),
            "launches_per_step": n, "avg_launch_us": round(1e6 * sec / n, 2),
            "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
            "conv_ms_per_step": round(1e3 * sum(v[2] for v in per.values()), 3),
            "issued_tflop_per_step": round(sum(v[1] * mfma_issue_ratio(k)[0] for k, v in per.items()) / 1e12, 4),
            "other_variants": {k: {"launches": v[0], "ms_per_step": round(1e3 * v[2], 3), "algorithmic_TFLOP/s": rate(v)}
                               for k, v in per.items() if k != dom}}


QUEUE_DEPTH = int(os.environ.get("SEAM_BENCH_QUEUE_DEPTH", "2"))     # timed steps the host may run ahead of the device (see the timed loop)


def trace_marker():
    """One launch of a kernel nothing else in the run uses (ATen's ``spin_kernel``, ~1 us) right before and right after the timed
    region: ``tools/kernel_trace_steps.py`` cuts rocprofv3's kernel trace at the two launches, so the per-kernel averages committed
    under ``profiles/`` are those of the K timed steps -- not diluted by the bank pass (same kernels on 256x256 shop images), the
    warm-up or the instrumented legs behind the timed region."""
    import torch
    if not torch.cuda.is_available():       # the CPU flow of tests/test_bench_launch.py
        return
    try:
        torch.cuda._sleep(1)
    except AttributeError:  # a torch build without the private helper: the trace is then summarised whole
        pass


def csrc_digest():
    """Identity of the kernel sources a committed counter file was collected on: sha256 over EVERY csrc/*.hip and csrc/*.h file
    (name + bytes, sorted) -- a stale counter file is then never reported as current for any kernel of the library."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "seam-match-rcnn_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def pmc_fields(kernel, dtype="f32"):
    """(source, traffic, mfma_busy, clock GHz) of `kernel` from the newest COMMITTED rocprofv3 PMC passes (profiles/*_pmc_traffic.json;
    the fp16 config-5 passes: profiles/*_f16_pmc_traffic.json, made
    by tools/pmc_bench_traffic.sh over this same bench command on an earlier run -- NOT measured in this process; `source`
    names the file so that a reader can tell).  traffic = HBM-side bytes per launch: FETCH_SIZE and WRITE_SIZE are collected in
    separate passes and reported in KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so the read
    side is doubled (MI355X_MICROARCH.md, HBM section).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))
                   if f.endswith("_f16_pmc_traffic.json") == (dtype == "f16"))
    if not files or dtype not in ("f32", "f16"):
        return None, None, None, None
    traffic = busy = clock = None
    try:
        d = json.load(open(files[-1]))
    except (ValueError, OSError):
        return None, None, None, None
    try:
        traffic = round((2.0 * d["FETCH_SIZE"][kernel]["avg_per_launch"] + d["WRITE_SIZE"][kernel]["avg_per_launch"]) * 1024)
    except KeyError:
        pass
    try:
        busy = d["mfma_busy_frac"][kernel]
    except KeyError:
        pass
    try:
        clock = d["clock_ghz"][kernel]
    except KeyError:
        pass
    src = {"file": os.path.relpath(files[-1], ROOT), "measured_in_this_run": False,
           "how": "tools/pmc_bench_traffic.sh: separate rocprofv3 --pmc passes over `bench.py --steps 2 --warmup 1`; "
                  "traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 per launch"}
    if isinstance(d.get("_meta"), dict):
        src.update(d["_meta"])
    # the counters describe the kernels of the tree they were collected on: a file from other kernel sources is named, not used
    src["csrc_digest_now"] = csrc_digest()
    src["matches_current_sources"] = src.get("csrc_digest") == src["csrc_digest_now"]
    if not src["matches_current_sources"]:
        src["stale"] = "collected on different kernel sources: traffic / mfma_busy_frac_pmc withheld (re-run tools/pmc_bench_traffic.sh)"
        traffic = busy = clock = None
    return src, traffic, busy, clock


# ------------------------------------------------------------------------------------------------ CPU baseline + parity
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


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.lower().startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(sd, frames_cpu, rois_np, n_frames, wl, runs, bank, shop_cpu):
    """The CPU oracle (kind = "port": torch-CPU restatement, the same ATen kernels the reference
    dispatches) timed on the host cores for ONE clip (bounded sample), frame by frame: 1 warm-up run, then the
    minimum of `runs` timed runs.  Also returns the outputs of the last run (the parity leg's reference)."""
    import torch
    from oracle import detection as OD
    from oracle import heads as OH
    from oracle import model as OM
    import seam_match_rcnn_amd.synth as synth

    T, R, G = wl["T"], wl["R"], wl["G"]
    cores = usable_cores()
    torch.set_num_threads(cores)
    rois = torch.from_numpy(rois_np)
    n_frames = max(1, min(n_frames, frames_cpu.shape[0]))
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    tap = OM.sub(sd, "roi_heads.temporal_aggregator.")
    # the bank is the GPU path's (extractor output, computed before the timed region); its first rows are re-derived here through
    # the oracle's own shop-side path and compared in the parity leg
    with torch.no_grad():
        batch, sizes = OD.transform(list(shop_cpu), SHOP_HW, SHOP_HW)
        sfe = OD.fpn(OD.resnet50_body(batch, sd), sd)
        srf = OD.multiscale_roi_align([sfe[k] for k in "0123"], [shop_box()] * len(shop_cpu), sizes, 14)
        bank_rows_oracle = OH.match_trunk(srf, tap)

    def one_clip():
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
        x5 = OH.pair_logits(out[0], bank, tap["last.weight"], tap["last.bias"], chunk=8 if G > 4096 else 64)
        idx, score = OH.rank_topk(x5, TOPK)
        return dict(roi_features=x, x3_1b=out[0], x5=x5, idx=idx, score=score, bank_rows=bank_rows_oracle)

    secs = []
    with torch.no_grad():
        for i in range(1 + runs):                      # run 0 = warm-up (page-in, oneDNN primitive caches)
            t0 = time.perf_counter()
            outs = one_clip()
            secs.append(time.perf_counter() - t0)
    sec = min(secs[1:])
    clip_sec = sec * (T / n_frames)      # per-frame work dominates; scale when fewer frames were timed
    return ({"value": round(1.0 / clip_sec, 5), "unit": "clips/s", "cores": cores, "kind": "port",
             "cpu": cpu_model(), "threads": cores, "torch": torch.__version__,
             "runs_s": [round(s, 2) for s in secs],
             "sample": f"1 clip, {n_frames} of {T} frames timed; 1 warm-up + min of {runs} runs ({sec:.2f} s), same stages / inputs as "
                       f"the GPU step; torch {torch.__version__} CPU fp32, {cores} threads on {cpu_model()}"}, outs)


def parity_leg(last, cpu_out, wl, n_frames, ta, dtype):
    """Clip 0 of the LAST timed GPU step vs the CPU oracle's outputs on the same inputs (cpu_baseline's last run).
    err_of_scale = max|got - ref| / max|ref| per tensor; the gate is the north_star tolerance for exact fp32."""
    import torch
    res, out, x5, idx, score, bank = last
    T, R = wl["T"], wl["R"]
    n = max(1, min(n_frames, T))

    def err(a, b):
        a, b = a.detach().float().cpu(), b.detach().float().cpu()
        return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

    rf = torch.cat([res[f]["roi_features"] for f in range(n)])
    with torch.no_grad():
        if n == T:
            x3_1b, gidx = out[0][:R], idx[:R]
            gx5 = x5[:R] if x5 is not None else ta.pair(out[0][:R].contiguous(), bank)
        else:       # fewer CPU frames than the step ran: the SEAM head again, on the same frame subset
            ids = torch.arange(R, dtype=torch.int64).repeat(n)
            o = ta(rf, torch.zeros(n * R, dtype=torch.int32), ids)
            from seam_match_rcnn_amd import ops
            x3_1b, gx5 = o[0], ta.pair(o[0], bank)
            gidx, _ = ops.rank_topk(gx5, TOPK)
    torch.cuda.synchronize()
    errs = {"roi_features": err(rf, cpu_out["roi_features"]), "x3_1b": err(x3_1b, cpu_out["x3_1b"]),
            "match_logits": err(gx5, cpu_out["x5"]),
            "bank_rows": err(bank[:cpu_out["bank_rows"].shape[0]], cpu_out["bank_rows"])}
    gi, ci = gidx.cpu(), cpu_out["idx"]
    rows_equal = int((gi == ci).all(1).sum())
    overlap = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(gi, ci)) / float(ci.numel())
    tol = PARITY_TOL if dtype == "f32" else (2e-3 if dtype == "bf16x3" else 5e-2)
    worst = max(errs.values())
    return {"checked": True, "against": "cpu_baseline outputs (oracle, same synthetic inputs), clip 0 of the last timed step",
            "frames_compared": n, "max_err_of_scale": float(f"{worst:.3e}"),
            "err_of_scale": {k: float(f"{v:.3e}") for k, v in errs.items()},
            "topk_equal": rows_equal == gi.shape[0], "topk_rows_identical": f"{rows_equal}/{gi.shape[0]}",
            "topk_set_overlap": round(overlap, 4), "tolerance": tol,
            "ok": bool(worst <= tol and overlap >= (0.99 if dtype == "f32" else 0.9))}


if __name__ == "__main__":
    main()
