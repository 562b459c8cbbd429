#!/usr/bin/env python3
"""Timing of the SEAM head kernels at config-2 / config-3 / config-4 sizes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import ops
from seam_match_rcnn_amd.models.match_head import pack_nlb_from_state

dev = torch.device("cuda:0")
p = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.temporal_aggregator_state(12).items()}
pk = pack_nlb_from_state(p)
W, B = p["last.weight"], p["last.bias"]

def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for s, t, g in ((32, 10, 1000), (256, 10, 20000), (256, 10, 50000), (64, 30, 1000)):
    seq = torch.randn(t, s, 256, device=dev)
    lens = torch.full((s,), t, dtype=torch.int32, device=dev)
    gal = torch.from_numpy(synth.gallery(7, g)).to(dev)
    a = torch.from_numpy(synth.normal(synth.stream_id(9, "q"), (s, 256))).to(dev)
    x5 = ops.pair_logits(a, gal, W, B)
    line = (f"S={s} T={t} G={g}: nlb+pool {timeit(lambda: ops.nlb_attnpool(seq, s*256, 256, lens, s, t, pk)):.1f} us | "
            f"pair_logits {timeit(lambda: ops.pair_logits(a, gal, W, B)):.1f} us | "
            f"rank_topk {timeit(lambda: ops.rank_topk(x5, 20)):.1f} us | "
            f"pair_topk(chunked VALU) {timeit(lambda: ops.pair_topk(a, gal, W, B, 20, mfma=False)):.1f} us | "
            f"pair_topk(fused VALU) {timeit(lambda: ops.pair_topk(a, gal, W, B, 20, fused=True, mfma=False)):.1f} us")
    if g >= 8192:
        st = torch.zeros(4, dtype=torch.int32, device=dev)
        us = timeit(lambda: ops.pair_topk(a, gal, W, B, 20, mfma=True, stats=st))
        flop = 2.0 * s * g * 256
        line += (f" | pair_topk(MFMA) {us:.1f} us = {flop / us / 1e6:.1f} TFLOP/s of the [Q,256]x[256,G] GEMM"
                 f" (direct-form pairs/s equivalent {1536.0 * s * g / us / 1e6:.0f} TFLOP/s); stats {st.tolist()}")
    print(line)
