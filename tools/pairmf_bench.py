#!/usr/bin/env python3
"""pair_topk(MFMA) alone at config-3 / config-4 sizes, for rocprofv3 --kernel-trace --stats (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import ops

dev = torch.device("cuda:0")
p = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in synth.temporal_aggregator_state(12).items()}
W, B = p["last.weight"], p["last.bias"]
sizes = [(256, 20000), (256, 50000)] if len(sys.argv) < 3 else [(int(sys.argv[1]), int(sys.argv[2]))]
for s, g in sizes:
    gal = torch.from_numpy(synth.gallery(7, g)).to(dev)
    a = torch.from_numpy(synth.normal(synth.stream_id(9, "q"), (s, 256))).to(dev)
    for _ in range(3):
        ops.pair_topk(a, gal, W, B, 20, mfma=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.pair_topk(a, gal, W, B, 20, mfma=True)
    e1.record(); torch.cuda.synchronize()
    print(f"Q={s} G={g}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per call")
