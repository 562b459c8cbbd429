#!/usr/bin/env python3
"""Per-parameter relative L2 error of the HIP training step vs the oracle (torch CPU autograd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import losses as OL
from test_gpu_train import make_heads, engine_step

tg = dict(np.load("tests/golden/train_golden.npz"))
types = torch.from_numpy(tg["types"])
x = torch.from_numpy(synth.roi_features(41, len(types)))
mp, ta = make_heads(3)
logits, l1, l2 = engine_step(mp, ta, x.to("cuda:0"), types, tg["prod_ids"].tolist(), tg["img_ids"].tolist())
ref = OL.train_step(x, types, tg["prod_ids"].tolist(), tg["img_ids"].tolist(), to_torch(synth.match_predictor_state(11)),
                    to_torch(synth.temporal_aggregator_state(12)), n_frames=3)
print("loss", float(l1), float(ref["match_loss"]), float(l2), float(ref["aggregation_loss"]))
for nm, m, gr in (("mp", mp, ref["grads_mp"]), ("ta", ta, ref["grads_ta"])):
    for k, p in m.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().double()
        w = gr[k].double()
        print(f"{nm}.{k:34s} |want|={float(w.norm()):.3e} relL2={float((g - w).norm() / (w.norm() + 1e-30)):.2e} maxabs={float((g-w).abs().max()):.2e}")

# ReLU-boundary flips between the GPU and CPU forward activations of the aggregator trunk
import torch.nn.functional as F
from seam_match_rcnn_amd import ops
from oracle import losses as OL2
plan = OL.aggregation_plan(ref["logits"], types, torch.as_tensor(tg["prod_ids"]), torch.as_tensor(tg["img_ids"]), 3)
xa = x[plan[0]]
sd = to_torch(synth.temporal_aggregator_state(12))
a_cpu, a_gpu = xa, ops.nchw_to_nhwc(xa.to("cuda:0"))
for i in (0, 2, 4, 6):
    a_cpu = F.relu(F.conv2d(a_cpu, sd[f"conv_seq.{i}.weight"], sd[f"conv_seq.{i}.bias"]))
    a_gpu = ops.conv2d(a_gpu, ops.pack_conv(sd[f"conv_seq.{i}.weight"].to("cuda:0"), sd[f"conv_seq.{i}.bias"].to("cuda:0")), relu=True)
    g = a_gpu.permute(0, 3, 1, 2).cpu()
    flips = ((g > 0) != (a_cpu > 0))
    print(f"conv_seq.{i}: {int(flips.sum())} mask flips of {flips.numel()}, max |diff| {float((g - a_cpu).abs().max()):.2e}, "
          f"values at flips: {a_cpu[flips].tolist()[:4]} {g[flips].tolist()[:4]}")
