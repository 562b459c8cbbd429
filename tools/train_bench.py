#!/usr/bin/env python3
"""Time the grad-enabled pass of the heads (row f2) at a training-size batch: K ROIs -> trunk fwd+bwd of both heads,
losses, backward.  usage: train_bench.py [K_products] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import ops
from seam_match_rcnn_amd.models.match_head import (MatchPredictor, TemporalAggregationNLB, MatchLossWeak,
                                                   NEWBalancedAggregationMatchLossWeak)

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Fr = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = "cuda:0"
tt = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
mp, ta = MatchPredictor(), TemporalAggregationNLB()
mp.load_state_dict(tt(synth.match_predictor_state(11))); ta.load_state_dict(tt(synth.temporal_aggregator_state(12)))
mp, ta = mp.to(dev).train(), ta.to(dev).train()
ta.n_frames = 3
types, prod, img = [], [], []
i = 0
for p in range(P):
    types.append(1); prod.append(p); img.append(i); i += 1
    for f in range(Fr):
        nb = 1 + (f % 2)
        types += [0] * nb; prod += [p] * nb; img += [i] * nb
        i += 1
K = len(types)
x = torch.from_numpy(synth.roi_features(50, K)).to(dev)
types_t = torch.IntTensor(types)
l1f, l2f = MatchLossWeak(dev), NEWBalancedAggregationMatchLossWeak(dev, ta)
opt = torch.optim.SGD(list(mp.parameters()) + list(ta.parameters()), lr=1e-4)

def step():
    _, logits = mp(x, types_t)
    loss = l1f(logits, types_t, prod, img) + l2f(logits, types_t, prod, img, x)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
# trunk: fwd 0.5338 GFLOP/ROI, bwd = dgrad (3 of 4 convs + linear) + wgrad (all) ~ 2x
print(f"K={K} ROIs ({P} products x {Fr} frames): {dt*1e3:.2f} ms/step, loss {float(loss):.4f}; "
      f"mp trunk fwd+bwd ~ {K*0.5338*3/1e3:.2f} TFLOP (+ aggregator subset)")
