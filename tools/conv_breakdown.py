#!/usr/bin/env python3
"""Per-launch conv efficiency of one config-2 bench step (GPU box): shape, us, TFLOP/s.
usage: conv_breakdown.py [clips per step = 8] [f32 | f16 | bf16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from seam_match_rcnn_amd import ops, retrieval
import seam_match_rcnn_amd.synth as synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
if DT == "f16":
    model.set_compute_dtype(torch.float16)
elif DT == "bf16x3":
    model.set_compute_dtype(ops.BX3)
ta = model.roi_heads.temporal_aggregator
T, R = bench.WORKLOADS["c2"]["T"], bench.WORKLOADS["c2"]["R"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # clips per step
frames = list(torch.cat([torch.from_numpy(synth.frames(c, T, 800, 800)) for c in range(B)]).to(dev).unbind(0))
rois = [torch.from_numpy(synth.fixed_rois(R, 800, 800)).to(dev) for _ in range(T * B)]
types = torch.zeros(B * T * R, dtype=torch.int32); ids = torch.cat([c * R + torch.arange(R).repeat(T) for c in range(B)])
def step():
    res, feats, rpn = model.forward_fixed_rois(frames, rois)
    rf = torch.cat([r["roi_features"] for r in res])
    return ta(rf, types, ids)
with torch.no_grad():
    step(); step(); torch.cuda.synchronize()
    ops.CONV_TRACE = []
    step(); torch.cuda.synchronize()
    tr, ops.CONV_TRACE = ops.CONV_TRACE, None
agg = {}
for variant, fl, e0, e1, shp, _b in tr:
    a = agg.setdefault((variant[-9:], shp), [0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3
tot = sum(v[2] for v in agg.values())
print(f"{'variant':>10} {'N,H,W,C,K,R,s':>34} {'n':>3} {'ms':>8} {'%':>5} {'TF/s':>6}")
for (v, shp), (n, fl, sec) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    print(f"{v:>10} {str(shp):>34} {n:3d} {sec*1e3:8.3f} {100*sec/tot:5.1f} {fl/sec/1e12:6.1f}")
print("total conv ms", tot * 1e3)
