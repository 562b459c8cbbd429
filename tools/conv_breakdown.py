#!/usr/bin/env python3
"""Per-layer conv table of one bench step (GPU box): shape, time, rate, and the layer's BINDING ROOF -- max(issued MFMA FLOP / MFMA
peak, algorithmic bytes / measured HBM copy rate) -- with the fraction of it the layer achieves; the last line is the time-weighted
fraction of the whole step (sum of roof times / sum of measured times).
usage: conv_breakdown.py [clips per step = 8] [f32 | f16 | bf16x3] [c2 | c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from seam_match_rcnn_amd import ops, retrieval
import seam_match_rcnn_amd.synth as synth

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev)
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
WL = sys.argv[3] if len(sys.argv) > 3 else "c2"
if DT == "f16":
    model.set_compute_dtype(torch.float16)
elif DT == "bf16x3":
    model.set_compute_dtype(ops.BX3)
ta = model.roi_heads.temporal_aggregator
wl = bench.WORKLOADS[WL]
T, R, H, W = wl["T"], wl["R"], wl["H"], wl["W"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8          # clips per step
frames = list(torch.cat([torch.from_numpy(synth.frames(c, T, H, W)) for c in range(B)]).to(dev).unbind(0))
s = min(800.0 / min(H, W), 1333.0 / max(H, W))
rois = [torch.from_numpy(synth.fixed_rois(R, int(H * s), int(W * s))).to(dev) for _ in range(T * B)]
types = torch.zeros(B * T * R, dtype=torch.int32); ids = torch.cat([c * R + torch.arange(R).repeat(T) for c in range(B)])
def step():
    res, feats, rpn = model.forward_fixed_rois(frames, rois)
    rf = torch.cat([r["roi_features"] for r in res])
    return ta(rf, types, ids)
import seam_match_rcnn_amd.models.detection as det
det.BODY_STREAMS, det.LEVEL_STREAMS = 1, False               # every kernel alone on the chip
with torch.no_grad():
    step(); step(); torch.cuda.synchronize()
    ops.CONV_TRACE = []
    step(); torch.cuda.synchronize()
    tr, ops.CONV_TRACE = ops.CONV_TRACE, None
PEAK = {"f32": bench.FP32_MFMA_PEAK_TFLOPS, "f16": bench.F16_MFMA_PEAK_TFLOPS, "bf16x3": bench.F16_MFMA_PEAK_TFLOPS}[DT] * 1e12
HBM = 6.29e12                                              # measured float4 copy rate (MI355X_MICROARCH.md); spec 8.0e12
agg = {}
for variant, fl, e0, e1, shp, nb in tr:
    a = agg.setdefault((variant, shp), [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += nb
tot = sum(v[2] for v in agg.values())
print(f"# {WL} {DT}, {B} clips per step ({B * T} frames {H}x{W}, {R} ROI/frame), single stream; roofs: MFMA {PEAK/1e12:.1f} TFLOP/s, HBM {HBM/1e12:.2f} TB/s")
print(f"{'variant':>28} {'N,H,W,C,K,R,s':>34} {'n':>3} {'ms':>8} {'%':>5} {'TF/s':>7} {'TB/s':>5} {'roof':>5} {'roof ms':>8} {'frac':>5}")
roof_tot = 0.0
for (v, shp), (n, fl, sec, nb) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    issue = bench.mfma_issue_ratio(v)[0]
    t_mfma, t_hbm = fl * issue / PEAK, nb / HBM
    roof = max(t_mfma, t_hbm)
    roof_tot += roof
    print(f"{v:>28} {str(shp):>34} {n:3d} {sec*1e3:8.3f} {100*sec/tot:5.1f} {fl/sec/1e12:7.1f} {nb/sec/1e12:5.2f} {'mfma' if t_mfma >= t_hbm else 'hbm':>5} "
          f"{roof*1e3:8.3f} {roof/sec:5.2f}")
print(f"total conv ms {tot*1e3:.3f}; sum of binding-roof times {roof_tot*1e3:.3f} ms => time-weighted fraction of the binding roofs {roof_tot/tot:.3f}")
