#!/usr/bin/env python3
"""A/B of the two RoIAlign kernels at the bench step's size (GPU box): 80 frames' FPN maps, 32 fixed ROIs per frame, 14x14, 256 ch.
usage: roialign_ab.py [f32|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import _native, ops

dev = torch.device("cuda:0")
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.float32
n = 80
feats = [torch.randn(n, s, s, 256, device=dev).to(dt) for s in (200, 100, 50, 25)]
b = torch.from_numpy(synth.fixed_rois(32, 800, 800))
rois = torch.cat([torch.cat([torch.full((32, 1), float(i)), b], 1) for i in range(n)]).to(dev)
scales = [0.25, 0.125, 0.0625, 0.03125]
lib = _native.lib()
out = {}
for mode, name in ((0, "gather (one wave per bin)"), (1, "LDS-staged ROI row tiles"), (2, "LDS-staged ROI quadrants")):
    lib.seam_roi_align_set_lds(mode)
    for _ in range(3):
        y = ops.roi_align(feats, rois, scales, 14)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.roi_align(feats, rois, scales, 14)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    out[mode] = y
    print(f"{name:28s} {us:8.1f} us   output {y.numel() * y.element_size() / us / 1e3:7.0f} GB/s   ({len(rois)} ROIs, {dt})")
print("bit-identical:", torch.equal(out[0], out[1]) and torch.equal(out[0], out[2]))
