#!/usr/bin/env python3
"""Micro-benchmark of seam_conv2d_f32 on chosen shapes (GPU box).
usage: conv_bench.py [N,H,W,C,K,R,stride,pad,res ...]   (defaults: the config-2 hot shapes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops

DEFAULT = ["10,200,200,256,256,3,1,1,0", "320,14,14,256,256,3,1,1,0", "10,200,200,64,256,1,1,0,1",
           "10,100,100,256,256,3,1,1,0", "10,50,50,256,256,3,1,1,0", "10,100,100,128,512,1,1,0,1",
           "10,50,50,256,1024,1,1,0,1", "320,8,8,256,1024,3,1,0,0", "10,200,200,64,64,3,1,1,0",
           "10,800,800,4,64,7,2,3,0", "10,25,25,256,256,3,1,1,0", "320,1,1,1024,256,1,1,0,0"]
f16 = "--f16" in sys.argv
bx3 = "--bx3" in sys.argv
dt = torch.float16 if f16 else torch.float32
pdt = ops.BX3 if bx3 else dt
shapes = [a for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
print(f"{'N,H,W,C,K,R,s,p,res':>30} {'us':>9} {'TF/s':>7} {'GB/s(alg)':>10}")
for s in shapes:
    n, h, w, c, k, r, st, pad, res = map(int, s.split(","))
    cin = 3 if c == 4 else c
    if f16 and c == 4:
        c = 8
    x = torch.randn(n, h, w, c, device=dev).to(dt)
    wt = torch.randn(k, cin, r, r, device=dev) * 0.05
    pc = ops.pack_conv(wt, torch.randn(k, device=dev), stride=st, pad=pad, cstore=c, dtype=pdt)
    ho, wo = (h + 2 * pad - r) // st + 1, (w + 2 * pad - r) // st + 1
    resid = torch.randn(n, ho, wo, k, device=dev).to(dt) if res else None
    y = torch.empty(n, ho, wo, k, device=dev, dtype=dt)
    for _ in range(3):
        ops.conv2d(x, pc, True, resid, out=y)
    torch.cuda.synchronize()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(x, pc, True, resid, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * n * ho * wo * k * r * r * cin
    by = float(x.element_size()) * (x.numel() + y.numel() * (2 if res else 1) + pc.w.numel())
    print(f"{s:>30} {us:9.1f} {fl/us/1e6:7.1f} {by/us/1e3:10.0f}")
