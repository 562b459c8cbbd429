#!/usr/bin/env python3
"""Winograd F(2x2,3x3) kernel vs the implicit-GEMM kernel on the path's stride-1 3x3 shapes (GPU box).
usage: wino_bench.py [N,H,W,C,K,pad ...]     env SEAM_WINO_MT=1|2 picks the tile variant"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops, _native
ops.WINO_MIN_FILL = 0

DEFAULT = ["80,200,200,256,256,1", "80,100,100,256,256,1", "80,50,50,256,256,1", "80,25,25,256,256,1", "80,13,13,256,256,1",
           "80,200,200,64,64,1", "80,100,100,128,128,1", "80,50,50,256,256,1", "80,25,25,512,512,1",
           "2560,14,14,256,256,1", "2560,14,14,256,256,0", "2560,12,12,256,256,0", "2560,10,10,256,256,0", "2560,8,8,256,1024,0",
           "10,200,200,256,256,1"]
shapes = [a for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
print(f"{'N,H,W,C,K,pad':>24} {'direct us':>10} {'TF/s':>7} {'F2x2 us':>10} {'TF/s(alg)':>9} {'x':>6} {'maxdiff/scale':>13} | {'F2x4 us':>10} {'TF/s(alg)':>9} {'x':>6} {'maxdiff/scale':>13} {'slots24/22':>10}")
for s in shapes:
    n, h, w, c, k, pad = map(int, s.split(","))
    x = torch.randn(n, h, w, c, device=dev)
    wt = torch.randn(k, c, 3, 3, device=dev) * (1.0 / (3 * c ** 0.5))
    pc = ops.pack_conv(wt, torch.randn(k, device=dev), stride=1, pad=pad)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    y = [torch.empty(n, ho, wo, k, device=dev) for _ in range(3)]
    us = []
    for wi in (0, 1, 2):
        ops.WINOGRAD = bool(wi)
        ops.WINOGRAD24 = 2 if wi == 2 else 0
        for _ in range(2):
            ops.conv2d(x, pc, True, out=y[wi])
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, pc, True, out=y[wi])
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / reps)
    fl = 2.0 * n * ho * wo * k * 9 * c
    diff = float((y[0] - y[1]).abs().max()) / float(y[0].abs().max())
    fill = _native.lib().seam_wino_slot_fill_pct(n, h, w, c, k, pad)
    diff24 = float((y[0] - y[2]).abs().max()) / float(y[0].abs().max())
    lib = _native.lib()
    ratio = lib.seam_wino24_issue_slots(n, h, w, c, k, pad) / max(1, lib.seam_wino_issue_slots(n, h, w, c, k, pad))
    print(f"{s:>24} {us[0]:10.1f} {fl/us[0]/1e6:7.1f} {us[1]:10.1f} {fl/us[1]/1e6:9.1f} {us[0]/us[1]:6.2f} {diff:13.2e} | "
          f"{us[2]:10.1f} {fl/us[2]/1e6:9.1f} {us[0]/us[2]:6.2f} {diff24:13.2e} {ratio:10.3f} fill {fill}%")
