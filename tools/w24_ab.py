#!/usr/bin/env python3
"""conv3x3_wino24 on the bench's layer shapes: time per launch + deviation from the exact implicit GEMM (GPU box).
Run once per kernel variant (the variant knobs are read once per process): SEAM_W24_NT=1|2 ...
usage: w24_ab.py [N,H,W,C,K,pad ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops, _native
ops.WINO_MIN_FILL = 0
ops.WINOGRAD24 = 2
# (shape, launches per 8-clip bench step)
DEFAULT = [("80,200,200,256,256,1", 2), ("2560,14,14,256,256,1", 4), ("2560,14,14,256,256,0", 2), ("80,100,100,256,256,1", 2),
           ("80,50,50,256,256,1", 7), ("80,200,200,64,64,1", 3), ("2560,12,12,256,256,0", 2), ("80,100,100,128,128,1", 3),
           ("80,25,25,512,512,1", 2), ("2560,10,10,256,256,0", 2), ("80,25,25,256,256,1", 2), ("80,13,13,256,256,1", 1)]
shapes = [(a, 1) for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
tot = 0.0
import hashlib
print(f"variant: SEAM_W24_NT={os.environ.get('SEAM_W24_NT', 'auto')} SEAM_W24_PC={os.environ.get('SEAM_W24_PC', '1')} {os.environ.get('SEAM_W24_VARIANT', '')}")
print(f"{'N,H,W,C,K,pad':>24} {'kernel':>18} {'us':>9} {'TF/s(alg)':>9} {'issued %peak':>12} {'maxdiff/scale':>13} {'sha(y)':>10}")
for s, cnt in shapes:
    n, h, w, c, k, pad = map(int, s.split(","))
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(n, h, w, c, device=dev, generator=g)
    wt = torch.randn(k, c, 3, 3, device=dev, generator=g) * (1.0 / (3 * c ** 0.5))
    pc = ops.pack_conv(wt, torch.randn(k, device=dev, generator=g), stride=1, pad=pad)
    ops.WINOGRAD = False
    ref = ops.conv2d(x, pc, True)
    ops.WINOGRAD = True
    y = torch.empty_like(ref)
    for _ in range(2):
        ops.conv2d(x, pc, True, out=y)
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d(x, pc, True, out=y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * n * ref.shape[1] * ref.shape[2] * k * 9 * c
    diff = float((y - ref).abs().max()) / float(ref.abs().max())
    tot += us * cnt
    lib = _native.lib()
    kern = "wino24pc" if lib.seam_wino24_form(n, h, w, c, k, pad) == 1 else f"wino24<{lib.seam_wino24_variant(n, h, w, c, k, pad)}>"
    sha = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]
    print(f"{s:>24} {kern:>18} {us:9.1f} {fl/us/1e6:9.1f} {100*fl/us/1e6/3/157.3:12.1f} {diff:13.2e} {sha:>10}")
print(f"weighted total per bench step: {tot/1e3:.3f} ms")
