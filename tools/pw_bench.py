#!/usr/bin/env python3
"""A/B of the weights-stationary pointwise kernel (csrc/seam_pw.hip) against the implicit GEMM on the 1x1 layers of the bench step:
max |difference| of the two results (both exact-fp32 fma chains; they differ by the folded scale and the k order) and us per launch.
usage: pw_bench.py [N,H,W,C,K,res,relu ...]      res: 0 none, 1 plain, 2 FPN top-down (coarse map H/2 x W/2), 3 dual (C = C1 + C1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops

DEFAULT = ["80,200,200,64,256,1,1", "80,200,200,64,256,0,1", "80,200,200,256,64,0,1", "80,200,200,64,64,0,1", "80,200,200,128,256,3,1",
           "80,200,200,256,128,0,1", "80,200,200,256,256,2,0", "80,100,100,128,512,1,1", "80,50,50,256,1024,1,1",
           "2560,14,14,256,1024,0,0"]
shapes = [a for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
torch.manual_seed(0)
print(f"{'N,H,W,C,K,res,relu':>26} {'igemm us':>9} {'TF/s':>6} {'sw us':>9} {'TF/s':>6} {'speedup':>7} {'max|d|/scale':>12}")
for s in shapes:
    n, h, w, c, k, res, relu = map(int, s.split(","))
    x = torch.randn(n, h, w, c if res != 3 else c // 2, device=dev).relu_()
    wt = torch.randn(k, c, 1, 1, device=dev) * (1.0 / c ** 0.5)
    bn = (torch.rand(k, device=dev) + 0.5, torch.randn(k, device=dev) * 0.1, torch.randn(k, device=dev) * 0.1, torch.rand(k, device=dev) + 0.5)
    if res == 3:
        x2 = torch.randn(n, h, w, c // 2, device=dev).relu_()
        pc = ops.pack_conv_dual(wt[:, :c // 2], bn, wt[:, c // 2:], bn)
        run = lambda: ops.conv2d_dual(x, x2, pc, 1, bool(relu))
    elif res == 2:
        top = torch.randn(n, (h + 1) // 2, (w + 1) // 2, k, device=dev)
        pc = ops.pack_conv(wt, torch.randn(k, device=dev) * 0.1)
        run = lambda: ops.conv2d_topdown(x, pc, top)
    else:
        pc = ops.pack_conv(wt, None, bn)
        resid = torch.randn(n, h, w, k, device=dev) if res == 1 else None
        run = lambda: ops.conv2d(x, pc, bool(relu), resid)
    out, tm = {}, {}
    for sw in (False, True):
        ops.SW = sw
        for _ in range(2):
            y = run()
        torch.cuda.synchronize()
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = run()
        e1.record(); torch.cuda.synchronize()
        tm[sw] = e0.elapsed_time(e1) * 1e3 / reps
        out[sw] = y
    d = float((out[True] - out[False]).abs().max()) / float(out[False].abs().max())
    fl = 2.0 * n * h * w * k * c
    print(f"{s:>26} {tm[False]:9.1f} {fl/tm[False]/1e6:6.1f} {tm[True]:9.1f} {fl/tm[True]/1e6:6.1f} {tm[False]/tm[True]:7.3f} {d:12.2e}", flush=True)
