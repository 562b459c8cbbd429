#!/usr/bin/env python3
"""seam_conv1x1_pc_f32 (producer / consumer pointwise kernel, round 5) vs conv_igemm<float,128,128> on the long-reduction 1x1 layers of
the config-2 step: time per launch, rate, fraction of the fp32-MFMA roof, deviation from each other (GPU box).
usage: pwpc_ab.py [N,H,W,C,K[,res] ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops
DEFAULT = ["80,50,50,1024,256", "80,100,100,512,256", "80,100,100,512,128", "80,25,25,2048,512", "80,25,25,512,2048,1", "80,50,50,1024,512",
           "80,25,25,2048,256", "2560,14,14,256,1024"]
shapes = [a for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
ops.PWPC_MIN_HW = 64
print(f"{'N,H,W,C,K':>22} {'igemm us':>10} {'TF/s':>7} {'pc us':>10} {'TF/s':>7} {'x':>5} {'%roof':>6} {'pc-igemm':>9}")
for s in shapes:
    v = list(map(int, s.split(",")))
    n, h, w, c, k = v[:5]
    use_res = len(v) > 5 and v[5]
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(n, h, w, c, device=dev, generator=g)
    wt = torch.randn(k, c, 1, 1, device=dev, generator=g) / c ** 0.5
    bias = torch.randn(k, device=dev, generator=g)
    res = torch.randn(n, h, w, k, device=dev, generator=g) if use_res else None
    ops.PWPC = True
    pc = ops.pack_conv(wt, bias)
    if pc.wq is None:
        print(f"{s:>22}  not served"); continue
    outs, us, names = [], [1e30, 1e30], []
    for on in (False, True):
        ops.PWPC = on
        ops.CONV_TRACE = []
        outs.append(ops.conv2d(x, pc, True, res))
        names.append(ops.CONV_TRACE[0][0])
        ops.CONV_TRACE = None
    for rnd in range(3):                 # interleaved rounds, best of three per kernel
        for i, on in enumerate((False, True)):
            ops.PWPC = on
            y = outs[i]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                ops.conv2d(x, pc, True, res, out=y)
            e1.record(); torch.cuda.synchronize()
            us[i] = min(us[i], e0.elapsed_time(e1) * 1e3 / reps)
    assert names[1] == "conv1x1_pc" and names[0].startswith("conv_igemm"), names
    fl = 2.0 * n * h * w * k * c
    sc = float(outs[0].abs().max())
    print(f"{s:>22} {us[0]:10.1f} {fl/us[0]/1e6:7.1f} {us[1]:10.1f} {fl/us[1]/1e6:7.1f} {us[0]/us[1]:5.2f} {100*fl/us[1]/1e6/157.3:6.1f} {float((outs[0]-outs[1]).abs().max())/sc:9.2e}")
