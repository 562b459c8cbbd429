#!/usr/bin/env python3
"""seam_conv1x1_f16pc (long-reduction fp16 1x1, round 6) vs what ops.conv2d otherwise picks (conv1x1_swh for C <= 512, the implicit
GEMM above) on the config-5 layer shapes: time per launch, rate, difference between the two (GPU box).
usage: pwhpc_ab.py [N,H,W,C,K ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops
DEFAULT = ["240,48,84,1024,256", "240,48,84,1024,512", "240,24,42,2048,512", "240,24,42,2048,256", "240,96,168,512,256", "240,96,168,512,128",
           "240,24,42,512,2048", "240,24,42,1024,2048", "15360,1,1,1024,256"]
shapes = [a for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
ops.PWHPC_MIN_C = 512
print(f"{'N,H,W,C,K':>22} {'other':>22} {'us':>9} {'TB/s':>6} {'f16pc us':>9} {'TB/s':>6} {'TF/s':>7} {'x':>5} {'max diff/scale':>14}")
for s in shapes:
    n, h, w, c, k = map(int, s.split(","))
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.relu(torch.randn(n, h, w, c, device=dev, generator=g)).half()          # post-ReLU operands, as the layers see them
    wt = torch.randn(k, c, 1, 1, device=dev, generator=g) / c ** 0.5
    bias = torch.randn(k, device=dev, generator=g)
    pc = ops.pack_conv(wt, bias, stride=1, pad=0, dtype=torch.float16)
    if pc.wph is None:
        print(f"{s:>22} not served"); continue
    outs, us, names = [], [], []
    for on in (False, True):
        ops.PWHPC = on
        ops.CONV_TRACE = []
        y = ops.conv2d(x, pc, True)
        names.append(ops.CONV_TRACE[0][0]); ops.CONV_TRACE = None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, pc, True, out=y)
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / reps)
        outs.append(y.float())
    by = 2.0 * (x.numel() + outs[0].numel())
    fl = 2.0 * n * h * w * c * k
    sc = float(outs[0].abs().max())
    print(f"{s:>22} {names[0]:>22} {us[0]:9.1f} {by/us[0]/1e6:6.2f} {us[1]:9.1f} {by/us[1]/1e6:6.2f} {fl/us[1]/1e6:7.1f} {us[0]/us[1]:5.2f} "
          f"{float((outs[0]-outs[1]).abs().max())/sc:14.2e}")
