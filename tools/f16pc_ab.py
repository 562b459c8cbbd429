#!/usr/bin/env python3
"""seam_conv3x3_f16pc (producer / consumer, round 5) vs conv_igemm<_Float16,...> on the config-5 layer shapes: time per launch, rate,
and the deviation of each from an fp32 convolution of the SAME fp16-rounded operands (GPU box).
usage: f16pc_ab.py [--norelu] [--full: reference on the whole batch] [--relu-input] [--zeros] [N,H,W,C,K,pad ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops, _native
DEFAULT = ["48,192,336,256,256,1", "1536,14,14,256,256,1", "48,96,168,256,256,1", "48,48,84,256,256,1", "1536,8,8,256,1024,0",
           "48,96,168,128,128,1", "1536,12,12,256,256,0", "1536,10,10,256,256,0", "1536,14,14,256,256,0", "48,24,42,512,512,1",
           "48,192,336,64,64,1"]
shapes = [a for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
lib = _native.lib()
ops.F16PC_RULE = False
print(f"{'N,H,W,C,K,pad':>24} {'igemm us':>10} {'TF/s':>7} {'f16pc us':>10} {'TF/s':>7} {'x':>5} {'%roof':>6} {'err igemm':>10} {'err f16pc':>10} {'pc-igemm':>9}")
for s in shapes:
    n, h, w, c, k, pad = map(int, s.split(","))
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(n, h, w, c, device=dev, generator=g).half()
    if "--relu-input" in sys.argv: x = torch.relu(x)          # the operand statistics of a layer behind a ReLU (half zeros)
    if "--zeros" in sys.argv: x = torch.zeros_like(x)          # no operand toggling: the clock stays at its maximum
    wt = torch.randn(k, c, 3, 3, device=dev, generator=g) * (1.0 / (3 * c ** 0.5))
    bias = torch.randn(k, device=dev, generator=g)
    relu = "--norelu" not in sys.argv
    pc = ops.pack_conv(wt, bias, stride=1, pad=pad, dtype=torch.float16)
    assert pc.wh is not None and lib.seam_conv3x3_f16pc_supported(n, h, w, c, k, pad) == 1, s
    outs, us = [], []
    for on in (False, True):
        ops.F16PC = on
        y = ops.conv2d(x, pc, relu)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ops.conv2d(x, pc, relu, out=y)
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / reps)
        outs.append(y.float())
    # fp32 reference on a slice of the batch (the fp16-rounded operands, fp32 arithmetic)
    m = n if "--full" in sys.argv else min(n, 4)
    ref = torch.nn.functional.conv2d(x[:m].float().permute(0, 3, 1, 2), wt.half().float(), bias, padding=pad).permute(0, 2, 3, 1)
    ref = torch.relu(ref) if relu else ref
    sc = float(ref.abs().max())
    fl = 2.0 * n * (h + 2 * pad - 2) * (w + 2 * pad - 2) * k * 9 * c
    print(f"{s:>24} {us[0]:10.1f} {fl/us[0]/1e6:7.1f} {us[1]:10.1f} {fl/us[1]/1e6:7.1f} {us[0]/us[1]:5.2f} {100*fl/us[1]/1e6/2500:6.1f} "
          f"{float((outs[0][:m]-ref).abs().max())/sc:10.2e} {float((outs[1][:m]-ref).abs().max())/sc:10.2e} {float((outs[0]-outs[1]).abs().max())/sc:9.2e}")
