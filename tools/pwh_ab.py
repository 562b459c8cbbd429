#!/usr/bin/env python3
"""seam_conv1x1_swh_f16 (streaming weights-stationary fp16 pointwise kernel, round 6) vs conv_igemm<_Float16,128,*> on the 1x1 layers of
the config-5 (fp16) step: time per launch, HBM rate, fraction of the binding roof (max of the 6.29 TB/s measured HBM roof and the
2.5 PFLOP/s fp16 MFMA roof), deviation from each other (GPU box).   usage: pwh_ab.py [N,H,W,C,K[,res[,C2]] ...]   (C2 > 0: two sources)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops
# the 1x1 layers of a config-5 step with their launches per step (profiles/r05_breakdown_c5_f16.txt)
DEFAULT = [("240,192,336,64,256,1", 2), ("240,48,84,256,1024,1", 5), ("240,96,168,128,512,1", 3), ("240,192,336,256,256", 1),
           ("240,48,84,1024,256", 6), ("15360,14,14,256,1024", 1), ("240,192,336,256,64", 2), ("240,96,168,512,256", 2),
           ("240,192,336,64,256,0,64", 1), ("240,96,168,512,128", 3), ("240,192,336,256,128", 1), ("240,48,84,1024,512", 1),
           ("240,192,336,64,64", 1), ("240,24,42,512,2048,1", 2)]
shapes = [(a, 1) for a in sys.argv[1:] if not a.startswith("--")] or DEFAULT
dev = torch.device("cuda:0")
H = torch.float16
tot = [0.0, 0.0]
print(f"{'N,H,W,C,K,res,C2':>26} {'kernel':>18} {'igemm us':>9} {'TB/s':>5} {'swh us':>9} {'TB/s':>5} {'x':>5} {'%roof':>6} {'swh-igemm':>9}")
for s, cnt in shapes:
    v = list(map(int, s.split(",")))
    n, h, w, c, k = v[:5]
    use_res = len(v) > 5 and v[5]
    c2 = v[6] if len(v) > 6 else 0
    g = torch.Generator(device=dev); g.manual_seed(1)
    m = n * h * w
    x = torch.randn(n, h, w, c, device=dev, generator=g).half()
    res = torch.randn(n, h, w, k, device=dev, generator=g).half() if use_res else None
    ops.SWH = True
    if c2:
        x2 = torch.randn(n, h, w, c2, device=dev, generator=g).half()
        bn = lambda: (torch.rand(k, device=dev, generator=g) + 0.5, torch.randn(k, device=dev, generator=g) * 0.1,      # noqa: E731
                      torch.randn(k, device=dev, generator=g) * 0.1, torch.rand(k, device=dev, generator=g) + 0.5)
        pc = ops.pack_conv_dual(torch.randn(k, c, 1, 1, device=dev, generator=g) / c ** 0.5, bn(),
                                torch.randn(k, c2, 1, 1, device=dev, generator=g) / c2 ** 0.5, bn(), dtype=H)
        run = lambda: ops.conv2d_dual(x, x2, pc, 1, relu=True)      # noqa: E731
    else:
        wt = torch.randn(k, c, 1, 1, device=dev, generator=g) / c ** 0.5
        pc = ops.pack_conv(wt, torch.randn(k, device=dev, generator=g), dtype=H)
        run = lambda: ops.conv2d(x, pc, True, res)      # noqa: E731
    if pc.wsh is None:
        print(f"{s:>26}  not served"); continue
    outs, us, names = [], [1e30, 1e30], []
    for on in (False, True):
        ops.SWH = on
        ops.CONV_TRACE = []
        outs.append(run())
        names.append(ops.CONV_TRACE[0][0])
        ops.CONV_TRACE = None
    for rnd in range(3):                 # interleaved rounds, best of three per kernel
        for i, on in enumerate((False, True)):
            ops.SWH = on
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            us[i] = min(us[i], e0.elapsed_time(e1) * 1e3 / reps)
    assert names[1].startswith("conv1x1_swh") and names[0].startswith("conv_igemm"), names
    fl = 2.0 * m * k * (c + c2)
    by = 2.0 * m * (c + c2 + k + (k if use_res else 0))
    roof_us = max(by / 6.29e6, fl / 2500e6)
    sc = float(outs[0].float().abs().max())
    tot[0] += us[0] * cnt; tot[1] += us[1] * cnt
    print(f"{s:>26} {names[1]:>18} {us[0]:9.1f} {by/us[0]/1e6:5.2f} {us[1]:9.1f} {by/us[1]/1e6:5.2f} {us[0]/us[1]:5.2f} {100*roof_us/us[1]:6.1f} "
          f"{float((outs[0].float()-outs[1].float()).abs().max())/sc:9.2e}")
print(f"weighted per config-5 step: igemm {tot[0]/1e3:.2f} ms, swh {tot[1]/1e3:.2f} ms")
