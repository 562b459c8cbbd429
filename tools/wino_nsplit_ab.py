#!/usr/bin/env python3
"""conv3x3_wino<MT> (F(2x2)) with the n-tile split over XCD groups (round 6) on / off: time per launch + result identity (GPU box).
usage: wino_nsplit_ab.py [N,H,W,C,K,pad ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops, _native
ops.WINO_MIN_FILL = 0
ops.WINOGRAD24 = 0
shapes = [a for a in sys.argv[1:]] or ["2560,8,8,256,1024,0", "640,8,8,256,1024,0", "2560,14,14,256,256,1", "80,25,25,512,512,1"]
dev = torch.device("cuda:0")
for s in shapes:
    n, h, w, c, k, pad = map(int, s.split(","))
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn(n, h, w, c, device=dev, generator=g)
    wt = torch.randn(k, c, 3, 3, device=dev, generator=g) / (3 * c ** 0.5)
    pc = ops.pack_conv(wt, torch.randn(k, device=dev, generator=g), stride=1, pad=pad)
    best, sha = {}, {}
    for rnd in range(3):
        for ns in (1, 0, 2, 4, 8):
            _native.set_option("SEAM_WINO_NSPLIT", ns)
            y = ops.conv2d(x, pc, True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv2d(x, pc, True, out=y)
            e1.record(); torch.cuda.synchronize()
            best[ns] = min(best.get(ns, 1e30), e0.elapsed_time(e1) * 100)
            sha[ns] = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:10]
    fl = 2.0 * n * (h + 2 * pad - 2) * (w + 2 * pad - 2) * k * 9 * c
    print(s, " ".join(f"nsplit={('rule' if ns == 0 else ns)}: {best[ns]:.1f} us ({100 * fl / best[ns] / 1e6 / 2.25 / 157.3:.1f} % issued)" for ns in (1, 2, 4, 8, 0)),
          "identical" if len(set(sha.values())) == 1 else f"DIFFERENT {sha}")
_native.set_option("SEAM_WINO_NSPLIT", 0)
