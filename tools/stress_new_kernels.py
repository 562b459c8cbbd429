#!/usr/bin/env python3
"""Random-shape stress of the round-5 kernels against the implicit GEMM on the same operands (GPU box):
conv3x3_f16pc (fp16: two roundings apart), conv1x1_pc (fp32: summation order at most -- observed bit-identical).
usage: stress_new_kernels.py [cases per kernel] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from seam_match_rcnn_amd import ops, _native
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
lib = _native.lib()
ops.F16PC_RULE = False
ops.PWPC_MIN_HW = 1
bad = 0


def operands(n, h, w, c, k, r, half):
    g = torch.Generator(device=dev); g.manual_seed(rng.randrange(1 << 30))
    x = torch.randn(n, h, w, c, device=dev, generator=g)
    wt = torch.randn(k, c, r, r, device=dev, generator=g) / (r * c ** 0.5)
    bn = None
    bias = torch.randn(k, device=dev, generator=g) * 0.2
    if rng.random() < 0.5:
        bn = (torch.rand(k, device=dev, generator=g) + 0.5, torch.randn(k, device=dev, generator=g) * 0.1,
              torch.randn(k, device=dev, generator=g) * 0.1, torch.rand(k, device=dev, generator=g) + 0.5)
        bias = None
    return (x.half() if half else x), wt, bias, bn


done = 0
while done < ncase:
    if rng.random() < 0.5:
        h, w = rng.randint(9, 70), rng.randint(26, 110)
    else:
        w = rng.choice([6, 8, 10, 12, 14, 16]); h = w if rng.random() < 0.7 else rng.choice([6, 8, 10, 12, 14, 16])
    pad = rng.randint(0, 1)
    n, c, k = rng.randint(1, 70), rng.choice([128, 256, 384]), rng.choice([128, 256, 384])
    if lib.seam_conv3x3_f16pc_supported(n, h, w, c, k, pad) != 1:
        continue
    done += 1
    relu = rng.random() < 0.6
    x, wt, bias, bn = operands(n, h, w, c, k, 3, True)
    ops.F16PC, ops.CONV_TRACE = True, []
    pc = ops.pack_conv(wt, bias, bn, stride=1, pad=pad, dtype=torch.float16)
    got = ops.conv2d(x, pc, relu)
    assert ops.CONV_TRACE[0][0] == "conv3x3_f16pc", ops.CONV_TRACE[0][0]
    ops.F16PC, ops.CONV_TRACE = False, None
    ref = ops.conv2d(x, pc, relu)
    err = float((got.float() - ref.float()).abs().max()) / max(float(ref.float().abs().max()), 1e-20)
    ok = err <= 2e-3 and bool(torch.isfinite(got.float()).all())
    bad += not ok
    print(f"f16pc n={n} {h}x{w} c={c} k={k} pad={pad} relu={relu} bn={bn is not None}: {err:.2e} {'ok' if ok else 'FAIL'}")
ops.F16PC = True
exact = done = 0
while done < ncase:
    n, h, w = rng.randint(1, 20), rng.randint(1, 60), rng.randint(1, 60)
    c, k = rng.choice([256, 384, 512, 640, 1024, 2048]), rng.choice([128, 256, 384, 512, 1024])
    relu, use_res = rng.random() < 0.6, rng.random() < 0.4
    x, wt, bias, bn = operands(n, h, w, c, k, 1, False)
    res = torch.randn(n, h, w, k, device=dev) if use_res else None
    ops.PWPC, ops.CONV_TRACE = True, []
    pc = ops.pack_conv(wt, bias, bn)
    if pc.wq is None:          # (C = 256 layers the weights-stationary kernel takes)
        continue
    done += 1
    got = ops.conv2d(x, pc, relu, res)
    assert ops.CONV_TRACE[0][0] == "conv1x1_pc", ops.CONV_TRACE[0][0]
    ops.PWPC, ops.CONV_TRACE = False, None
    ref = ops.conv2d(x, pc, relu, res)
    err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-20)
    ok = err <= 2e-5 and bool(torch.isfinite(got).all())
    exact += bool(torch.equal(got, ref))
    bad += not ok
    print(f"pwpc M={n*h*w} c={c} k={k} relu={relu} bn={bn is not None} res={use_res}: {err:.2e} {'ok' if ok else 'FAIL'}")
ops.PWPC = True
print(f"failures: {bad}; conv1x1_pc bit-identical to the implicit GEMM in {exact} of {ncase} cases")
sys.exit(1 if bad else 0)
