"""GPU parity of the Winograd kernels -- F(2x2,3x3) (csrc/seam_wino.hip) and F(2x4,3x3) (csrc/seam_wino24.hip) -- through the C ABI: vs the fp32 CPU oracle
(torch conv2d = the ATen kernel the reference dispatches) and vs the implicit-GEMM kernel on the same inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

import seam_match_rcnn_amd.synth as synth

pytestmark = pytest.mark.gpu


def rnd(seed, shape, name="x"):
    return torch.from_numpy(synth.normal(synth.stream_id(seed, name), shape))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def ops():
    import seam_match_rcnn_amd.ops as ops
    saved = ops.WINO_MIN_FILL, ops.WINOGRAD24
    ops.WINO_MIN_FILL = 0            # force the Winograd kernel on every eligible shape, however poorly it fills its tiles
    yield ops
    ops.WINO_MIN_FILL, ops.WINOGRAD24 = saved


FORMS = [pytest.param(0, id="F2x2"), pytest.param(2, id="F2x4")]     # ops.WINOGRAD24: 0 = never, 2 = always


WINO_CASES = [
    # N, C, H, W, K, pad, bn, res, relu
    (2, 64, 20, 24, 64, 1, True, False, True),        # layer1 3x3
    (1, 128, 17, 19, 128, 1, True, False, True),      # odd map: half-filled edge tiles
    (2, 256, 13, 13, 256, 1, False, False, True),     # RPN head on the pool level
    (1, 256, 25, 32, 256, 1, False, False, False),    # FPN output conv (bias only)
    (3, 256, 14, 14, 256, 1, False, False, True),     # mask head
    (3, 256, 14, 14, 256, 0, False, False, True),     # match trunk: valid 3x3, 14 -> 12
    (2, 256, 8, 8, 1024, 0, False, False, True),      # match trunk last conv, 8 -> 6
    (1, 512, 7, 9, 512, 1, True, True, True),         # layer4-like with a residual
    (1, 8, 5, 70, 32, 1, False, False, False),        # smallest legal channels, wide patch (several blocks per row)
    (1, 64, 3, 3, 32, 0, False, False, False),        # single output pixel
    (2, 24, 37, 41, 96, 1, False, True, False),       # C, K not powers of two; residual without ReLU
    (7, 256, 10, 10, 256, 0, False, False, True),     # 4x4 tiles per image: several images share a block, last block ragged
    (11, 256, 8, 8, 64, 0, False, True, True),        # 3x3 tiles per image, 6 (or 3) images per block, residual
    (5, 64, 12, 12, 64, 0, False, False, False),      # 5x5 tiles per image, 2 images per block, odd image count
    (1, 32, 200, 200, 32, 1, False, False, True),     # 100x100 tiles: main region + right strip + bottom strip
    (2, 16, 100, 50, 32, 1, False, True, False),      # 25 x 50 tiles: strips with their own patch shapes, residual
    (1, 8, 51, 35, 32, 0, False, False, False),       # odd valid map: 17 x 25 tiles (half tiles on both edges)
]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("case", WINO_CASES)
def test_wino_vs_oracle(ops, case, form):
    n, c, h, w, k, pad, bn, res, relu = case
    ops.WINOGRAD24 = form
    x = rnd(31, (n, c, h, w))
    wt = rnd(32, (k, c, 3, 3), "w") * (1.0 / math.sqrt(c * 9))
    bias = None if bn else rnd(33, (k,), "b") * 0.1
    ref = F.conv2d(x, wt, bias, 1, pad)
    bnp = None
    if bn:
        bw = torch.from_numpy(synth.uniform(synth.stream_id(34, "bw"), (k,), 0.5, 1.5))
        bb, rm = rnd(35, (k,), "bb") * 0.1, rnd(36, (k,), "rm") * 0.1
        rv = torch.from_numpy(synth.uniform(synth.stream_id(37, "rv"), (k,), 0.5, 1.5))
        bnp = (bw, bb, rm, rv)
        sc = bw * (rv + 1e-5).rsqrt()
        ref = ref * sc[None, :, None, None] + (bb - rm * sc)[None, :, None, None]
    resid = None
    if res:
        resid = rnd(38, ref.shape, "res")
        ref = ref + resid
    if relu:
        ref = F.relu(ref)
    d = torch.device("cuda:0")
    pc = ops.pack_conv(wt.to(d), None if bias is None else bias.to(d), None if bnp is None else tuple(t.to(d) for t in bnp),
                       stride=1, pad=pad)
    assert pc.u is not None and pc.u24 is not None, "layer should be Winograd-eligible"
    xin = nhwc(x).to(d)
    rin = None if resid is None else nhwc(resid).to(d)
    saved = ops.WINOGRAD
    try:
        ops.WINOGRAD = True
        yw = ops.conv2d(xin, pc, relu, rin)
        ops.WINOGRAD = False
        yd = ops.conv2d(xin, pc, relu, rin)
    finally:
        ops.WINOGRAD = saved
    torch.cuda.synchronize()
    yw, yd = yw.permute(0, 3, 1, 2).cpu(), yd.permute(0, 3, 1, 2).cpu()
    scale = float(ref.abs().max())
    # tolerance written here: 1e-3 relative (north_star) + 1e-4 of the tensor's max; measured: ~1e-6 of the max
    tol = 1e-3 * ref.abs() + 1e-4 * scale
    assert bool(((yw - ref).abs() <= tol).all()), f"winograd vs oracle: max err {float((yw - ref).abs().max()):.3e} (scale {scale:.3e})"
    assert float((yw - yd).abs().max()) <= 2e-5 * scale, f"winograd vs implicit GEMM: {float((yw - yd).abs().max()):.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("form", FORMS)
def test_wino_dgrad_weights(ops, form):
    """mode 2 pack (input-gradient weights: taps rotated, channels swapped) with the ReLU-mask epilogue (relu = 2)."""
    ops.WINOGRAD24 = form
    d = torch.device("cuda:0")
    wt = rnd(41, (64, 96, 3, 3), "w") / math.sqrt(96 * 9)        # forward conv 96 -> 64
    dy = rnd(42, (2, 64, 12, 10), "dy")
    act = rnd(43, (2, 96, 12, 10), "act")
    ref = F.conv_transpose2d(dy, wt, None, 1, 1) * (act > 0)
    pc = ops.pack_conv_dgrad(wt.to(d), pad_fwd=1)
    assert pc.u is not None
    saved = ops.WINOGRAD
    try:
        ops.WINOGRAD = True
        got = ops.conv2d(nhwc(dy).to(d), pc, 2, nhwc(act).to(d))
    finally:
        ops.WINOGRAD = saved
    got = got.permute(0, 3, 1, 2).cpu()
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


@pytest.mark.parametrize("form", FORMS)
def test_wino_batch_invariance_and_determinism(ops, form):
    """Each image's result is independent of the batch it rides in, and repeat launches are bit-identical."""
    ops.WINOGRAD24 = form
    d = torch.device("cuda:0")
    x = nhwc(rnd(51, (5, 256, 26, 30))).to(d)
    wt = (rnd(52, (256, 256, 3, 3), "w") / 48.0).to(d)
    pc = ops.pack_conv(wt, None, stride=1, pad=1)
    saved = ops.WINOGRAD
    try:
        ops.WINOGRAD = True
        y5 = ops.conv2d(x, pc, True)
        y5b = ops.conv2d(x, pc, True)
        y1 = ops.conv2d(x[3:4].contiguous(), pc, True)
    finally:
        ops.WINOGRAD = saved
    assert torch.equal(y5, y5b)
    assert torch.equal(y5[3:4], y1)


def test_wino24_two_ntile_variant_in_subprocess():
    """conv3x3_wino24<2> (two n-tiles per block, AccVGPR accumulators; opt-in with SEAM_W24_NT=2, read once at load time):
    kept parity-clean against the implicit GEMM in a child process."""
    import os
    import subprocess
    import sys
    code = r'''
import math, torch
import seam_match_rcnn_amd.ops as ops
import seam_match_rcnn_amd.synth as synth
ops.WINO_MIN_FILL = 0
d = torch.device("cuda:0")
worst = 0.0
for (n, c, h, w, k, pad) in [(2, 64, 20, 24, 64, 1), (3, 256, 14, 14, 256, 1), (3, 256, 14, 14, 128, 0), (1, 32, 51, 35, 192, 1)]:
    x = torch.from_numpy(synth.normal(synth.stream_id(5, "x"), (n, h, w, c))).to(d)
    wt = torch.from_numpy(synth.normal(synth.stream_id(6, "w"), (k, c, 3, 3))).to(d) / math.sqrt(9 * c)
    pc = ops.pack_conv(wt, None, stride=1, pad=pad)
    ops.WINOGRAD, ops.WINOGRAD24 = True, 2
    a = ops.conv2d(x, pc, True)
    ops.WINOGRAD = False
    b = ops.conv2d(x, pc, True)
    worst = max(worst, float((a - b).abs().max()) / float(b.abs().max()))
print("WORST", worst)
assert worst < 2e-5, worst
'''
    env = dict(os.environ, SEAM_W24_NT="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]


_PC_CHILD = r'''
import hashlib, math, sys, torch
import seam_match_rcnn_amd.ops as ops
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import _native
ops.WINO_MIN_FILL = 0
d = torch.device("cuda:0")
lib = _native.lib()
# N, C, H, W, K, pad, residual mode (0 none, 1 add, 2 ReLU mask), relu, scale
CASES = [(8, 256, 96, 88, 256, 1, 0, 1, True),       # main region + right strip + bottom strip, FrozenBN scale
         (24, 256, 50, 50, 256, 1, 1, 1, False),     # residual add
         (640, 256, 14, 14, 256, 1, 0, 1, False),    # stacked mode (maps narrower than a block), ROI-sized
         (640, 256, 12, 12, 256, 0, 0, 1, False),    # valid convolution (the match trunk)
         (16, 128, 100, 100, 128, 1, 0, 1, True),    # the short K loop (16 chunks)
         (48, 512, 25, 25, 512, 1, 0, 0, False),     # 64 chunks, no ReLU
         (40, 256, 77, 91, 64, 1, 2, 0, False),      # one n-tile pair, ReLU-mask epilogue, ragged map
         (4, 256, 200, 200, 128, 1, 0, 1, True),     # many tiles per block (persistent walk), two n-tile pairs
         (40, 64, 100, 100, 64, 1, 1, 1, True)]      # 8 chunks per tile: the request stage works a whole tile ahead
for (n, c, h, w, k, pad, res, relu, bn) in CASES:
    x = torch.from_numpy(synth.normal(synth.stream_id(5, "x"), (n, h, w, c))).to(d)
    wt = torch.from_numpy(synth.normal(synth.stream_id(6, "w"), (k, c, 3, 3))).to(d) / math.sqrt(9 * c)
    bias = torch.from_numpy(synth.normal(synth.stream_id(7, "b"), (k,))).to(d)
    bnp = None
    if bn:
        bnp = (torch.from_numpy(synth.uniform(synth.stream_id(8, "g"), (k,))).to(d) + 0.5, bias, bias * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(9, "v"), (k,))).to(d) + 0.5)
    pc = ops.pack_conv(wt, None, bnp, stride=1, pad=pad) if bn else ops.pack_conv(wt, bias, stride=1, pad=pad)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    r = torch.from_numpy(synth.normal(synth.stream_id(10, "r"), (n, ho, wo, k))).to(d) if res else None
    mode = 2 if res == 2 else bool(relu)
    ops.WINOGRAD, ops.WINOGRAD24 = True, 2
    a = ops.conv2d(x, pc, mode, r)
    a2 = ops.conv2d(x, pc, mode, r)
    assert torch.equal(a, a2), "repeat launches differ"
    ops.WINOGRAD = False
    b = ops.conv2d(x, pc, mode, r)
    form = lib.seam_wino24_form(n, h, w, c, k, pad)
    print("CASE", n, c, h, w, k, pad, res, "form", form, "sha", hashlib.sha1(a.cpu().numpy().tobytes()).hexdigest(),
          "err", float((a - b).abs().max()) / float(b.abs().max()))
'''


def _run_pc_child(env_extra):
    import os
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _PC_CHILD], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    rows = [ln.split() for ln in r.stdout.splitlines() if ln.startswith("CASE")]
    assert len(rows) == 9, r.stdout[-2000:]
    return rows


def test_wino24_producer_consumer_kernel():
    """conv3x3_wino24pc (round 5: the NT = 2 block as MFMA-only consumer waves + transform / load producer waves, persistent over
    its XCD's tiles) on layer shapes that exercise every branch of it -- three-region maps, stacked maps, valid convolution, both
    K-loop lengths (8, 16, 32, 64 chunks), every epilogue mode, many tiles per block: within 2e-5 of the exact implicit GEMM, repeat launches identical,
    and BIT-IDENTICAL to conv3x3_wino24<2> (SEAM_W24_PC=0, the round-4 kernel) and to its own one-tile-per-block launch
    (SEAM_W24_PERSIST=0): the three compute the same fma chains."""
    pc = _run_pc_child({})
    r4 = _run_pc_child({"SEAM_W24_PC": "0"})
    one = _run_pc_child({"SEAM_W24_PERSIST": "0"})
    for a, b, c in zip(pc, r4, one):
        assert a[1:8] == b[1:8] == c[1:8]
        assert a[9] == "1" and b[9] == "0" and c[9] == "1", (a, b, c)       # the launcher really took the kernel under test
        assert float(a[13]) < 2e-5 and float(b[13]) < 2e-5, (a, b)
        assert a[11] == b[11] == c[11], ("results differ between the kernels", a, b, c)


def test_full_size_layers_against_the_exact_kernel(ops):
    """BASELINE-size layers (one clip's 10 frames at 200^2 x 256 channels) through the kernels the bench uses -- F(2x4)
    Winograd for the 3x3, the row-stream kernel for the 15 RPN outputs -- against the exact implicit GEMM on the same inputs
    (size-independent property: both are the same linear map; max deviation in units of the output scale), plus linearity."""
    d = torch.device("cuda:0")
    n, h, w, c = 10, 200, 200, 256
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn((n, h, w, c), generator=g).abs_().to(d)                   # ReLU-like activations
    wt = (torch.randn((256, c, 3, 3), generator=g) / math.sqrt(9 * c)).to(d)
    pc = ops.pack_conv(wt, torch.zeros(256, device=d), stride=1, pad=1)
    saved = ops.WINOGRAD, ops.WINOGRAD24, ops.NARROW
    try:
        ops.WINOGRAD, ops.WINOGRAD24 = True, 2
        y24 = ops.conv2d(x, pc, False)
        y24_2x = ops.conv2d(2.0 * x, pc, False)
        ops.WINOGRAD = False
        yd = ops.conv2d(x, pc, False)
        scale = float(yd.abs().max())
        assert float((y24 - yd).abs().max()) <= 2e-5 * scale
        assert torch.equal(y24_2x, 2.0 * y24)                                  # scaling by a power of two is exact in every form
        wl = (torch.randn((15, c, 1, 1), generator=g) / math.sqrt(c)).to(d)
        pl = ops.pack_conv(wl, torch.randn(15, generator=g).to(d))
        ops.NARROW = True
        a = ops.conv2d(y24, pl)
        ops.NARROW = False
        b = ops.conv2d(y24, pl)
        assert a.shape == (n, h, w, 15) and float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    finally:
        ops.WINOGRAD, ops.WINOGRAD24, ops.NARROW = saved
