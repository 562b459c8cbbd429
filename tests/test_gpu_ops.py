"""GPU parity: every C-ABI kernel vs the CPU oracle on seeded inputs (run with -m gpu)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import detection as OD
from oracle import heads as OH

pytestmark = pytest.mark.gpu

RTOL = 1e-3          # north_star: 1e-3 relative fp32


def dev():
    return torch.device("cuda:0")


def assert_close(got, ref, rtol=RTOL, atol_scale=1e-4):
    """allclose(rtol, atol = atol_scale * max|ref|): logits cross zero."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    atol = atol_scale * max(float(ref.abs().max()), 1e-30)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), f"max err {float(err.max()):.3e} (tol {float(tol.min()):.3e}), {int(bad.sum())} bad of {bad.numel()}"


def rnd(seed, shape, name="x"):
    return torch.from_numpy(synth.normal(synth.stream_id(seed, name), shape))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.fixture(scope="module")
def ops():
    import seam_match_rcnn_amd.ops as ops
    return ops


CONV_CASES = [
    # N, C, H, W, K, R, stride, pad, bn, res, relu
    (2, 64, 20, 24, 64, 1, 1, 0, True, False, True),
    (1, 64, 30, 26, 256, 1, 1, 0, True, True, True),
    (2, 128, 17, 19, 128, 3, 1, 1, True, False, True),
    (2, 128, 18, 22, 128, 3, 2, 1, True, False, True),
    (1, 256, 14, 14, 512, 1, 2, 0, True, False, False),
    (1, 3, 64, 80, 64, 7, 2, 3, True, False, True),       # stem (3 -> 4 stored channels)
    (3, 256, 14, 14, 256, 3, 1, 0, False, False, True),   # match trunk: valid 3x3
    (2, 256, 8, 8, 1024, 3, 1, 0, False, False, True),
    (1, 256, 13, 16, 15, 1, 1, 0, False, False, False),   # RPN logits+deltas fused (K=15)
    (1, 256, 25, 32, 256, 3, 1, 1, False, False, False),  # FPN output conv (bias)
    (2, 2048, 7, 9, 256, 1, 1, 0, False, False, False),   # FPN lateral C5
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(ops, case):
    n, c, h, w, k, r, stride, pad, bn, res, relu = case
    x = rnd(1, (n, c, h, w))
    wt = rnd(2, (k, c, r, r), "w") * (1.0 / math.sqrt(c * r * r))
    bias = None if bn else rnd(3, (k,), "b") * 0.1
    bnp = None
    ref = F.conv2d(x, wt, bias, stride, pad)
    if bn:
        bw = torch.from_numpy(synth.uniform(synth.stream_id(4, "bw"), (k,), 0.5, 1.5))
        bb, rm = rnd(5, (k,), "bb") * 0.1, rnd(6, (k,), "rm") * 0.1
        rv = torch.from_numpy(synth.uniform(synth.stream_id(7, "rv"), (k,), 0.5, 1.5))
        bnp = (bw, bb, rm, rv)
        sc = bw * (rv + 1e-5).rsqrt()
        ref = ref * sc[None, :, None, None] + (bb - rm * sc)[None, :, None, None]
    resid = None
    if res:
        resid = rnd(8, ref.shape, "res")
        ref = ref + resid
    if relu:
        ref = F.relu(ref)
    d = dev()
    pc = ops.pack_conv(wt.to(d), None if bias is None else bias.to(d), None if bnp is None else tuple(t.to(d) for t in bnp),
                       stride=stride, pad=pad)
    xin = nhwc(x)
    if c % 4:
        xin = F.pad(xin, (0, 4 - c % 4))
    y = ops.conv2d(xin.to(d), pc, relu, None if resid is None else nhwc(resid).to(d))
    torch.cuda.synchronize()
    assert_close(y.permute(0, 3, 1, 2), ref)


def test_linear_and_fc6_as_conv(ops):
    d = dev()
    x = rnd(11, (37, 1024))
    w, b = rnd(12, (256, 1024), "w") / 32.0, rnd(13, (256,), "b")
    pc = ops.pack_conv(w.to(d), b.to(d))
    assert_close(ops.linear(x.to(d), pc, relu=True), F.relu(F.linear(x, w, b)))
    # fc6: flatten(C,7,7) @ W^T  ==  7x7 valid conv on the NHWC ROI tile
    xr = rnd(14, (5, 256, 7, 7))
    w6, b6 = rnd(15, (128, 256 * 49), "w6") / 112.0, rnd(16, (128,), "b6")
    pc6 = ops.pack_conv(w6.view(128, 256, 7, 7).to(d), b6.to(d))
    y = ops.conv2d(nhwc(xr).to(d), pc6, True).view(5, 128)
    assert_close(y, F.relu(F.linear(xr.flatten(1), w6, b6)))


def test_conv_transpose_2x2(ops):
    d = dev()
    x = rnd(17, (3, 256, 14, 14))
    wt, b = rnd(18, (256, 64, 2, 2), "wt") / 16.0, rnd(19, (64,), "bt")
    ref = F.relu(F.conv_transpose2d(x, wt, b, 2))                      # [3,64,28,28]
    pc = ops.pack_conv(wt.to(d), b.to(d), transposed2x2=True)
    y = ops.conv2d(nhwc(x).to(d), pc, True)                             # [3,14,14,4*64]
    y = y.view(3, 14, 14, 2, 2, 64).permute(0, 5, 1, 3, 2, 4).reshape(3, 64, 28, 28)
    assert_close(y, ref)


@pytest.mark.parametrize("hw", [((64, 80), (64, 80)), ((60, 90), None), ((108, 192), None)])
def test_preprocess(ops, hw):
    (h, w), _ = hw
    imgs = [torch.from_numpy(synth.uniform(synth.stream_id(20 + i, "img"), (3, h, w))) for i in range(2)]
    ref, sizes = OD.transform(imgs, min_size=96, max_size=160)
    # product side computes the same sizes on the host
    from seam_match_rcnn_amd.models.detection import resized_size
    psz = [resized_size(h, w, 96, 160)[:2] for _ in imgs]
    assert [tuple(s) for s in sizes] == psz
    hp, wp = ref.shape[-2:]
    out = ops.preprocess([i.to(dev()) for i in imgs], psz, hp, wp)
    assert_close(out[..., :3].permute(0, 3, 1, 2), ref, atol_scale=1e-5)
    assert float(out[..., 3].abs().max()) == 0.0


def test_maxpool_upsample_transpose_avgpool(ops):
    d = dev()
    x = rnd(30, (2, 64, 31, 37))
    assert_close(ops.maxpool2d(nhwc(x).to(d), 3, 2, 1).permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1), rtol=0, atol_scale=0)
    assert_close(ops.maxpool2d(nhwc(x).to(d), 1, 2, 0).permute(0, 3, 1, 2), F.max_pool2d(x, 1, 2, 0), rtol=0, atol_scale=0)
    # the 3x3 / stride-2 / pad-1 form has its own kernel (clamped window coordinates): even, odd and tiny maps, a channel count of one
    # 16-byte vector, an all-negative map (the padding must never win)
    for i, shp in enumerate([(3, 64, 40, 56), (1, 8, 2, 2), (2, 8, 3, 5), (1, 128, 7, 64), (2, 16, 2, 9)]):
        xs = rnd(300 + i, shp) - (5.0 if i == 1 else 0.0)
        assert torch.equal(ops.maxpool2d(nhwc(xs).to(d), 3, 2, 1).permute(0, 3, 1, 2).cpu(), F.max_pool2d(xs, 3, 2, 1)), shp
        xh = xs.half()
        assert torch.equal(ops.maxpool2d(nhwc(xh).to(d), 3, 2, 1).permute(0, 3, 1, 2).float().cpu(), F.max_pool2d(xh.float(), 3, 2, 1)), shp
    lat, top = rnd(31, (2, 32, 26, 34)), rnd(32, (2, 32, 13, 17))
    ref = lat + F.interpolate(top, size=lat.shape[-2:], mode="nearest")
    got = ops.upsample_add_(nhwc(lat).to(d), nhwc(top).to(d))
    assert_close(got.permute(0, 3, 1, 2), ref, rtol=1e-6)
    r = rnd(33, (5, 256, 14, 14))
    t = ops.nchw_to_nhwc(r.to(d))
    assert torch.equal(t.cpu(), nhwc(r))
    assert torch.equal(ops.nhwc_to_nchw(t).cpu(), r)
    a = rnd(34, (7, 1024, 6, 6))
    assert_close(ops.avgpool(nhwc(a).to(d)), F.avg_pool2d(a, 6).flatten(1), rtol=1e-5)


@pytest.mark.parametrize("shape", [((3, 64, 26, 34), (13, 17)), ((5, 128, 25, 25), (13, 13)), ((2, 512, 50, 50), (25, 25)),
                                   ((9, 64, 13, 13), (7, 7))])
def test_conv_topdown_fused(ops, shape):
    """Lateral 1x1 conv + nearest top-down merge in one launch == conv followed by upsample_add_ (bit-exact: same
    rounding order) == the torch reference (FPN forward [TV])."""
    d = dev()
    (n, c, h, w), (ht, wt) = shape
    x, top = rnd(60, (n, c, h, w)), rnd(61, (n, 256, ht, wt))
    wgt, b = rnd(62, (256, c, 1, 1), "w") / (c ** 0.5), rnd(63, (256,), "b")
    pc = ops.pack_conv(wgt.to(d), b.to(d))
    got = ops.conv2d_topdown(nhwc(x).to(d), pc, nhwc(top).to(d))
    two = ops.upsample_add_(ops.conv2d(nhwc(x).to(d), pc), nhwc(top).to(d))
    assert torch.equal(got, two)
    ref = F.conv2d(x, wgt, b) + F.interpolate(top, size=(h, w), mode="nearest")
    assert_close(got.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("shape", [(3, 64, 20, 24, 64, 1, 256), (2, 128, 13, 17, 256, 2, 512), (5, 32, 7, 9, 96, 2, 64),
                                   (1, 512, 25, 25, 1024, 2, 2048)])
def test_conv_dual_source(ops, shape):
    """Projection-shortcut block as one GEMM over two 1x1 sources == bn3(conv3(h)) + bn_d(conv_d(x)) + ReLU (torch),
    and == the two-launch form of the same ops within the rounding of the folded scales."""
    d = dev()
    n, c1, ho, wo, c2, s2, k = shape
    h = rnd(70, (n, c1, ho, wo))
    x = rnd(71, (n, c2, ho * s2 - (s2 - 1), wo * s2 - (s2 - 1)))            # odd input size for stride 2
    w3, wd = rnd(72, (k, c1, 1, 1), "w3") / (c1 ** 0.5), rnd(73, (k, c2, 1, 1), "wd") / (c2 ** 0.5)
    def bn(seed):
        return (torch.from_numpy(synth.uniform(synth.stream_id(seed, "bw"), (k,), 0.5, 1.5)), rnd(seed + 1, (k,), "bb") * 0.1,
                rnd(seed + 2, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(seed + 3, "rv"), (k,), 0.5, 1.5)))
    b3, bd = bn(74), bn(78)
    def fbn(y, b):
        sc = b[0] * (b[3] + 1e-5).rsqrt()
        return y * sc[None, :, None, None] + (b[1] - b[2] * sc)[None, :, None, None]
    ref = F.relu(fbn(F.conv2d(h, w3), b3) + fbn(F.conv2d(x, wd, None, s2), bd))
    pc = ops.pack_conv_dual(w3.to(d), tuple(t.to(d) for t in b3), wd.to(d), tuple(t.to(d) for t in bd))
    got = ops.conv2d_dual(nhwc(h).to(d), nhwc(x).to(d), pc, s2, relu=True)
    assert_close(got.permute(0, 3, 1, 2), ref)
    p3 = ops.pack_conv(w3.to(d), None, tuple(t.to(d) for t in b3))
    pd = ops.pack_conv(wd.to(d), None, tuple(t.to(d) for t in bd), stride=s2)
    two = ops.conv2d(nhwc(h).to(d), p3, relu=True, residual=ops.conv2d(nhwc(x).to(d), pd))
    assert float((got - two).abs().max()) <= 2e-5 * float(two.abs().max())


@pytest.mark.parametrize("hw", [(64, 96), (70, 100)])
def test_stem_space_to_depth(ops, hw):
    """Space-to-depth form of the stem (preprocess -> [N,H/2,W/2,12], 4x4 / stride-1 conv with re-indexed weights, cropped
    output grid) == the oracle's transform + 7x7 / stride-2 / pad-3 conv + FrozenBN + ReLU, and == the NHWC4 form."""
    d = dev()
    h, w = hw
    from seam_match_rcnn_amd.models.detection import resized_size
    clip = torch.from_numpy(synth.uniform(synth.stream_id(80, "clip"), (3, 3, h, w)))
    imgs = list(clip.to(d).unbind(0))
    ref_in, sizes = OD.transform(list(clip.unbind(0)), min_size=96, max_size=160)
    hp, wp = ref_in.shape[-2:]
    sz = [resized_size(h, w, 96, 160)[:2]] * 3
    wt = rnd(81, (64, 3, 7, 7), "w") / (147 ** 0.5)
    bnp = (torch.from_numpy(synth.uniform(synth.stream_id(82, "bw"), (64,), 0.5, 1.5)), rnd(83, (64,), "bb") * 0.1,
           rnd(84, (64,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(85, "rv"), (64,), 0.5, 1.5)))
    sc = bnp[0] * (bnp[3] + 1e-5).rsqrt()
    ref = F.relu(F.conv2d(ref_in, wt, None, 2, 3) * sc[None, :, None, None] + (bnp[1] - bnp[2] * sc)[None, :, None, None])
    x12 = ops.preprocess(imgs, sz, hp, wp, s2d=True)
    assert x12.shape == (3, hp // 2, wp // 2, 12)
    x4 = ops.preprocess(imgs, sz, hp, wp)
    # layout: channel (dy*2+dx)*3 + c of cell (Y, X) = colour c of pixel (2Y+dy, 2X+dx)
    back = x12.view(3, hp // 2, wp // 2, 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(3, hp, wp, 3)
    assert torch.equal(back, x4[..., :3])
    w8 = F.pad(wt, (1, 0, 1, 0))
    ws = w8.view(64, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(64, 12, 4, 4).contiguous()
    bn_d = tuple(t.to(d) for t in bnp)
    pc12 = ops.pack_conv(ws.to(d), None, bn_d, stride=1, pad=2, cstore=12, wino=False)
    got = ops.conv2d(x12, pc12, relu=True, out_hw=(hp // 2, wp // 2))
    assert_close(got.permute(0, 3, 1, 2), ref)
    pc4 = ops.pack_conv(wt.to(d), None, bn_d, stride=2, pad=3, cstore=4)
    old = ops.conv2d(x4, pc4, relu=True)
    assert old.shape == got.shape and float((old - got).abs().max()) <= 2e-5 * float(old.abs().max())


@pytest.mark.parametrize("shape", [(2, 23, 31, 256, 15), (1, 1, 1, 256, 14), (3, 7, 5, 64, 16), (1, 40, 33, 128, 3), (5000, 1, 1, 256, 14)])
def test_linear_narrow(ops, shape):
    """<= 16-output 1x1 layers (RPN logits + deltas, mask logits) on the row-stream kernel == torch, and == the implicit GEMM
    within fp32 summation-order noise; ragged row counts (M not a multiple of 16)."""
    d = dev()
    n, h, w, c, k = shape
    x = rnd(90, (n, c, h, w))
    wt, b = rnd(91, (k, c, 1, 1), "w") / (c ** 0.5), rnd(92, (k,), "b")
    pc = ops.pack_conv(wt.to(d), b.to(d))
    assert pc.wn is not None
    saved = ops.NARROW
    try:
        ops.NARROW = True
        got = ops.conv2d(nhwc(x).to(d), pc, relu=(k == 16))
        ops.NARROW = False
        ref_gemm = ops.conv2d(nhwc(x).to(d), pc, relu=(k == 16))
    finally:
        ops.NARROW = saved
    ref = F.conv2d(x, wt, b)
    if k == 16:
        ref = F.relu(ref)
    assert_close(got.permute(0, 3, 1, 2), ref)
    assert float((got - ref_gemm).abs().max()) <= 2e-5 * float(ref_gemm.abs().max())


def test_preprocess_batched_clip_tensor(ops):
    """The frames of one clip tensor (same-shape views at a constant stride) go through ONE launch; identical to the
    per-image launches."""
    d = dev()
    from seam_match_rcnn_amd.models.detection import resized_size
    clip = torch.from_numpy(synth.uniform(synth.stream_id(64, "clip"), (5, 3, 60, 90))).to(d)
    sz = [resized_size(60, 90, 96, 160)[:2]] * 5
    hp, wp = 96, 160
    batched = ops.preprocess(list(clip.unbind(0)), sz, hp, wp)
    single = torch.cat([ops.preprocess([clip[i].clone()], sz[:1], hp, wp) for i in range(5)])
    assert torch.equal(batched, single)
    # every other frame: still a constant stride
    strided = ops.preprocess(list(clip[::2].unbind(0)), sz[:3], hp, wp)
    assert torch.equal(strided, single[::2])
    ref, _ = OD.transform(list(clip.cpu().unbind(0)), min_size=96, max_size=160)
    assert_close(batched[..., :3].permute(0, 3, 1, 2), ref, atol_scale=1e-5)


def test_roi_align_multiscale(ops):
    d = dev()
    sizes = [(200, 200)] * 2
    feats = [rnd(40 + i, (2, 256, 200 // 2 ** (i + 2) if i < 3 else 7, 200 // 2 ** (i + 2) if i < 3 else 7))
             for i in range(4)]
    feats = [rnd(40 + i, (2, 256, s, s)) for i, s in enumerate((50, 25, 13, 7))]
    boxes = [torch.tensor([[10., 12., 60., 80.], [0., 0., 200., 200.], [150., 20., 199., 70.],
                           [30.5, 40.25, 31.0, 40.5], [100., 100., 190., 195.], [-5., -8., 40., 30.]]),
             torch.from_numpy(synth.fixed_rois(8, 200, 200))]
    for pooled in (7, 14):
        ref = OD.multiscale_roi_align(feats, boxes, sizes, pooled)
        rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i)), b], 1) for i, b in enumerate(boxes)])
        scales = OD.infer_scales([f.shape[-2:] for f in feats], sizes)
        out = ops.roi_align([nhwc(f).to(d) for f in feats], rois.to(d), scales, pooled)
        assert_close(out.permute(0, 3, 1, 2), ref, atol_scale=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_roi_align_lds_tiles_equal_the_gather_kernel(ops, dtype):
    """The LDS-staged ROI-tile kernel and the one-wave-per-bin gather kernel evaluate the same expression per sample in the same
    order: bit-identical outputs -- on the config-2 ROI set, on boxes hanging over every image edge, degenerate boxes, and a box
    wider than the 64-pixel staging window (gather path inside the LDS kernel)."""
    from seam_match_rcnn_amd import _native
    d = dev()
    feats = [nhwc(rnd(140 + i, (2, 256, h, w))).to(d).to(dtype) for i, (h, w) in enumerate(((200, 336), (100, 168), (50, 84), (25, 42)))]
    boxes = torch.cat([torch.from_numpy(synth.fixed_rois(32, 800, 800)),
                       torch.tensor([[-30., -20., 90., 70.], [700., 750., 1400., 900.], [5., 5., 5.5, 5.2], [0., 0., 1344., 800.],
                                     [10., 300., 1330., 340.],          # 1320 px wide, 40 tall: level 2 (stride 4) -> 330 feature px
                                     [600., -100., 640., 900.], [1340., 790., 1344., 800.], [200.3, 100.7, 457.9, 388.1]])])
    rois = torch.cat([torch.cat([torch.full((len(boxes), 1), float(i)), boxes], 1) for i in range(2)]).to(d)
    scales = [0.25, 0.125, 0.0625, 0.03125]
    lib = _native.lib()
    try:
        for pooled in (14, 7):
            lib.seam_roi_align_set_lds(0)
            ref = ops.roi_align(feats, rois, scales, pooled)
            for mode in (1, 2):                     # row-staged tiles, whole-quadrant tiles
                lib.seam_roi_align_set_lds(mode)
                out = ops.roi_align(feats, rois, scales, pooled)
                assert torch.equal(out, ref), (mode, pooled, float((out.float() - ref.float()).abs().max()))
    finally:
        lib.seam_roi_align_set_lds(2)


def test_match_trunk_one_call_equals_the_launch_sequence(ops):
    """seam_match_trunk_f32: bit-identical to conv x4 + avg-pool + Linear/BN issued one by one, with the form of every
    conv chosen by the library (form = NULL) or handed down; and against the oracle's trunk."""
    import ctypes as C
    from seam_match_rcnn_amd import _native
    from seam_match_rcnn_amd.models.match_head import MatchPredictor
    d = dev()
    p = to_torch(synth.match_predictor_state(11))
    mp = MatchPredictor()
    mp.load_state_dict(p)
    mp = mp.to(d).eval()
    x = torch.from_numpy(synth.roi_features(61, 37))
    xn = nhwc(x).to(d)
    convs, lin = mp._packed_trunk()
    with torch.no_grad():
        y = xn
        for pc in convs:
            y = ops.conv2d(y, pc, relu=True)
        ref = ops.linear(ops.avgpool(y), lin, out_f32=True)
        one = ops.match_trunk(xn, convs, lin)
        via_module = mp.trunk_nhwc(xn)
    assert torch.equal(one, ref) and torch.equal(via_module, ref)
    assert_close(one, OH.match_trunk(x, p))
    # form = NULL: the C side picks the forms itself -- the same ones
    lib = _native.lib()
    layer = lambda pc: _native.TrunkLayer(*(None if t is None else t.data_ptr() for t in (pc.w, pc.u, pc.u24, pc.scale, pc.shift)))
    layers = (_native.TrunkLayer * 4)(*[layer(pc) for pc in convs])
    out = torch.empty((37, 256), device=d)
    ws = torch.empty((int(lib.seam_match_trunk_workspace_floats(37)),), device=d)
    rc = lib.seam_match_trunk_f32(xn.data_ptr(), layers, C.byref(layer(lin)), out.data_ptr(), 37, ws.data_ptr(), None,
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0 and torch.equal(out, ref)
    assert ops.match_trunk(xn[:0], convs, lin).shape == (0, 256)


def test_nlb_attnpool_golden_and_oracle(ops, golden):
    d = dev()
    from seam_match_rcnn_amd.models.match_head import pack_nlb_from_state
    p = to_torch(synth.temporal_aggregator_state(12))
    pk = pack_nlb_from_state({k: v.to(d) for k, v in p.items()})
    # golden Mode-B case: lens 10,1,4,7 ; time-major [11,4,256] with dummy row 0
    seq = rnd(33, (11, 4, 256), "seq")
    lens = golden["taB_lens"].tolist()
    seq[0] = 0
    for i, n in enumerate(lens):
        seq[n + 1:, i] = 0
    sd = seq.to(d)
    out, att = ops.nlb_attnpool(sd[1:], 4 * 256, 256, torch.tensor(lens, dtype=torch.int32, device=d), 4, 10, pk,
                                True, True)
    assert_close(out, torch.from_numpy(golden["taB_x3_1b"]))
    for i, n in enumerate(lens):
        assert_close(att[i, :n], torch.from_numpy(golden[f"taB_att{i}"])[:, 0])
    # long / ragged sequences vs the oracle (exercises the global-workspace path, T > 96)
    lens2 = [130, 2, 97, 33, 0, 1]
    tmax = max(lens2)
    x = rnd(50, (len(lens2), tmax, 256), "long")
    out2, att2 = ops.nlb_attnpool(x.to(d), 256, tmax * 256, torch.tensor(lens2, dtype=torch.int32, device=d),
                                  len(lens2), tmax, pk, True, True)
    for i, n in enumerate(lens2):
        if n == 0:
            assert float(out2[i].abs().max()) == 0.0
            continue
        ref, atts = OH.aggregate_sequences([x[i, :n]], p)
        assert_close(out2[i], ref[0])
        assert_close(att2[i, :n], atts[0][:, 0])


def test_nlb_attnpool_mfma_vs_valu_and_oracle(ops):
    """The matrix-core form of the block (G = X Wg, Y = f G, Z = Y Ww^T on v_mfma_f32_32x32x2_f32, theta / phi folded into two
    256-vectors) against the VALU kernel and the oracle: every row-tile boundary (1 ... 96 rows), both memory layouts
    (sequence-major and the reference's time-major x3_1_seq), z / attention outputs, use_nlb = 0 / 1 / 2."""
    d = dev()
    from seam_match_rcnn_amd.models.match_head import pack_nlb_from_state
    p = to_torch(synth.temporal_aggregator_state(12))
    pk = pack_nlb_from_state({k: v.to(d) for k, v in p.items()})
    lens = [1, 2, 7, 10, 31, 32, 33, 63, 64, 65, 95, 96, 0, 30]
    tmax = max(lens)
    assert tmax <= ops._native.lib().seam_nlb_mfma_max_len()
    x = rnd(51, (len(lens), tmax, 256), "mf")
    ld = torch.tensor(lens, dtype=torch.int32, device=d)
    for use in (1, 2, 0):
        res = {}
        for flag in (True, False):
            ops.NLB_MFMA = flag
            try:
                res[flag] = ops.nlb_attnpool(x.to(d), 256, tmax * 256, ld, len(lens), tmax, pk, use, True, True)
            finally:
                ops.NLB_MFMA = True
        (o1, a1, z1), (o0, a0, z0) = res[True], res[False]
        assert_close(o1, o0, rtol=1e-4, atol_scale=1e-5)
        assert_close(a1, a0, rtol=1e-4, atol_scale=1e-5)
        assert_close(z1, z0, rtol=1e-4, atol_scale=1e-5)
        if use == 1:
            for i, n in enumerate(lens):
                if n == 0:
                    assert float(o1[i].abs().max()) == 0.0
                    continue
                ref, atts = OH.aggregate_sequences([x[i, :n]], p)
                assert_close(o1[i], ref[0])
                assert_close(a1[i, :n], atts[0][:, 0])
    # time-major layout [T, S, 256] (what TemporalAggregationNLB hands over): t_stride = S*256, s_stride = 256
    s_, t_ = 37, 10
    xt = rnd(52, (t_, s_, 256), "tm")
    lt = torch.tensor([(i % t_) + 1 for i in range(s_)], dtype=torch.int32, device=d)
    o1, _ = ops.nlb_attnpool(xt.to(d), s_ * 256, 256, lt, s_, t_, pk)
    ops.NLB_MFMA = False
    try:
        o0, _ = ops.nlb_attnpool(xt.to(d), s_ * 256, 256, lt, s_, t_, pk)
    finally:
        ops.NLB_MFMA = True
    assert_close(o1, o0, rtol=1e-4, atol_scale=1e-5)
    ref, _ = OH.aggregate_sequences([xt[:int(lt[i]), i] for i in range(s_)], p)
    assert_close(o1, ref)


@pytest.mark.parametrize("qg", [(3, 5), (32, 1000), (70, 333), (256, 5000)])
def test_pair_logits_and_topk(ops, qg):
    q, g = qg
    d = dev()
    a, b = rnd(60, (q, 256), "a"), rnd(61, (g, 256), "b")
    w, bias = rnd(62, (2, 256), "w") / 16.0, rnd(63, (2,), "bias")
    ref = OH.pair_logits(a, b, w, bias)
    got = ops.pair_logits(a.to(d), b.to(d), w.to(d), bias.to(d))
    assert_close(got, ref)
    k = min(20, g)
    idx, sc = ops.rank_topk(got, k)
    ridx, rsc = OH.rank_topk(got.cpu(), k)          # same logits -> exact index parity (ties: lower index)
    assert torch.equal(idx.cpu(), ridx)
    assert_close(sc, rsc)


def test_c2_golden_topk(ops, golden):
    """BASELINE config 2 shapes (S=32,T=10,G=1000) against the reference-captured fixture."""
    d = dev()
    from seam_match_rcnn_amd.models.match_head import pack_nlb_from_state
    p = to_torch(synth.temporal_aggregator_state(12))
    pd = {k: v.to(d) for k, v in p.items()}
    pk = pack_nlb_from_state(pd)
    s, t, g = 32, 10, 1000
    seq = torch.from_numpy(synth.normal(synth.stream_id(35, "seq_c2"), (t, s, 256))).to(d)
    gal = torch.from_numpy(synth.gallery(36, g)).to(d)
    out, _ = ops.nlb_attnpool(seq, s * 256, 256, torch.full((s,), t, dtype=torch.int32, device=d), s, t, pk)
    assert_close(out, torch.from_numpy(golden["c2_x3_1b"]))
    x5 = ops.pair_logits(out, gal, pd["last.weight"], pd["last.bias"])
    assert_close(x5.reshape(-1)[::16], torch.from_numpy(golden["c2_x5_sample"]))
    idx, sc = ops.rank_topk(x5, 20)
    assert_close(sc, torch.from_numpy(golden["c2_top20_score"]))
    for qi in range(s):
        assert len(set(idx[qi].tolist()) ^ set(golden["c2_top20"][qi].tolist())) <= 2


def test_decode_nms_maskselect(ops):
    d = dev()
    n = 700
    ctr = torch.from_numpy(synth.uniform(synth.stream_id(70, "c"), (n, 2), 20, 380))
    wh = torch.from_numpy(synth.uniform(synth.stream_id(71, "wh"), (n, 2), 8, 120))
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    deltas = rnd(72, (n, 8), "d") * 0.5
    ref = OD.clip_boxes(OD.decode_boxes(deltas, boxes, (10., 10., 5., 5.)).view(n, 2, 4), (400, 416)).view(n, 8)
    got = ops.decode_boxes(deltas.to(d), boxes.to(d), (10., 10., 5., 5.), (400, 416))
    assert_close(got, ref, atol_scale=1e-6)
    scores = torch.from_numpy(synth.uniform(synth.stream_id(73, "s"), (n,)))
    order = torch.argsort(scores, descending=True, stable=True)
    for thr in (0.3, 0.5, 0.7):
        keep_ref = OD.nms(boxes, scores, thr)
        keep = ops.nms_sorted(boxes[order].to(d), thr).cpu().bool()
        assert torch.equal(order[keep], keep_ref)
    logits = rnd(74, (6, 14, 28, 28), "ml")
    labels = torch.tensor([1, 13, 0, 5, 5, 7])
    ref = OD.maskrcnn_inference(logits, [labels])[0]
    sub = logits.view(6, 14, 14, 2, 14, 2).permute(0, 2, 4, 3, 5, 1).reshape(6, 14, 14, 4 * 14).contiguous()
    assert_close(ops.mask_select(sub.to(d), labels.to(d), 14), ref, rtol=1e-5)


def test_batched_nms_and_paste_masks(ops):
    d = dev()
    from seam_match_rcnn_amd.models.detection import batched_nms_images
    b, n = 3, 900
    ctr = torch.from_numpy(synth.uniform(synth.stream_id(80, "c"), (b, n, 2), 20, 380))
    wh = torch.from_numpy(synth.uniform(synth.stream_id(81, "wh"), (b, n, 2), 4, 150))
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2)
    scores = torch.from_numpy(synth.uniform(synth.stream_id(82, "s"), (b, n)))
    cls = (torch.from_numpy(synth.uniform(synth.stream_id(83, "l"), (b, n))) * 5).to(torch.int64)
    valid = scores > 0.2
    valid[1, ::7] = False
    order, sel, exact = batched_nms_images(boxes.to(d), scores.to(d), cls.to(d), valid.to(d), 0.5, 100)
    assert bool(exact.all())
    # prefix form: with a prefix that reaches each image's last survivor the selection equals the full call's, and wherever
    # the call reports `exact` the prefix really held 100 survivors (or every valid candidate)
    last = int(max(int(torch.nonzero(sel[i]).max()) for i in range(b))) + 1
    order_p, sel_p, exact_p = batched_nms_images(boxes.to(d), scores.to(d), cls.to(d), valid.to(d), 0.5, 100, prefix=last)
    assert torch.equal(order_p, order) and torch.equal(sel_p, sel)
    for i in range(b):
        assert bool(exact_p[i]) == (int(sel[i].sum()) >= 100 or int(valid[i].sum()) <= last)
    # a prefix too short to hold 100 survivors that does not cover the valid candidates must say so
    _, sel_s, exact_s = batched_nms_images(boxes.to(d), scores.to(d), cls.to(d), valid.to(d), 0.5, 100, prefix=8)
    assert not bool(exact_s.any()) and int(sel_s[:, 8:].sum()) == 0
    for i in range(b):
        v = valid[i]
        ref = OD.batched_nms(boxes[i][v], scores[i][v], cls[i][v], 0.5)[:100]
        ref_idx = torch.nonzero(v).squeeze(1)[ref]
        got_idx = order[i][sel[i]].cpu()
        assert torch.equal(got_idx, ref_idx)
    # mask paste vs the oracle (boxes partly outside the image, tiny and large)
    k = 7
    masks = torch.from_numpy(synth.uniform(synth.stream_id(84, "m"), (k, 1, 28, 28)))
    bx = torch.tensor([[10.2, 12.7, 59.9, 80.1], [-15.5, -4.2, 30.0, 44.0], [100., 50., 180.5, 119.9],
                       [0., 0., 191.9, 127.9], [150.3, 100.1, 230.0, 160.0], [40.5, 40.5, 41.2, 41.0],
                       [5., 90., 120., 131.]])
    ref = OD.paste_masks_in_image(masks, bx, (128, 192))
    got = ops.paste_masks(masks.to(d), bx.to(d), (128, 192))
    assert_close(got, ref, rtol=1e-5, atol_scale=1e-6)


F16_CONV_CASES = [
    # N, C, H, W, K, R, stride, pad, res, relu, out_f32
    (2, 64, 20, 24, 64, 1, 1, 0, False, True, False),
    (1, 64, 30, 26, 256, 1, 1, 0, True, True, False),
    (2, 128, 17, 19, 128, 3, 1, 1, False, True, False),
    (2, 128, 18, 22, 256, 3, 2, 1, False, True, False),
    (1, 3, 64, 80, 64, 7, 2, 3, False, True, False),       # stem (3 -> 8 stored halves)
    (3, 256, 14, 14, 256, 3, 1, 0, False, True, False),    # match trunk: valid 3x3
    (1, 256, 13, 16, 15, 1, 1, 0, False, False, False),
    (5, 1024, 1, 1, 256, 1, 1, 0, False, False, True),     # trunk Linear -> fp32 descriptors
]


@pytest.mark.parametrize("case", F16_CONV_CASES)
def test_conv2d_f16(ops, case):
    """fp16 MFMA path (config 5): operands rounded to fp16, fp32 accumulation -> compare against the
    fp32 oracle evaluated on the SAME fp16-rounded operands (isolates the kernel from input rounding)."""
    n, c, h, w, k, r, stride, pad, res, relu, out_f32 = case
    x = rnd(1, (n, c, h, w)).half().float()
    wt = (rnd(2, (k, c, r, r), "w") * (1.0 / math.sqrt(c * r * r))).half().float()
    bias = rnd(3, (k,), "b") * 0.1
    ref = F.conv2d(x, wt, bias, stride, pad)
    resid = None
    if res:
        resid = rnd(8, ref.shape, "res").half().float()
        ref = ref + resid
    if relu:
        ref = F.relu(ref)
    d = dev()
    pc = ops.pack_conv(wt.to(d), bias.to(d), stride=stride, pad=pad, dtype=torch.float16)
    xin = nhwc(x)
    if c % 8:
        xin = F.pad(xin, (0, 8 - c % 8))
    y = ops.conv2d(xin.half().to(d), pc, relu, None if resid is None else nhwc(resid).half().to(d), out_f32=out_f32)
    assert y.dtype == (torch.float32 if out_f32 else torch.float16)
    # fp32 accumulate; the only extra error is the final fp16 rounding of the output (2^-11 relative)
    assert_close(y.float().permute(0, 3, 1, 2), ref, rtol=2e-3 if not out_f32 else 1e-3, atol_scale=1e-3)


def test_preprocess_uint8_frames(ops):
    """row f4 (device side): uint8 HWC frame -> ToTensor + normalise + resize + pad in one kernel."""
    d = dev()
    raw = [(torch.from_numpy(synth.uniform(synth.stream_id(90 + i, "u8"), (60, 90, 3))) * 256).to(torch.uint8) for i in range(2)]
    imgs = [r.permute(2, 0, 1).float().div(255) for r in raw]          # torchvision F.to_tensor semantics
    ref, sizes = OD.transform(imgs, min_size=96, max_size=160)
    out = ops.preprocess([r.to(d) for r in raw], [tuple(s) for s in sizes], ref.shape[-2], ref.shape[-1])
    assert_close(out[..., :3].permute(0, 3, 1, 2), ref, atol_scale=1e-5)
    ref1, sz1 = OD.transform(imgs, min_size=60, max_size=90)          # identity scale
    out1 = ops.preprocess([r.to(d) for r in raw], [tuple(s) for s in sz1], ref1.shape[-2], ref1.shape[-1])
    assert_close(out1[..., :3].permute(0, 3, 1, 2), ref1, atol_scale=1e-6)


def test_nms_scan_with_max_keep_is_the_truncated_full_scan(ops):
    """seam_nms_sorted_topn_f32: the first max_keep survivors of the full greedy scan, bit for bit, on sizes that cross the
    64-box block / 64-word lane-slot boundaries (n up to 13 000 = the 13-class x 1000-proposal candidate set)."""
    d = dev()
    for seed, (b, n, span) in enumerate([(2, 700, 380.0), (3, 4507, 800.0), (2, 13000, 800.0), (1, 16384, 1200.0), (2, 65, 60.0)]):
        ctr = torch.from_numpy(synth.uniform(synth.stream_id(900 + seed, "c"), (b, n, 2), 10, span))
        wh = torch.from_numpy(synth.uniform(synth.stream_id(910 + seed, "wh"), (b, n, 2), 4, 0.35 * span))
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 2).to(d)
        full = ops.nms_sorted(boxes, 0.5)
        assert 0 < int(full[0].sum()) < n
        for mk in (1, 7, 100, 1000, n + 5):
            got = ops.nms_sorted(boxes, 0.5, max_keep=mk)
            ref = full.bool() & (torch.cumsum(full, 1) <= mk)
            assert torch.equal(got.bool(), ref), (n, mk)
    # against the oracle's greedy NMS on one image
    ctr = torch.from_numpy(synth.uniform(synth.stream_id(930, "c"), (3000, 2), 10, 600.0))
    wh = torch.from_numpy(synth.uniform(synth.stream_id(931, "wh"), (3000, 2), 4, 200.0))
    boxes1 = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    scores = torch.linspace(1.0, 0.0, 3000)
    keep_ref = OD.nms(boxes1, scores, 0.5)
    got = ops.nms_sorted(boxes1.to(d), 0.5, max_keep=50).cpu().bool()
    assert torch.equal(torch.nonzero(got).squeeze(1), keep_ref[:50])


def test_rpn_topk_decode_matches_sort_and_oracle(ops):
    """seam_rpn_topk_decode_f32 (radix select + decode + clip + sigmoid, one launch per level) vs a stable descending sort
    + the decode kernel (bit-identical boxes, same order) and vs the oracle's per-level top-k / decode / clip."""
    d = dev()
    from seam_match_rcnn_amd.models.detection import grid_anchors
    n_img, a = 3, 3
    sizes = [(200, 304), (192, 300), (180, 250)]
    for lvl, (h, w, k) in enumerate([(50, 76, 1000), (25, 38, 1000), (13, 19, 741), (7, 10, 97), (40, 64, 1024), (1, 1, 3), (2, 3, 18), (1, 5, 1)]):
        head = rnd(950 + lvl, (n_img, h, w, 5 * a), "head")
        head[..., :a] *= 3.0
        if lvl == 1:        # heavy ties across the k-th position: quantised logits (lowest anchor index must win)
            head[..., :a] = torch.round(head[..., :a] * 2) / 2
        if lvl in (3, 6):
            head[1, ..., :a] = 0.25        # a constant row: the winners are anchors 0 .. k-1
        anc = torch.from_numpy(grid_anchors((208, 320), [(h, w)], sizes=(32 * 2 ** min(lvl, 4),))[0])
        clip = torch.tensor([[float(s[0]), float(s[1])] for s in sizes])
        ktot = k + 11
        bx = torch.full((n_img, ktot, 4), -7.0, device=d)
        sc = torch.full((n_img, ktot), -7.0, device=d)
        ix = torch.full((n_img, ktot), -7, dtype=torch.int64, device=d)
        hd = head.to(d)
        ops.rpn_topk_decode(hd, a, anc.to(d), clip.to(d), k, bx, sc, 5, ix)
        torch.cuda.synchronize()
        assert float(bx[:, :5].max()) == -7.0 and float(bx[:, 5 + k:].max()) == -7.0        # rows outside the level untouched
        obj = head[..., :a].reshape(n_img, -1)
        dlt = head[..., a:].reshape(n_img, -1, 4)
        top = torch.argsort(obj, dim=1, descending=True, stable=True)[:, :k]
        assert torch.equal(ix[:, 5:5 + k].cpu(), top), lvl
        for i in range(n_img):
            dd = dlt[i][top[i]]
            ref_dev = ops.decode_boxes(dd.contiguous().to(d), anc[top[i]].contiguous().to(d), (1., 1., 1., 1.), sizes[i])
            assert torch.equal(bx[i, 5:5 + k], ref_dev)
            ref = OD.clip_boxes(OD.decode_boxes(dd, anc[top[i]]), sizes[i])
            assert_close(bx[i, 5:5 + k], ref, atol_scale=1e-6)
            assert_close(sc[i, 5:5 + k], torch.sigmoid(obj[i][top[i]]), rtol=1e-5, atol_scale=1e-6)
    with pytest.raises(ValueError):
        ops.rpn_topk_decode(hd, a, anc.to(d), clip.to(d), 1025, bx, sc, 0)


def test_pointwise_kernel_1000_launches_bit_identical(ops):
    """every shipped epilogue mode of seam_conv1x1_sw_f32 -- none / ReLU / residual + ReLU / FPN top-down merge /
    two-source reduction -- launched 1000 times on a ragged map: every result bit-identical to the first (the kernel's waves are
    independent after the slab copy and its epilogue goes through a wave-private LDS transpose with the residual requested a step
    ahead: a missing wait or a shared LDS region would show as run-to-run differences, as it did in the reverted round-4 variant)."""
    d = dev()
    n, h, w, c, k = 3, 41, 53, 128, 256
    x, res = rnd(190, (n, c, h, w)), rnd(191, (n, k, h, w))
    top = rnd(192, (n, k, (h + 1) // 2, (w + 1) // 2))
    wgt = rnd(193, (k, c, 1, 1), "w") / (c ** 0.5)
    bn = (torch.from_numpy(synth.uniform(synth.stream_id(194, "bw"), (k,), 0.5, 1.5)), rnd(195, (k,), "bb") * 0.1,
          rnd(196, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(197, "rv"), (k,), 0.5, 1.5)))
    pc = ops.pack_conv(wgt.to(d), None, tuple(t.to(d) for t in bn))
    pcb = ops.pack_conv(wgt.to(d), rnd(198, (k,), "b").to(d))
    c1 = c // 2
    pcd = ops.pack_conv_dual(wgt[:, :c1].to(d), tuple(t.to(d) for t in bn), wgt[:, c1:].to(d), tuple(t.to(d) for t in bn))
    xd, rd, td = nhwc(x).to(d), nhwc(res).to(d), nhwc(top).to(d)
    xa, xb = nhwc(x[:, :c1]).contiguous().to(d), nhwc(x[:, c1:]).contiguous().to(d)
    cases = [("plain", lambda: ops.conv2d(xd, pc)), ("relu", lambda: ops.conv2d(xd, pc, relu=True)),
             ("residual", lambda: ops.conv2d(xd, pc, relu=True, residual=rd)), ("topdown", lambda: ops.conv2d_topdown(xd, pcb, td)),
             ("dual", lambda: ops.conv2d_dual(xa, xb, pcd, 1, relu=True))]
    saved = ops.SW
    try:
        ops.SW = True
        assert pc.ws is not None and h * w >= ops.SW_MIN_HW
        for name, run in cases:
            first = run()
            bad = torch.zeros((), dtype=torch.int64, device=d)
            for _ in range(1000):
                bad += (run() != first).sum()
            assert int(bad) == 0, (name, int(bad))
    finally:
        ops.SW = saved


def test_pointwise_unserved_channel_counts_fall_back(ops):
    """K = 768 (slab count 3 does not divide an XCD's 32 blocks) is not given to seam_conv1x1_sw_f32 (it would have redone
    tiles; K = 8448 would have hung): the C ABI rejects it, pack_conv prepares no slab for it, conv2d runs the implicit GEMM."""
    d = dev()
    from seam_match_rcnn_amd import _native
    lib = _native.lib()
    x, wgt, bias = rnd(70, (2, 64, 20, 20)), rnd(71, (768, 64, 1, 1), "w") / 8.0, rnd(72, (768,), "b")
    pc = ops.pack_conv(wgt.to(d), bias.to(d))
    assert pc.ws is None and lib.seam_conv1x1_sw_config(800, 64, 0, 768) == 0
    xd = nhwc(x).to(d)
    y = torch.empty((2, 20, 20, 768), device=d)
    rc = lib.seam_conv1x1_sw_f32(xd.data_ptr(), None, wgt.to(d).reshape(768, 64).contiguous().data_ptr(), bias.to(d).data_ptr(), None, y.data_ptr(),
                                 800, 64, 0, 768, 0, 0, 0, 0, 0, 0, torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    assert_close(ops.conv2d(xd, pc, relu=True).permute(0, 3, 1, 2), F.relu(F.conv2d(x, wgt, bias)))


@pytest.mark.parametrize("shape", [(2, 48, 72, 256, 128), (2, 48, 72, 256, 256), (3, 40, 56, 64, 256), (1, 14, 14, 256, 1024),
                                   (5, 33, 31, 128, 512), (2, 20, 20, 256, 64), (4, 50, 50, 64, 64)])
def test_pointwise_weights_stationary_kernel(ops, shape):
    """seam_conv1x1_sw_f32 (csrc/seam_pw.hip) on ragged maps, every epilogue mode -- none / ReLU / residual + ReLU / FPN top-down
    merge / two-source reduction -- against the torch reference (1e-3 contract) and against the implicit GEMM on the same inputs
    (two exact-fp32 chains: <= 2e-5 of scale apart); five repetitions of every launch must be bit-identical (the kernel has no
    barriers after its prologue: a race would show as run-to-run differences)."""
    d = dev()
    n, h, w, c, k = shape
    x, res = rnd(90, (n, c, h, w)), rnd(91, (n, k, h, w))
    top = rnd(92, (n, k, (h + 1) // 2, (w + 1) // 2))
    wgt = rnd(93, (k, c, 1, 1), "w") / (c ** 0.5)
    bn = (torch.from_numpy(synth.uniform(synth.stream_id(94, "bw"), (k,), 0.5, 1.5)), rnd(95, (k,), "bb") * 0.1,
          rnd(96, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(97, "rv"), (k,), 0.5, 1.5)))
    sc = bn[0] * (bn[3] + 1e-5).rsqrt()
    conv = F.conv2d(x, wgt) * sc[None, :, None, None] + (bn[1] - bn[2] * sc)[None, :, None, None]
    pc = ops.pack_conv(wgt.to(d), None, tuple(t.to(d) for t in bn))
    assert pc.ws is not None and h * w >= ops.SW_MIN_HW
    bias = rnd(98, (k,), "b")
    pcb = ops.pack_conv(wgt.to(d), bias.to(d))
    xd, rd, td = nhwc(x).to(d), nhwc(res).to(d), nhwc(top).to(d)
    cases = [("plain", lambda: ops.conv2d(xd, pc), conv),
             ("relu", lambda: ops.conv2d(xd, pc, relu=True), F.relu(conv)),
             ("residual", lambda: ops.conv2d(xd, pc, relu=True, residual=rd), F.relu(conv + res)),
             ("topdown", lambda: ops.conv2d_topdown(xd, pcb, td),
              F.conv2d(x, wgt, bias) + F.interpolate(top, size=(h, w), mode="nearest"))]
    if c % 64 == 0:
        c1 = c // 2
        pcd = ops.pack_conv_dual(wgt[:, :c1].to(d), tuple(t.to(d) for t in bn), wgt[:, c1:].to(d), tuple(t.to(d) for t in bn))
        xa, xb = nhwc(x[:, :c1]).contiguous().to(d), nhwc(x[:, c1:]).contiguous().to(d)
        sh2 = 2 * (bn[1] - bn[2] * sc)
        cases.append(("dual", lambda: ops.conv2d_dual(xa, xb, pcd, 1, relu=True),
                      F.relu(F.conv2d(x, wgt) * sc[None, :, None, None] + sh2[None, :, None, None])))
    saved = ops.SW
    try:
        for name, run, ref in cases:
            ops.SW = True
            got = run()
            assert_close(got.permute(0, 3, 1, 2), ref)
            for _ in range(4):
                assert torch.equal(run(), got), name
            ops.SW = False
            other = run()
            assert float((got - other).abs().max()) <= 2e-5 * float(other.abs().max()), name
    finally:
        ops.SW = saved


@pytest.mark.parametrize("case", [
    # n, C, H, W, K, relu, bn, residual                (M = n H W: ragged against the 128-pixel tile, one case below it)
    (2, 512, 25, 25, 2048, True, True, True), (1, 1024, 50, 37, 256, True, True, False), (3, 2048, 13, 11, 512, False, False, False),
    (2, 512, 41, 29, 128, True, False, True), (1, 384, 9, 9, 256, False, True, False), (1, 256 + 128, 64, 40, 640, True, True, True),
])
def test_pointwise_pc_kernel(ops, case):
    """seam_conv1x1_pc_f32 (producer / consumer pointwise kernel, long reductions) == torch's fp32 convolution at the
    tolerance of the other exact-fp32 kernels, == the implicit GEMM BIT FOR BIT (the same k order; ADVICE r5: the header's claim is now what the test asserts), the same bits from launch to launch,
    and the same bits for an image alone as inside a batch (an output is one wave's fixed fma chain)."""
    d = dev()
    n, c, h, w, k, relu, use_bn, use_res = case
    x = rnd(300, (n, c, h, w))
    wgt = rnd(301, (k, c, 1, 1), "w") / (c ** 0.5)
    res = rnd(302, (n, k, h, w)) if use_res else None
    if use_bn:
        bn = (torch.from_numpy(synth.uniform(synth.stream_id(303, "bw"), (k,), 0.5, 1.5)), rnd(304, (k,), "bb") * 0.1,
              rnd(305, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(306, "rv"), (k,), 0.5, 1.5)))
        pc = ops.pack_conv(wgt.to(d), None, tuple(t.to(d) for t in bn))
        sc = bn[0] * (bn[3] + 1e-5).rsqrt()
        ref = F.conv2d(x, wgt) * sc[None, :, None, None] + (bn[1] - bn[2] * sc)[None, :, None, None]
    else:
        b = rnd(303, (k,), "b")
        pc = ops.pack_conv(wgt.to(d), b.to(d))
        ref = F.conv2d(x, wgt, b)
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    assert pc.wq is not None and pc.ws is None
    xd = nhwc(x).to(d)
    rd = nhwc(res).to(d) if use_res else None
    saved, ops.CONV_TRACE = (ops.PWPC, ops.CONV_TRACE, ops.PWPC_MIN_HW), []
    try:
        ops.PWPC, ops.PWPC_MIN_HW = True, 64
        got = ops.conv2d(xd, pc, relu=relu, residual=rd)
        assert [t[0] for t in ops.CONV_TRACE] == ["conv1x1_pc"]
        ops.CONV_TRACE = None
        assert_close(got.permute(0, 3, 1, 2), ref)
        for _ in range(20):
            assert torch.equal(ops.conv2d(xd, pc, relu=relu, residual=rd), got)
        one = ops.conv2d(xd[n - 1:], pc, relu=relu, residual=None if rd is None else rd[n - 1:])
        assert torch.equal(one[0], got[n - 1])
        ops.PWPC = False
        ig = ops.conv2d(xd, pc, relu=relu, residual=rd)
        assert torch.equal(ig, got), "conv1x1_pc and the implicit GEMM accumulate in the same k order: the same bits (seam_pwpc.hip header)"
    finally:
        ops.PWPC, ops.CONV_TRACE, ops.PWPC_MIN_HW = saved
