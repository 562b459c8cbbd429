"""GPU parity of the fp16-MFMA / fp32-accumulate path (BASELINE config 5).  Kernel-level tests feed the
fp32 oracle the same fp16-rounded operands; the model-level test compares against the fp32 oracle with a
norm-wise bound that reflects fp16 storage of every activation (2^-11 per rounding, ~50 layers)."""
import math

import pytest
import torch
import torch.nn.functional as F

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import detection as OD
from oracle import heads as OH
from oracle import model as OM
from test_gpu_ops import assert_close, dev, nhwc, rnd

pytestmark = pytest.mark.gpu
H = torch.float16


@pytest.fixture(scope="module")
def ops():
    import seam_match_rcnn_amd.ops as ops
    return ops


def test_elementwise_f16(ops):
    d = dev()
    x = rnd(30, (2, 64, 31, 37)).half()
    got = ops.maxpool2d(nhwc(x).to(d), 3, 2, 1)
    assert got.dtype == H
    assert torch.equal(got.permute(0, 3, 1, 2).float().cpu(), F.max_pool2d(x.float(), 3, 2, 1))
    lat, top = rnd(31, (2, 32, 26, 34)).half(), rnd(32, (2, 32, 13, 17)).half()
    ref = (lat.float() + F.interpolate(top.float(), size=lat.shape[-2:], mode="nearest")).half()
    got = ops.upsample_add_(nhwc(lat).to(d), nhwc(top).to(d))
    assert torch.equal(got.permute(0, 3, 1, 2).cpu(), ref)
    r = rnd(33, (5, 256, 14, 14))
    t = ops.nchw_to_nhwc(r.to(d), H)
    assert t.dtype == H and torch.equal(t.cpu(), nhwc(r).half())
    assert torch.equal(ops.nhwc_to_nchw(t).cpu(), r.half().float())
    a = rnd(34, (7, 1024, 6, 6)).half()
    assert_close(ops.avgpool(nhwc(a).to(d)).float(), F.avg_pool2d(a.float(), 6).flatten(1), rtol=1e-3)
    imgs = [torch.from_numpy(synth.uniform(synth.stream_id(20 + i, "img"), (3, 60, 90))) for i in range(2)]
    ref, sizes = OD.transform(imgs, min_size=96, max_size=160)
    out = ops.preprocess([i.to(d) for i in imgs], [tuple(s) for s in sizes], ref.shape[-2], ref.shape[-1], H)
    assert out.shape[-1] == 8 and float(out[..., 3:].abs().max()) == 0.0
    assert_close(out[..., :3].permute(0, 3, 1, 2).float(), ref, rtol=1e-3, atol_scale=1e-3)


def test_roi_align_and_mask_select_f16(ops):
    d = dev()
    sizes = [(200, 200)] * 2
    feats = [rnd(40 + i, (2, 256, s, s)).half() for i, s in enumerate((50, 25, 13, 7))]
    boxes = [torch.tensor([[10., 12., 60., 80.], [0., 0., 200., 200.], [150., 20., 199., 70.], [-5., -8., 40., 30.]]),
             torch.from_numpy(synth.fixed_rois(8, 200, 200))]
    ref = OD.multiscale_roi_align([f.float() for f in feats], boxes, sizes, 14)
    rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i)), b], 1) for i, b in enumerate(boxes)])
    scales = OD.infer_scales([f.shape[-2:] for f in feats], sizes)
    out = ops.roi_align([nhwc(f).to(d) for f in feats], rois.to(d), scales, 14)
    assert out.dtype == H
    assert_close(out.permute(0, 3, 1, 2).float(), ref, rtol=2e-3, atol_scale=1e-3)
    logits = rnd(74, (6, 14, 28, 28), "ml").half()
    labels = torch.tensor([1, 13, 0, 5, 5, 7])
    ref = OD.maskrcnn_inference(logits.float(), [labels])[0]
    sub = logits.view(6, 14, 14, 2, 14, 2).permute(0, 2, 4, 3, 5, 1).reshape(6, 14, 14, 4 * 14).contiguous()
    assert_close(ops.mask_select(sub.to(d), labels.to(d), 14), ref, rtol=1e-5)


def test_fixed_roi_forward_fp16_vs_fp32_oracle():
    """Whole extractor + trunks in fp16 storage / fp16 MFMA (fp32 accumulate), descriptors fp32."""
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    m = m.to(dev()).eval().set_compute_dtype(torch.float16)
    m.transform.min_size, m.transform.max_size = 256, 320
    imgs = [torch.from_numpy(synth.frames(7 + i, 1, 256, 320)[0]) for i in range(2)]
    rois = [torch.from_numpy(synth.fixed_rois(8, 256, 320))] * 2
    with torch.no_grad():
        res, feats, _ = m.forward_fixed_rois([i.to(dev()) for i in imgs], rois)
    assert feats["0"].dtype == H and res[0]["roi_features"].dtype == torch.float32
    batch, sizes = OD.transform(imgs, 256, 320)
    ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
    orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], rois, sizes, 14)
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    ox3 = OH.match_trunk(orf, mp)

    def rel(a, b):      # norm-wise relative error
        a, b = a.float().cpu(), b.float().cpu()
        return float((a - b).norm() / b.norm())

    assert rel(feats["0"].permute(0, 3, 1, 2), ofe["0"]) < 5e-3
    assert rel(feats["3"].permute(0, 3, 1, 2), ofe["3"]) < 5e-3
    assert rel(torch.cat([r["roi_features"] for r in res]), orf) < 5e-3
    x3 = torch.cat([r["match_features"] for r in res])
    assert x3.dtype == torch.float32 and rel(x3, ox3) < 1e-2
    # SEAM head on top (fp32 heads on fp16-trunk descriptors): logits within 2 % norm-wise of the fp32 oracle
    ta = m.roi_heads.temporal_aggregator
    x = torch.cat([r["roi_features"] for r in res])
    types = torch.IntTensor([0] * 8 + [1] * 8)
    ids = torch.LongTensor([0, 1, 0, 1, 0, 1, 0, 1] + [0] * 8)
    out = ta(x, types, ids)
    ref = OH.temporal_aggregation_forward(orf, types, ids, OM.sub(sd, "roi_heads.temporal_aggregator."))
    assert rel(out[0], ref[0]) < 1e-2 and rel(out[2], ref[2]) < 2e-2
    # and the fp32 mode of the same model object is still exact
    m.set_compute_dtype(torch.float32)
    with torch.no_grad():
        res32, feats32, _ = m.forward_fixed_rois([i.to(dev()) for i in imgs], rois)
    assert_close(torch.cat([r["match_features"] for r in res32]), ox3)


@pytest.mark.parametrize("shape", [(3, 64, 20, 24, 64, 1, 256), (2, 128, 13, 17, 256, 2, 512), (1, 512, 25, 25, 1024, 2, 2048)])
def test_conv_dual_source_f16(ops, shape):
    """fp16 twin of the dual-source shortcut GEMM: bn3(conv3(h)) + bn_d(conv_d(x)) + ReLU on fp16 operands, fp32 accumulate.
    Reference: fp32 on the fp16-rounded activations with the FOLDED weights rounded to fp16 (what the pack stores)."""
    d = dev()
    n, c1, ho, wo, c2, s2, k = shape
    h = rnd(170, (n, c1, ho, wo)).half().float()
    x = rnd(171, (n, c2, ho * s2 - (s2 - 1), wo * s2 - (s2 - 1))).half().float()
    w3, wd = rnd(172, (k, c1, 1, 1), "w3") / (c1 ** 0.5), rnd(173, (k, c2, 1, 1), "wd") / (c2 ** 0.5)

    def bn(seed):
        return (torch.from_numpy(synth.uniform(synth.stream_id(seed, "bw"), (k,), 0.5, 1.5)), rnd(seed + 1, (k,), "bb") * 0.1,
                rnd(seed + 2, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(seed + 3, "rv"), (k,), 0.5, 1.5)))
    b3, bd = bn(174), bn(178)

    def fold(w, b):
        sc = b[0] * (b[3] + 1e-5).rsqrt()
        return (w * sc[:, None, None, None]).half().float(), b[1] - b[2] * sc
    w3f, t3 = fold(w3, b3)
    wdf, td = fold(wd, bd)
    ref = F.relu(F.conv2d(h, w3f) + F.conv2d(x, wdf, None, s2) + (t3 + td)[None, :, None, None])
    pc = ops.pack_conv_dual(w3.to(d), tuple(t.to(d) for t in b3), wd.to(d), tuple(t.to(d) for t in bd), dtype=H)
    got = ops.conv2d_dual(nhwc(h).half().to(d), nhwc(x).half().to(d), pc, s2, relu=True)
    assert got.dtype == H
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)


def test_vector_epilogue_f16_tails_and_residual(ops):
    """The 16-byte fp16 epilogue (K % 8 == 0) on ragged row counts and K that is not a multiple of the tile width, with
    residual + ReLU, against the element-wise epilogue's contract (fp32 oracle on fp16-rounded operands)."""
    d = dev()
    for seed, (n, c, hh, ww, k, res) in enumerate([(1, 64, 9, 11, 72, True), (2, 128, 5, 7, 200, False), (1, 256, 33, 3, 1024, True),
                                                  (3, 64, 1, 1, 8, True)]):
        x = rnd(190 + seed, (n, c, hh, ww)).half().float()
        wt = (rnd(195 + seed, (k, c, 1, 1), "w") / math.sqrt(c)).half().float()
        bias = rnd(199 + seed, (k,), "b") * 0.1
        ref = F.conv2d(x, wt, bias)
        resid = rnd(205 + seed, ref.shape, "r").half().float() if res else None
        if res:
            ref = ref + resid
        ref = F.relu(ref)
        pc = ops.pack_conv(wt.to(d), bias.to(d), dtype=H)
        y = ops.conv2d(nhwc(x).half().to(d), pc, True, None if resid is None else nhwc(resid).half().to(d))
        assert y.dtype == H
        assert_close(y.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)


def _f16pc_case(ops, seed, n, c, hh, ww, k, pad, relu, bn):
    d = dev()
    x = rnd(seed, (n, c, hh, ww)).half().float()
    wt = (rnd(seed + 1, (k, c, 3, 3), "w") / (3 * math.sqrt(c)))
    if bn:
        b = (torch.from_numpy(synth.uniform(synth.stream_id(seed + 2, "bw"), (k,), 0.5, 1.5)), rnd(seed + 3, (k,), "bb") * 0.1,
             rnd(seed + 4, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(seed + 5, "rv"), (k,), 0.5, 1.5)))
        pc = ops.pack_conv(wt.to(d), None, stride=1, pad=pad, bn=tuple(t.to(d) for t in b), dtype=H)
    else:
        b = rnd(seed + 2, (k,), "b") * 0.1
        pc = ops.pack_conv(wt.to(d), b.to(d), stride=1, pad=pad, dtype=H)
    return x, wt, b, pc


def _f16pc_ref(x, wt, b, pc, pad, relu, bn):
    """fp32 convolution of the fp16-rounded operands the pack stores (scale / shift stay fp32 in the epilogue)."""
    if bn:
        sc = b[0] * (b[3] + 1e-5).rsqrt()
        folded = pc.scale is None                      # the pack either folds the BN scale into the weights or keeps it in the epilogue
        w16 = ((wt * sc[:, None, None, None]) if folded else wt).half().float()
        ref = F.conv2d(x, w16, None, 1, pad)
        ref = (ref if folded else ref * sc[None, :, None, None]) + (b[1] - b[2] * sc)[None, :, None, None]
    else:
        ref = F.conv2d(x, wt.half().float(), b, 1, pad)
    return F.relu(ref) if relu else ref


@pytest.mark.parametrize("case", [
    # n, C, H, W, K, pad, relu, bn            mode 0: 8 x 32 output patches, ragged right / bottom edges, both paddings
    (3, 128, 27, 45, 128, 1, True, True), (2, 256, 40, 70, 256, 0, False, False), (1, 128, 9, 26, 256, 1, True, False),
    # mode 1: whole small maps, G per block, a partial last block (n % G != 0), every supported patch width
    (37, 256, 14, 14, 256, 1, True, False), (23, 256, 8, 8, 1024, 0, True, False), (9, 256, 12, 12, 256, 0, False, True),
    (11, 128, 10, 10, 128, 0, True, False), (5, 128, 14, 14, 128, 0, True, False), (4, 256, 16, 16, 128, 0, True, True),
    (7, 128, 6, 6, 128, 1, True, False),
    # C = K = 64 (conv3x3_f16pc64: one chunk per tile, weights stationary in registers): ragged edges, both paddings, 1 .. many tiles per block
    (2, 64, 27, 45, 64, 1, True, True), (1, 64, 40, 70, 64, 0, False, False), (3, 64, 9, 26, 64, 1, True, False), (5, 64, 96, 168, 64, 1, True, True),
])
def test_conv3x3_f16pc_matches_fp32_on_fp16_operands(ops, case):
    """seam_conv3x3_f16pc (producer / consumer fp16 3x3) against the fp32 convolution of the same fp16-rounded operands -- the bound of
    the implicit-GEMM tests above -- and against conv_igemm<_Float16> to accumulation-order rounding."""
    import seam_match_rcnn_amd._native as native
    n, c, hh, ww, k, pad, relu, bn = case
    assert native.lib().seam_conv3x3_f16pc_supported(n, hh, ww, c, k, pad) == 1
    x, wt, b, pc = _f16pc_case(ops, 900, n, c, hh, ww, k, pad, relu, bn)
    assert pc.wh is not None
    ref = _f16pc_ref(x, wt, b, pc, pad, relu, bn)
    xd = nhwc(x).half().to(dev())
    old = ops.F16PC, ops.F16PC_RULE, ops.CONV_TRACE
    try:
        ops.F16PC_RULE = False
        ops.F16PC = True
        ops.CONV_TRACE = []
        got = ops.conv2d(xd, pc, relu)
        assert [t[0] for t in ops.CONV_TRACE] == ["conv3x3_f16pc"]
        ops.CONV_TRACE = None
        ops.F16PC = False
        ig = ops.conv2d(xd, pc, relu)
    finally:
        ops.F16PC, ops.F16PC_RULE, ops.CONV_TRACE = old
    assert got.dtype == H and got.shape == ig.shape
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)
    scale = float(ref.abs().max())
    assert float((got.float() - ig.float()).abs().max()) <= 2e-3 * scale        # two fp16 roundings of nearly equal fp32 sums


def test_conv3x3_f16pc_many_tiles_per_block_is_repeatable(ops):
    """The persistent walk (several tiles per block: the cross-tile patch pipeline, the finished tile laid over the free patch buffer and
    drained by the producer waves beside the next tile's first chunk): five launches bit-identical, equal to the implicit GEMM to
    rounding on EVERY image, no output left unwritten.  (The drain once raced the next tile's patch stores: a few wrong pixels in
    one launch of three, on exactly these shapes.)"""
    d = dev()
    old = ops.F16PC, ops.F16PC_RULE
    try:
        ops.F16PC_RULE = False
        for seed, (n, c, hh, ww, k, pad) in enumerate([(1536, 256, 12, 12, 256, 0), (1536, 256, 8, 8, 1024, 0), (24, 256, 96, 168, 256, 1),
                                                       (24, 64, 192, 336, 64, 1), (7, 64, 50, 75, 64, 0)]):
            g = torch.Generator(device=d); g.manual_seed(40 + seed)
            x = torch.randn(n, hh, ww, c, device=d, generator=g).half()
            wt = torch.randn(k, c, 3, 3, device=d, generator=g) / (3 * math.sqrt(c))
            bias = torch.randn(k, device=d, generator=g) * 0.1
            pc = ops.pack_conv(wt, bias, stride=1, pad=pad, dtype=H)
            ops.F16PC = False
            ig = ops.conv2d(x, pc, True)
            ops.F16PC = True
            first = None
            for rep in range(5):
                y = torch.full_like(ig, 777.0)
                ops.conv2d(x, pc, True, out=y)
                assert int((y == 777.0).sum()) == 0
                if first is None:
                    first = y
                    scale = float(ig.float().abs().max())
                    assert float((y.float() - ig.float()).abs().max()) <= 2e-3 * scale
                else:
                    assert torch.equal(y, first), f"launch {rep} differs"
    finally:
        ops.F16PC, ops.F16PC_RULE = old


@pytest.mark.parametrize("case", [
    # n, C, H, W, K, relu, bn      M ragged against the 256-row tiles, 4 .. 16 chunks, 1 .. 4 n-tiles, one tile .. many tiles per block
    (3, 512, 27, 45, 128, True, True), (2, 1024, 40, 70, 256, True, False), (1, 2048, 9, 26, 512, False, False),
    (5, 768, 33, 31, 384, True, True), (240, 1024, 12, 21, 256, True, True), (1, 512, 14, 14, 2048, False, True),
])
def test_conv1x1_f16pc_matches_fp32_on_fp16_operands(ops, case):
    """seam_conv1x1_f16pc (producer / consumer fp16 1x1 with a long reduction, round 6) against the fp32 product of the same fp16-rounded
    operands and against conv_igemm<_Float16> / conv1x1_swh to accumulation-order rounding; twice onto a poisoned output, bit for bit."""
    import seam_match_rcnn_amd._native as native
    n, c, hh, ww, k, relu, bn = case
    d = dev()
    assert native.lib().seam_conv1x1_f16pc_supported(n * hh * ww, c, k) == 1
    x = rnd(700, (n, c, hh, ww)).half().float()
    wt = rnd(701, (k, c, 1, 1), "w") / math.sqrt(c)
    if bn:
        b = (torch.from_numpy(synth.uniform(synth.stream_id(702, "bw"), (k,), 0.5, 1.5)), rnd(703, (k,), "bb") * 0.1,
             rnd(704, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(705, "rv"), (k,), 0.5, 1.5)))
        pc = ops.pack_conv(wt.to(d), None, stride=1, pad=0, bn=tuple(t.to(d) for t in b), dtype=H)
        sc = b[0] * (b[3] + 1e-5).rsqrt()
        folded = pc.scale is None
        w16 = ((wt * sc[:, None, None, None]) if folded else wt).half().float()
        ref = F.conv2d(x, w16)
        ref = (ref if folded else ref * sc[None, :, None, None]) + (b[1] - b[2] * sc)[None, :, None, None]
    else:
        bias = rnd(702, (k,), "b") * 0.1
        pc = ops.pack_conv(wt.to(d), bias.to(d), stride=1, pad=0, dtype=H)
        ref = F.conv2d(x, wt.half().float(), bias)
    ref = F.relu(ref) if relu else ref
    assert pc.wph is not None
    xd = nhwc(x).half().to(d)
    old = ops.PWHPC, ops.CONV_TRACE, ops.PWHPC_MIN_C
    try:
        ops.PWHPC = True
        ops.PWHPC_MIN_C = 512
        ops.CONV_TRACE = []
        got = torch.full((n, hh, ww, k), 777.0, dtype=H, device=d)
        ops.conv2d(xd, pc, relu, out=got)
        assert [t[0] for t in ops.CONV_TRACE] == ["conv1x1_f16pc"]
        ops.CONV_TRACE = None
        again = torch.full((n, hh, ww, k), -333.0, dtype=H, device=d)
        ops.conv2d(xd, pc, relu, out=again)
        ops.PWHPC = False
        other = ops.conv2d(xd, pc, relu)
    finally:
        ops.PWHPC, ops.CONV_TRACE, ops.PWHPC_MIN_C = old
    assert torch.equal(got, again)
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)
    scale = float(ref.abs().max())
    assert float((got.float() - other.float()).abs().max()) <= 2e-3 * scale


def test_conv1x1_f16pc_refuses_what_it_does_not_serve(ops):
    import seam_match_rcnn_amd._native as native
    lib = native.lib()
    sup = lib.seam_conv1x1_f16pc_supported
    assert sup(1000, 512, 128) == 1 and sup(1, 2048, 2048) == 1
    assert sup(1000, 256, 128) == 0 and sup(1000, 640, 128) == 0 and sup(1000, 512, 64) == 0 and sup(0, 512, 128) == 0
    d = dev()
    z = torch.zeros(1 << 16, dtype=H, device=d)
    assert lib.seam_conv1x1_f16pc(z.data_ptr(), z.data_ptr(), None, None, z.data_ptr(), z.data_ptr(), 64, 512, 128, 1, None) != 0     # residual
    assert lib.seam_conv1x1_f16pc(z.data_ptr(), z.data_ptr(), None, None, None, z.data_ptr(), 64, 384, 128, 1, None) != 0             # C


def test_conv3x3_f16pc_refuses_a_residual_and_dispatch_rule(ops):
    import seam_match_rcnn_amd._native as native
    lib = native.lib()
    d = dev()
    x = torch.zeros(2, 16, 16, 128, dtype=H, device=d)
    pc = ops.pack_conv(torch.zeros(128, 128, 3, 3, device=d), torch.zeros(128, device=d), stride=1, pad=1, dtype=H)
    y = torch.zeros(2, 16, 16, 128, dtype=H, device=d)
    rc = lib.seam_conv3x3_f16pc(x.data_ptr(), pc.wh.data_ptr(), None, pc.shift.data_ptr(), y.data_ptr(), y.data_ptr(), 2, 16, 16, 128, 128, 1, 1, None)
    assert rc != 0
    # a residual layer stays on the implicit GEMM
    ops.CONV_TRACE = []
    try:
        ops.conv2d(x, pc, True, y.clone())
        assert ops.CONV_TRACE[0][0].startswith("conv_igemm")
    finally:
        ops.CONV_TRACE = None
    # the dispatch rule: full tiles pay, half-empty ones do not
    assert lib.seam_conv3x3_f16pc_pays(48, 192, 336, 256, 256, 1) == 1
    assert lib.seam_conv3x3_f16pc_pays(1536, 14, 14, 256, 256, 1) == 1
    assert lib.seam_conv3x3_f16pc_pays(1536, 14, 14, 256, 256, 0) == 1      # 144 slots of a 160-slot tile
    assert lib.seam_conv3x3_f16pc_pays(48, 24, 42, 512, 512, 1) == 0        # 42 columns in 64
    assert lib.seam_conv3x3_f16pc_pays(1, 192, 336, 256, 256, 1) == 1       # the batch size takes no part
    assert lib.seam_conv3x3_f16pc_pays(240, 192, 336, 64, 64, 1) == 1       # layer1's 3x3 layers (conv3x3_f16pc64)
    assert lib.seam_conv3x3_f16pc_supported(64, 14, 14, 64, 64, 1) == 0     # ... on large maps only
    assert lib.seam_conv3x3_f16pc_supported(8, 64, 64, 64, 128, 1) == 0 and lib.seam_conv3x3_f16pc_supported(8, 64, 64, 128, 64, 1) == 0


@pytest.mark.parametrize("hw", [(64, 96), (70, 100)])
def test_stem_space_to_depth_f16(ops, hw):
    """fp16 twin of tests/test_gpu_ops.py::test_stem_space_to_depth: preprocess -> [N,H/2,W/2,16] fp16 (12 channels + 4 zeros), the stem
    as a 4x4 / stride-1 conv with re-indexed weights and a cropped output grid == the oracle's transform + 7x7 / stride-2 conv +
    FrozenBN + ReLU to fp16 accuracy, and == the 8-channel fp16 form to its rounding."""
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
    x16 = ops.preprocess(imgs, sz, hp, wp, H, s2d=True)
    assert x16.shape == (3, hp // 2, wp // 2, 16) and x16.dtype == H
    x8 = ops.preprocess(imgs, sz, hp, wp, H)
    back = x16[..., :12].reshape(3, hp // 2, wp // 2, 2, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(3, hp, wp, 3)
    assert torch.equal(back, x8[..., :3]) and float(x16[..., 12:].abs().max()) == 0.0
    w8 = F.pad(wt, (1, 0, 1, 0))
    ws = w8.view(64, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(64, 12, 4, 4).contiguous()
    bn_d = tuple(t.to(d) for t in bnp)
    pc16 = ops.pack_conv(ws.to(d), None, bn_d, stride=1, pad=2, cstore=16, wino=False, dtype=H)
    got = ops.conv2d(x16, pc16, relu=True, out_hw=(hp // 2, wp // 2))
    assert got.dtype == H
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=4e-3, atol_scale=2e-3)
    pc8 = ops.pack_conv(wt.to(d), None, bn_d, stride=2, pad=3, cstore=8, dtype=H)
    old = ops.conv2d(x8, pc8, relu=True)
    assert old.shape == got.shape and float((old.float() - got.float()).abs().max()) <= 2e-3 * float(old.float().abs().max())


@pytest.mark.parametrize("case", [
    # n, C, H, W, K, relu, bn, residual        wave tiles <1,8> (C <= 256, K % 256 == 0), <2,4> (C <= 512, K % 128 == 0), <4,2> (C <= 128)
    (3, 64, 41, 53, 256, True, True, True), (2, 256, 33, 31, 1024, True, True, True), (2, 256, 48, 72, 256, True, False, False),
    (2, 128, 40, 56, 512, False, True, False), (3, 512, 25, 42, 128, True, True, False), (2, 512, 24, 42, 256, True, False, True),
    (2, 384, 24, 42, 256, True, True, False), (1, 512, 15, 17, 512, False, False, True), (2, 128, 20, 20, 64, True, True, True),
    (5, 64, 14, 14, 64, True, False, False), (1, 64, 200, 301, 256, True, True, True),
])
def test_pointwise_streaming_kernel_f16(ops, case):
    """seam_conv1x1_swh_f16 (csrc/seam_pwh.hip, round 6: weights stationary in LDS, independent waves, 16-byte NHWC pieces) against
    the fp32 convolution of the same fp16-rounded operands (the bound of the implicit-GEMM tests above), against
    conv_igemm<_Float16> on the same packed layer (two fp32 accumulation orders + one rounding: <= 2 fp16 ulps of the scale),
    repeated launches bit-identical, and an image alone == the same image inside the batch (an output is one wave's fixed chain)."""
    import seam_match_rcnn_amd._native as native
    d = dev()
    n, c, hh, ww, k, relu, bn, use_res = case
    assert native.lib().seam_conv1x1_swh_config(n * hh * ww, c, 0, k) != 0
    x = rnd(700, (n, c, hh, ww)).half().float()
    wt = (rnd(701, (k, c, 1, 1), "w") / math.sqrt(c)).half().float()
    res = rnd(702, (n, k, hh, ww), "r").half().float() if use_res else None
    if bn:
        b = (torch.from_numpy(synth.uniform(synth.stream_id(703, "bw"), (k,), 0.5, 1.5)), rnd(704, (k,), "bb") * 0.1,
             rnd(705, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(706, "rv"), (k,), 0.5, 1.5)))
        pc = ops.pack_conv(wt.to(d), None, bn=tuple(t.to(d) for t in b), dtype=H)
        sc = b[0] * (b[3] + 1e-5).rsqrt()
        ref = F.conv2d(x, wt) * sc[None, :, None, None] + (b[1] - b[2] * sc)[None, :, None, None]
    else:
        bias = rnd(703, (k,), "b") * 0.1
        pc = ops.pack_conv(wt.to(d), bias.to(d), dtype=H)
        ref = F.conv2d(x, wt, bias)
    if use_res:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    assert pc.wsh is not None and hh * ww >= ops.SW_MIN_HW
    xd = nhwc(x).half().to(d)
    rd = nhwc(res).half().to(d) if use_res else None
    saved, ops.CONV_TRACE = ops.SWH, []
    try:
        ops.SWH = True
        got = ops.conv2d(xd, pc, relu, rd)
        assert ops.CONV_TRACE[0][0].startswith("conv1x1_swh"), ops.CONV_TRACE[0][0]
        ops.CONV_TRACE = None
        assert got.dtype == H
        assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)
        for _ in range(4):
            assert torch.equal(ops.conv2d(xd, pc, relu, rd), got)
        alone = ops.conv2d(xd[n - 1:].contiguous(), pc, relu, None if rd is None else rd[n - 1:].contiguous())
        assert torch.equal(alone, got[n - 1:])
        ops.SWH = False
        other = ops.conv2d(xd, pc, relu, rd)
        scale = float(other.float().abs().max())
        assert float((got.float() - other.float()).abs().max()) <= 2.0 ** -9 * scale
    finally:
        ops.SWH, ops.CONV_TRACE = saved, None


def test_pointwise_streaming_kernel_f16_dual_and_unserved(ops):
    """The two-source form (stride-1 projection shortcut: [W_a | W_b] . [h ; x]) on the streaming kernel == the implicit GEMM's dual
    form to rounding; unserved shapes (C not a multiple of 64, C > 512, 64-channel slabs with C > 128, K / slab not dividing 32) are refused by the C ABI and stay
    on the implicit GEMM in ops.conv2d."""
    import seam_match_rcnn_amd._native as native
    d = dev()
    lib = native.lib()
    n, hh, ww, c1, c2, k = 2, 37, 45, 64, 64, 256
    h1, x2 = rnd(720, (n, c1, hh, ww)).half().float(), rnd(721, (n, c2, hh, ww)).half().float()
    wa, wb = rnd(722, (k, c1, 1, 1), "w") / math.sqrt(c1), rnd(723, (k, c2, 1, 1), "w") / math.sqrt(c2)
    bn = lambda s: (torch.from_numpy(synth.uniform(synth.stream_id(s, "bw"), (k,), 0.5, 1.5)), rnd(s + 1, (k,), "bb") * 0.1,      # noqa: E731
                    rnd(s + 2, (k,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(s + 3, "rv"), (k,), 0.5, 1.5)))
    b1, b2 = bn(730), bn(740)
    pcd = ops.pack_conv_dual(wa.to(d), tuple(t.to(d) for t in b1), wb.to(d), tuple(t.to(d) for t in b2), dtype=H)
    assert pcd.wsh is not None
    xa, xb = nhwc(h1).half().to(d), nhwc(x2).half().to(d)
    saved, ops.CONV_TRACE = ops.SWH, []
    try:
        ops.SWH = True
        got = ops.conv2d_dual(xa, xb, pcd, 1, relu=True)
        assert ops.CONV_TRACE[0][0].startswith("conv1x1_swh"), ops.CONV_TRACE[0][0]
        ops.CONV_TRACE = None
        assert torch.equal(ops.conv2d_dual(xa, xb, pcd, 1, relu=True), got)
        ops.SWH = False
        other = ops.conv2d_dual(xa, xb, pcd, 1, relu=True)
    finally:
        ops.SWH, ops.CONV_TRACE = saved, None
    assert float((got.float() - other.float()).abs().max()) <= 2.0 ** -9 * float(other.float().abs().max())
    s1, s2 = b1[0] * (b1[3] + 1e-5).rsqrt(), b2[0] * (b2[3] + 1e-5).rsqrt()
    ref = F.relu(F.conv2d(h1, wa) * s1[None, :, None, None] + F.conv2d(x2, wb) * s2[None, :, None, None]
                 + ((b1[1] - b1[2] * s1) + (b2[1] - b2[2] * s2))[None, :, None, None])
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=4e-3, atol_scale=2e-3)
    dummy = torch.zeros(1 << 16, dtype=H, device=d)
    for (c, kk) in [(96, 256), (1024, 256), (256, 64), (32, 64), (64, 96), (64, 8448 * 2)]:
        assert lib.seam_conv1x1_swh_config(1000, c, 0, kk) == 0
        assert lib.seam_conv1x1_swh_f16(dummy.data_ptr(), None, dummy.data_ptr(), None, None, None, dummy.data_ptr(), 64, c, 0, kk, 0, 0, 0, 0, 0, 0,
                                        torch.cuda.current_stream().cuda_stream) != 0


@pytest.mark.parametrize("shape", [(2, 256, 48, 84, 256), (3, 512, 23, 31, 256), (1, 256, 97, 50, 256)])
def test_fpn_topdown_merge_on_the_streaming_kernel_f16(ops, shape):
    """conv2d_topdown on the fp16 path (round 6): the lateral 1x1 conv with the coarser level added through a nearest-neighbour
    upsample in seam_conv1x1_swh_f16's epilogue (res_mode 2) against the fp32 reference of the same fp16-rounded operands
    [TV FeaturePyramidNetwork.forward: inner + F.interpolate(top, size, "nearest")] and against the two-kernel form it replaces
    (implicit GEMM + seam_upsample_add_f16: one more fp16 rounding)."""
    d = dev()
    n, c, hh, ww, k = shape
    x = rnd(760, (n, c, hh, ww)).half().float()
    wt = (rnd(761, (k, c, 1, 1), "w") / math.sqrt(c)).half().float()
    bias = rnd(762, (k,), "b") * 0.1
    top = rnd(763, (n, k, (hh + 1) // 2, (ww + 1) // 2)).half().float()
    ref = F.conv2d(x, wt, bias) + F.interpolate(top, size=(hh, ww), mode="nearest")
    pc = ops.pack_conv(wt.to(d), bias.to(d), dtype=H)
    xd, td = nhwc(x).half().to(d), nhwc(top).half().to(d)
    saved, ops.CONV_TRACE = ops.SWH, []
    try:
        ops.SWH = True
        got = ops.conv2d_topdown(xd, pc, td)
        assert ops.CONV_TRACE[0][0].startswith("conv1x1_swh"), ops.CONV_TRACE[0][0]
        ops.CONV_TRACE = None
        assert torch.equal(ops.conv2d_topdown(xd, pc, td), got)
        ops.SWH = False
        two = ops.conv2d_topdown(xd, pc, td)
    finally:
        ops.SWH, ops.CONV_TRACE = saved, None
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)
    assert float((got.float() - two.float()).abs().max()) <= 2.0 ** -8 * float(two.float().abs().max())


@pytest.mark.parametrize("shape", [(2, 21, 35), (3, 64, 50), (1, 97, 131), (2, 8, 9)])
def test_stem_on_the_streaming_kernel_f16(ops, shape):
    """seam_stem_s2d_swh_f16 (round 6): ResNet.conv1 + bn1 + relu on the zero-padded space-to-depth frame -- a tap is a constant
    shift of the flattened cell index -- against the fp32 reference of the same fp16-rounded operands (the 4x4 / pad-2 convolution
    of seam_conv2d_crop_f16's call site) and against that kernel (<= 2 fp16 ulps); repeat launches identical; an image alone == the
    same image inside the batch; ragged sizes (tiles that straddle rows, images and the end of the frame)."""
    d = dev()
    n, h2, w2 = shape
    x = rnd(780, (n, 16, h2, w2)).half().float()
    x[:, 12:] = 0                                      # the four zero channels of the fp16 cells
    ws = (rnd(781, (64, 12, 4, 4), "w") / math.sqrt(12 * 16)).half().float()
    bn = (torch.from_numpy(synth.uniform(synth.stream_id(782, "bw"), (64,), 0.5, 1.5)), rnd(783, (64,), "bb") * 0.1,
          rnd(784, (64,), "rm") * 0.1, torch.from_numpy(synth.uniform(synth.stream_id(785, "rv"), (64,), 0.5, 1.5)))
    pc = ops.pack_conv(ws.to(d), None, tuple(t.to(d) for t in bn), stride=1, pad=2, cstore=16, wino=False, dtype=H)
    rows = F.pad(ws, (0, 0, 0, 0, 0, 4)).permute(0, 2, 3, 1).reshape(64, 256).half().contiguous().to(d)
    xd = nhwc(x).half().to(d)
    got = ops.stem_s2d_f16(xd, rows, pc.scale, pc.shift, relu=True)
    assert got.dtype == H and tuple(got.shape) == (n, h2, w2, 64)
    assert torch.equal(ops.stem_s2d_f16(xd, rows, pc.scale, pc.shift, relu=True), got)
    assert torch.equal(ops.stem_s2d_f16(xd[n - 1:].contiguous(), rows, pc.scale, pc.shift, relu=True), got[n - 1:])
    sc = bn[0] * (bn[3] + 1e-5).rsqrt()
    full = F.conv2d(F.pad(x[:, :12], (2, 1, 2, 1)), ws)
    ref = F.relu(full * sc[None, :, None, None] + (bn[1] - bn[2] * sc)[None, :, None, None])
    assert_close(got.float().permute(0, 3, 1, 2), ref, rtol=2e-3, atol_scale=1e-3)
    old = ops.conv2d(xd, pc, relu=True, out_hw=(h2, w2))
    assert float((got.float() - old.float()).abs().max()) <= 2.0 ** -9 * float(old.float().abs().max())


def test_padded_space_to_depth_frame_f16(ops):
    """preprocess(..., s2d=True, s2d_pad=(2, 1)) (seam_preprocess_s2d_pad_batch_f16): the interior is the unpadded frame bit for
    bit, the border cells are zero -- single-launch batch form (frames of one clip tensor) and the per-image form (ragged sizes)."""
    d = dev()
    clip = torch.from_numpy(synth.uniform(synth.stream_id(790, "img"), (3, 3, 60, 90))).to(d)
    ref_sizes = [(64, 96)] * 3
    for imgs, sizes in (([clip[i] for i in range(3)], ref_sizes),
                        ([clip[0], clip[1, :, :50, :80].contiguous()], [(64, 96), (53, 85)])):
        plain = ops.preprocess(imgs, sizes, 64, 96, H, s2d=True)
        padded = ops.preprocess(imgs, sizes, 64, 96, H, s2d=True, s2d_pad=(2, 1))
        assert tuple(padded.shape) == (len(imgs), 32 + 3, 48 + 3, 16)
        assert torch.equal(padded[:, 2:-1, 2:-1], plain)
        border = padded.clone()
        border[:, 2:-1, 2:-1] = 0
        assert float(border.float().abs().max()) == 0.0
