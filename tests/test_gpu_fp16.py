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
