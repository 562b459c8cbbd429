"""Split-bf16 ("bx3") convolution path: fp32 activations, 3 bf16 MFMAs per product, fp32 accumulate -- parity with the
fp32 oracle at a tolerance 30x tighter than the 1e-3 contract."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import seam_match_rcnn_amd.synth as synth
from test_gpu_ops import CONV_CASES, nhwc, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, ref):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max() / ref.abs().max()), float((got - ref).norm() / ref.norm())


@pytest.mark.parametrize("case", CONV_CASES + [(4, 256, 50, 50, 256, 3, 1, 1, False, False, True), (1, 512, 9, 9, 2048, 1, 1, 0, True, True, True)])
def test_conv2d_bx3(case):
    from seam_match_rcnn_amd import ops
    n, c, h, w, k, r, stride, pad, bn, res, relu = case
    x = rnd(1, (n, c, h, w))
    wt = rnd(2, (k, c, r, r), "w") * (1.0 / math.sqrt(c * r * r))
    bias = rnd(3, (k,), "b") * 0.1
    ref = F.conv2d(x.double(), wt.double(), bias.double(), stride, pad)
    resid = None
    if res:
        resid = rnd(8, ref.shape, "res")
        ref = ref + resid.double()
    if relu:
        ref = F.relu(ref)
    pc = ops.pack_conv(wt.to(DEV), bias.to(DEV), stride=stride, pad=pad, dtype=ops.BX3)
    pc32 = ops.pack_conv(wt.to(DEV), bias.to(DEV), stride=stride, pad=pad)
    xin = nhwc(x)
    if c % 4:
        xin = F.pad(xin, (0, 4 - c % 4))
    rd = None if resid is None else nhwc(resid).to(DEV)
    y = ops.conv2d(xin.to(DEV), pc, relu, rd).permute(0, 3, 1, 2)
    y32 = ops.conv2d(xin.to(DEV), pc32, relu, rd).permute(0, 3, 1, 2)
    emax, el2 = rel_err(y, ref)
    emax32, _ = rel_err(y32, ref)
    assert emax < 3e-5 and el2 < 1e-5, (emax, el2, emax32)


def test_split_is_exactly_three_terms():
    """one product, hand-checkable: a = 1 + 2^-9 + 2^-17 -> hi = 1 + 2^-8?? no: rn_bf16(a); the kernel must return
    hi*hi + hi*lo + lo*hi in fp32 for a 1x1 conv with a single non-zero weight."""
    from seam_match_rcnn_amd import ops
    a = np.float32(1.0 + 2.0 ** -9 + 2.0 ** -15)
    b = np.float32(3.0 + 2.0 ** -7 + 2.0 ** -14)
    def split(v):
        t = torch.tensor([v], dtype=torch.float32)
        hi = t.to(torch.bfloat16).to(torch.float32)
        lo = (t - hi).to(torch.bfloat16).to(torch.float32)
        return float(hi), float(lo)
    ah, al = split(a)
    bh, bl = split(b)
    want = np.float32(np.float32(np.float32(ah * bl) + np.float32(al * bh)) + np.float32(ah * bh))
    x = torch.zeros(1, 1, 1, 32)
    x[..., 5] = float(a)
    wt = torch.zeros(64, 32, 1, 1)
    wt[7, 5] = float(b)
    y = ops.conv2d(x.to(DEV), ops.pack_conv(wt.to(DEV), dtype=ops.BX3))
    got = float(y[0, 0, 0, 7])
    assert abs(got - float(want)) <= 2 ** -22 * abs(float(want)), (got, float(want), float(a) * float(b))
    assert float(y.abs().sum()) == abs(got)


def test_fixed_roi_forward_bx3_vs_fp32_oracle():
    """Whole extractor + trunks + SEAM heads with split-bf16 contractions vs the fp32 CPU oracle: every output of
    north_star's parity list (ROI features, descriptors, match logits, top-k) far inside 1e-3."""
    from conftest import to_torch
    from oracle import detection as OD, heads as OH, model as OM
    from seam_match_rcnn_amd import ops
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    m = m.to(DEV).eval().set_compute_dtype(ops.BX3)
    m.transform.min_size, m.transform.max_size = 256, 320
    imgs = [torch.from_numpy(synth.frames(7 + i, 1, 256, 320)[0]) for i in range(3)]
    rois = [torch.from_numpy(synth.fixed_rois(8, 256, 320))] * 3
    with torch.no_grad():
        res, feats, _ = m.forward_fixed_rois([i.to(DEV) for i in imgs], rois)
    assert feats["0"].dtype == torch.float32
    batch, sizes = OD.transform(imgs, 256, 320)
    ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
    orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], rois, sizes, 14)
    ox3 = OH.match_trunk(orf, OM.sub(sd, "roi_heads.match_predictor."))

    def rel(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).abs().max() / b.abs().max()), float((a - b).norm() / b.norm())

    for k in "0123":
        emax, el2 = rel(feats[k].permute(0, 3, 1, 2), ofe[k])
        assert emax < 2e-4 and el2 < 3e-5, (k, emax, el2)
    emax, el2 = rel(torch.cat([r["roi_features"] for r in res]), orf)
    assert emax < 2e-4 and el2 < 3e-5, ("roi_features", emax, el2)
    emax, el2 = rel(torch.cat([r["match_features"] for r in res]), ox3)
    assert emax < 2e-4 and el2 < 5e-5, ("match_features", emax, el2)
    # SEAM head: 8 sequences of 3 frames vs a 40-product bank -> logits and exact top-5
    ta = m.roi_heads.temporal_aggregator
    x = torch.cat([r["roi_features"] for r in res])
    types = torch.zeros(24, dtype=torch.int32)
    ids = torch.arange(8).repeat(3)
    bank = torch.from_numpy(synth.gallery(36, 40))
    with torch.no_grad():
        out = ta(x, types, ids)
        x5 = ta.pair(out[0], bank.to(DEV))
        idx, _ = ops.rank_topk(x5, 5)
    tap = OM.sub(sd, "roi_heads.temporal_aggregator.")
    oo = OH.temporal_aggregation_forward(orf, types, ids, tap)
    ox5 = OH.pair_logits(oo[0], bank, tap["last.weight"], tap["last.bias"])
    emax, el2 = rel(x5, ox5)
    assert emax < 2e-4 and el2 < 5e-5, ("x5", emax, el2)
    oidx, _ = OH.rank_topk(ox5, 5)
    assert torch.equal(idx.cpu(), oidx)
