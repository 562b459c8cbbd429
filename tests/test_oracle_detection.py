"""CPU: analytic known-answer tests pinning the torchvision restatement (oracle/detection.py).
torchvision itself is absent here and unpinned upstream (PARITY UNPINNED vs the real library);
these fix the published semantics of SURVEY.md Appendix A."""
import math

import torch

from oracle import detection as OD


def test_roi_align_constant_and_linear_field():
    # constant feature -> constant output; linear ramp -> exact bilinear value at the sample centres
    h = w = 16
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    feat = torch.stack([torch.full((h, w), 3.5), xx, yy, 2 * xx + 3 * yy])[None]
    roi = torch.tensor([[0., 2., 3., 10., 9.]])          # interior ROI, scale 1
    out = OD.roi_align(feat, roi, 1.0, 4, 2)[0]
    assert torch.allclose(out[0], torch.full((4, 4), 3.5))
    bw, bh = 8 / 4, 6 / 4
    cx = torch.tensor([2 + (p + 0.5) * bw for p in range(4)])      # mean of the 2 sample points = bin centre
    cy = torch.tensor([3 + (p + 0.5) * bh for p in range(4)])
    assert torch.allclose(out[1], cx[None, :].expand(4, 4), atol=1e-5)
    assert torch.allclose(out[2], cy[:, None].expand(4, 4), atol=1e-5)
    assert torch.allclose(out[3], 2 * cx[None, :] + 3 * cy[:, None], atol=1e-4)


def test_roi_align_vectorised_equals_scalar_restatement_incl_edges():
    torch.manual_seed(0)
    feat = torch.randn(2, 5, 13, 11)
    rois = torch.tensor([[0., -3., -2., 20., 30.],      # overhangs every border
                         [1., 1.2, 3.4, 1.3, 3.5],       # degenerate: width/height clamped to 1
                         [1., 0., 0., 44., 52.],         # whole map at scale 1/4
                         [0., 39., 47., 44., 52.]])      # bottom-right corner samples (y >= H-1 rule)
    for pooled in (2, 7):
        vec = OD.roi_align(feat, rois, 0.25, pooled, 2)
        for i in range(rois.shape[0]):
            assert torch.allclose(vec[i], OD.roi_align_scalar(feat, rois[i], 0.25, pooled, 2), atol=1e-5)


def test_level_mapper_boundaries():
    def box(side):
        return torch.tensor([[0., 0., side, side]])
    got = [int(OD.map_levels(box(s))[0]) for s in (10, 111.9, 112, 223.9, 224, 447.9, 448, 2000)]
    assert got == [0, 0, 1, 1, 2, 2, 3, 3]
    assert OD.infer_scales([(200, 200), (100, 100), (50, 50), (25, 25)], [(800, 800)]) == [0.25, 0.125, 0.0625, 0.03125]
    assert OD.infer_scales([(192, 336), (96, 168), (48, 84), (24, 42)], [(749, 1333)]) == [0.25, 0.125, 0.0625, 0.03125]


def test_nms_known_case_and_batched_offsets():
    boxes = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.], [0., 0., 10., 5.],
                          [20., 20., 30., 30.]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6, 0.7])
    # IoU(0,1) = 81/119 = 0.68 > 0.5 -> 1 suppressed; IoU(0,3) = 0.5 (not >) -> kept; equal-score twins: lower index wins
    assert OD.nms(boxes, scores, 0.5).tolist() == [0, 2, 3]
    assert OD.nms(boxes, scores, 0.7).tolist() == [0, 1, 2, 3]
    # different classes never suppress each other
    assert OD.batched_nms(boxes, scores, torch.tensor([0, 1, 0, 0, 1]), 0.5).tolist() == [0, 1, 2, 4, 3]


def test_decode_identity_clamp_and_anchor_quirks():
    a = torch.tensor([[10., 20., 50., 100.]])
    assert torch.allclose(OD.decode_boxes(torch.zeros(1, 4), a), a)
    big = OD.decode_boxes(torch.tensor([[0., 0., 100., 100.]]), a)          # dw, dh clamped to log(1000/16)
    assert math.isclose(float(big[0, 2] - big[0, 0]), 40 * 1000 / 16, rel_tol=1e-5)
    d = OD.decode_boxes(torch.tensor([[10., 0., 0., 0.]]), a, (10., 10., 5., 5.))
    assert torch.allclose(d, a + torch.tensor([[40., 0., 40., 0.]]))
    anc = OD.grid_anchors((800, 800), [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)])
    assert [x.shape[0] for x in anc] == [120000, 30000, 7500, 1875, 507]
    assert anc[4][3 * 13].tolist()[1] == 61.0 - 181.0                       # row 1 of the 'pool' level: y shift 61


def test_frozen_bn_and_transform_padding():
    p = {"bn.weight": torch.tensor([2.0]), "bn.bias": torch.tensor([0.5]), "bn.running_mean": torch.tensor([1.0]),
         "bn.running_var": torch.tensor([4.0 - 1e-5])}
    y = OD.frozen_bn(torch.tensor([[[[3.0]]]]), p, "bn")
    assert math.isclose(float(y), (3.0 - 1.0) / 2.0 * 2.0 + 0.5, rel_tol=1e-6)
    imgs = [torch.rand(3, 50, 70), torch.rand(3, 40, 90)]
    batch, sizes = OD.transform(imgs, min_size=64, max_size=100)
    assert batch.shape[-2] % 32 == 0 and batch.shape[-1] % 32 == 0
    for i, (h, w) in enumerate(sizes):
        assert float(batch[i, :, h:, :].abs().sum()) == 0 and float(batch[i, :, :, w:].abs().sum()) == 0
    # identity scale: pure normalisation
    one, sz = OD.transform([imgs[0]], min_size=50, max_size=70)
    ref = (imgs[0] - torch.tensor(OD.IMAGE_MEAN)[:, None, None]) / torch.tensor(OD.IMAGE_STD)[:, None, None]
    assert torch.allclose(one[0, :, :50, :70], ref, atol=1e-6)


def test_mask_inference_and_paste():
    logits = torch.zeros(2, 3, 28, 28)
    logits[0, 1] = 10.0
    logits[1, 2] = -10.0
    pr = OD.maskrcnn_inference(logits, [torch.tensor([1]), torch.tensor([2])])
    assert float(pr[0].min()) > 0.9999 and float(pr[1].max()) < 1e-4
    pasted = OD.paste_masks_in_image(torch.ones(1, 1, 28, 28), torch.tensor([[10., 12., 29., 40.]]), (64, 48))
    assert pasted.shape == (1, 1, 64, 48)
    assert float(pasted[0, 0, 20:35, 14:26].min()) > 0.99 and float(pasted[0, 0, :8].max()) == 0.0
