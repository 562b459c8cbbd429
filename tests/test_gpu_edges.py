"""Edge cases of the drop-in heads (empty / degenerate inputs), checked against the oracle's behaviour, which follows
the reference's (ref models/match_head.py:66-76,90-169)."""
import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import heads as OH
from test_gpu_ops import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def heads():
    from seam_match_rcnn_amd.models.match_head import MatchPredictor, TemporalAggregationNLB
    mp, ta = MatchPredictor(), TemporalAggregationNLB()
    mp.load_state_dict(to_torch(synth.match_predictor_state(11)))
    ta.load_state_dict(to_torch(synth.temporal_aggregator_state(12)))
    return mp.to(DEV).eval(), ta.to(DEV).eval()


def test_match_predictor_one_sided_types(heads):
    mp, _ = heads
    p = to_torch(synth.match_predictor_state(11))
    x = torch.from_numpy(synth.roi_features(60, 3))
    with torch.no_grad():
        for types in ([0, 0, 0], [1, 1, 1], [0, 1, 1]):
            t = torch.IntTensor(types)
            x3, x5 = mp(x.to(DEV), t)
            ox3, ox5 = OH.match_predictor_forward(x, t, p)
            assert tuple(x5.shape) == tuple(ox5.shape) == (types.count(0), types.count(1), 2)
            assert_close(x3, ox3)
            if x5.numel():
                assert_close(x5, ox5)


def test_aggregator_mode_a_degenerate(heads):
    _, ta = heads
    p = to_torch(synth.temporal_aggregator_state(12))
    x = torch.from_numpy(synth.roi_features(61, 4))
    with torch.no_grad():
        # no street rows at all: the reference returns (None, x3_2, None, None, None, empty ids)   (:123-125,163-164)
        out = ta(x.to(DEV), torch.IntTensor([1, 1, 1, 1]), torch.LongTensor([0, 1, 2, 3]))
        ref = OH.temporal_aggregation_forward(x, torch.IntTensor([1, 1, 1, 1]), torch.LongTensor([0, 1, 2, 3]), p)
        assert out[0] is None and out[2] is None and ref[0] is None and ref[2] is None
        assert_close(out[1], ref[1])
        assert out[5].numel() == 0
        # only street rows, every sequence of length 1 (NLB bypass), no shop rows -> x5 [S,0,2]
        out = ta(x.to(DEV), torch.IntTensor([0, 0, 0, 0]), torch.LongTensor([5, 3, 9, 1]))
        ref = OH.temporal_aggregation_forward(x, torch.IntTensor([0, 0, 0, 0]), torch.LongTensor([5, 3, 9, 1]), p)
        assert tuple(out[2].shape) == tuple(ref[2].shape) == (4, 0, 2)
        assert_close(out[0], ref[0])
        assert torch.equal(out[4].cpu(), ref[4]) and tuple(out[3].shape) == tuple(ref[3].shape) == (2, 4, 256)
        # a single ROI, a single sequence
        out = ta(x[:1].to(DEV), torch.IntTensor([0]), torch.LongTensor([7]), getatt=True)
        ref = OH.temporal_aggregation_forward(x[:1], torch.IntTensor([0]), torch.LongTensor([7]), p, getatt=True)
        assert_close(out[0], ref[0])
        assert_close(out[6][0], ref[6][0])


def test_aggregator_mode_b_degenerate(heads):
    _, ta = heads
    p = to_torch(synth.temporal_aggregator_state(12))
    # one sequence of one frame vs a one-product bank; and a fully unmasked sequence
    seq = torch.zeros((3, 2, 256))
    seq[1:, :] = torch.from_numpy(synth.normal(synth.stream_id(62, "s"), (2, 2, 256)))
    mask = torch.zeros((2, 3), dtype=torch.bool)
    mask[0, 2:] = True                      # sequence 0 has length 1, sequence 1 length 2 (no masked entry)
    bank = torch.from_numpy(synth.gallery(63, 1))
    with torch.no_grad():
        out = ta(None, None, None, x3_1_seq=seq.to(DEV), x3_1_mask=mask.to(DEV), x3_2=bank.to(DEV))
    ref = OH.temporal_aggregation_forward(None, None, None, p, x3_1_seq=seq, x3_1_mask=mask, x3_2=bank)
    assert tuple(out[2].shape) == (2, 1, 2)
    assert_close(out[0], ref[0])
    assert_close(out[2], ref[2])


def test_kernel_entry_points_with_empty_or_tiny_problems():
    from seam_match_rcnn_amd import ops
    a = torch.from_numpy(synth.gallery(64, 3)).to(DEV)
    w = torch.from_numpy(synth.match_predictor_state(11)["last.weight"]).to(DEV)
    b = torch.from_numpy(synth.match_predictor_state(11)["last.bias"]).to(DEV)
    assert tuple(ops.pair_logits(a[:0], a, w, b).shape) == (0, 3, 2)
    assert tuple(ops.pair_logits(a, a[:0], w, b).shape) == (3, 0, 2)
    x5 = ops.pair_logits(a, a[:1], w, b)                     # G = 1
    idx, sc = ops.rank_topk(x5, 20)                          # k clamped to G
    assert tuple(idx.shape) == (3, 1) and int(idx.max()) == 0
    i2, s2 = ops.pair_topk(a, a, w, b, 20)
    oi, _ = OH.rank_topk(OH.pair_logits(a.cpu(), a.cpu(), w.cpu(), b.cpu()), 3)
    assert torch.equal(i2.cpu(), oi)
    # zero ROIs through RoIAlign and the trunk
    feats = [torch.zeros((1, s, s, 256), device=DEV) for s in (50, 25, 13, 7)]
    out = ops.roi_align(feats, torch.zeros((0, 5), device=DEV), [0.25, 0.125, 0.0625, 0.03125], 14)
    assert tuple(out.shape) == (0, 14, 14, 256)
    from seam_match_rcnn_amd.models.match_head import MatchPredictor
    mp = MatchPredictor().to(DEV).eval()
    with torch.no_grad():
        assert tuple(mp.trunk_nhwc(out).shape) == (0, 256)


def test_single_image_model_forward_has_no_shop_side():
    """model([one image]): every ROI is of type 0 (ref models/video_matchrcnn.py:299-307), the discarded x5 is [n0,0,2]."""
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(to_torch(synth.video_matchrcnn_state(5)))
    m = m.to(DEV).eval()
    m.transform.min_size, m.transform.max_size = 128, 160
    with torch.no_grad():
        out = m([torch.from_numpy(synth.frames(70, 1, 128, 160)[0]).to(DEV)])
    assert len(out) == 1 and out[0]["match_features"].shape[0] == out[0]["boxes"].shape[0] == out[0]["roi_features"].shape[0] >= 1
