"""BASELINE configs[0] at its FULL size -- one 800x800 synthetic frame, the 8 fixed ROIs of SURVEY 8d, a 16-product
gallery -- against the CPU oracle, plus the size-independent properties the domain offers at the config-2 size:
batch invariance (a frame's outputs do not depend on what else is in the batch) and run-to-run determinism, bit for bit."""
import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import detection as OD
from oracle import heads as OH
from oracle import model as OM
from test_gpu_ops import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def model_and_state():
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def test_config1_full_size_vs_oracle(model_and_state):
    m, sd = model_and_state
    img = torch.from_numpy(synth.frames(0, 1, 800, 800)[0])
    rois = torch.from_numpy(synth.fixed_rois(8, 800, 800))            # incl. the full-image fallback box
    bank = torch.from_numpy(synth.gallery(7, 16))
    with torch.no_grad():
        res, feats, _ = m.forward_fixed_rois([img.to(DEV)], [rois])
        ta = m.roi_heads.temporal_aggregator
        x = res[0]["roi_features"]
        out = ta(x, torch.zeros(8, dtype=torch.int32), torch.arange(8))        # 8 sequences of one frame
        x5 = ta.pair(out[0], bank.to(DEV))
        from seam_match_rcnn_amd import ops
        idx, score = ops.rank_topk(x5, 5)
    batch, sizes = OD.transform([img], 800, 1333)
    assert tuple(batch.shape[-2:]) == (800, 800)
    ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
    for k in ofe:
        assert_close(feats[k].permute(0, 3, 1, 2), ofe[k])
    orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], [rois], sizes, 14)
    assert_close(x, orf)
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    assert_close(res[0]["match_features"], OH.match_trunk(orf, mp))
    tap = OM.sub(sd, "roi_heads.temporal_aggregator.")
    oo = OH.temporal_aggregation_forward(orf, torch.zeros(8, dtype=torch.int32), torch.arange(8), tap)
    ox5 = OH.pair_logits(oo[0], bank, tap["last.weight"], tap["last.bias"])
    assert_close(x5, ox5)
    oidx, oscore = OH.rank_topk(ox5, 5)
    assert torch.equal(idx.cpu(), oidx)
    assert_close(score, oscore)
    probs = OD.maskrcnn_inference(OD.mask_head(orf, sd), [torch.ones(8, dtype=torch.int64)])[0]
    assert_close(res[0]["masks"], probs)


def test_batch_invariance_and_determinism_at_config2_size(model_and_state):
    """fp32 accumulation order is fixed per output element (k loop), independent of tiling and batch: a frame's FPN maps,
    ROI features and descriptors are BIT-identical alone, inside a batch of 5, and from run to run."""
    m, _ = model_and_state
    frames = [torch.from_numpy(f).to(DEV) for f in synth.frames(3, 5, 800, 800)]
    rois = [torch.from_numpy(synth.fixed_rois(32, 800, 800))] * 5
    with torch.no_grad():
        res_b, feats_b, _ = m.forward_fixed_rois(frames, rois)
        res_b2, feats_b2, _ = m.forward_fixed_rois(frames, rois)
        res_1, feats_1, _ = m.forward_fixed_rois(frames[2:3], rois[:1])
    for k in feats_b:
        assert torch.equal(feats_b[k], feats_b2[k])
        assert torch.equal(feats_b[k][2], feats_1[k][0]), k
    for key in ("roi_features", "match_features", "masks"):
        assert torch.equal(res_b[2][key], res_b2[2][key])
        assert torch.equal(res_b[2][key], res_1[0][key]), key
