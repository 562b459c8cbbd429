"""Model-level parity of the DISCRETE stages (RPN proposal selection, box-head post-processing, the image model of row a15)
against the oracle, compared as EXACT sets: every oracle box must have a partner on the device with the same label, coordinates
within 0.05 px and the same score, and vice versa.  Top-k / NMS decisions on values that differ by 1e-6 between two fp32
implementations can flip for a handful of near-ties: those are counted, printed and bounded (tests/parity_sets.py); the
full-size (800x800) variant is tests/test_gpu_forward_fullsize.py."""
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import detection as OD
from oracle import heads as OH
from oracle import model as OM
from parity_sets import assert_same_set
from test_gpu_ops import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def iou_matrix(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter).clamp_min(1e-12)


def match_sets(a_boxes, b_boxes, a_labels=None, b_labels=None, thr=0.99):
    """-> (fraction of a matched in b, fraction of b matched in a, partner index in b of each a entry or -1)."""
    if len(a_boxes) == 0 or len(b_boxes) == 0:
        return float(len(a_boxes) == 0), float(len(b_boxes) == 0), torch.full((len(a_boxes),), -1, dtype=torch.int64)
    iou = iou_matrix(a_boxes, b_boxes)
    if a_labels is not None:
        iou = torch.where(a_labels.cpu()[:, None] == b_labels.cpu()[None], iou, torch.zeros(()))
    best, arg = iou.max(1)
    partner = torch.where(best >= thr, arg, torch.full_like(arg, -1))
    return float((best >= thr).float().mean()), float((iou.max(0)[0] >= thr).float().mean()), partner


@pytest.fixture(scope="module")
def model_and_state():
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def test_rpn_proposals_match_oracle(model_and_state):
    """filter_proposals (per-level top-k, decode, clip, small-box filter, per-level NMS, post-NMS top-n) as a set."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 256, 320
    imgs = [torch.from_numpy(synth.frames(120 + i, 1, 256, 320)[0]) for i in range(2)]
    with torch.no_grad():
        feats, sizes, orig, padded = m.extract_features([i.to(DEV) for i in imgs])
        props = m.rpn(feats, sizes, padded)
    ofe, osz, opad = OM.extract_features(imgs, sd, 256, 320)
    oprops, _, _ = OM.rpn_proposals(ofe, osz, opad, sd)
    for i, (p, o) in enumerate(zip(props, oprops)):
        partner = assert_same_set(o, p, tol_px=1e-2, max_flips=2, what=f"RPN proposals 256x320 image {i}")
        # proposals come out in descending objectness order: position by position, flips aside
        ok = partner >= 0
        assert int((partner[ok] == torch.arange(len(o))[ok]).sum()) >= int(ok.sum()) - 10


def test_full_forward_detections_match_oracle(model_and_state):
    """boxes / labels / scores of the whole drop-in forward (RPN -> box head -> per-class NMS -> top-100) as matched sets."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 256, 320
    imgs = [torch.from_numpy(synth.frames(130 + i, 1, 256, 320)[0]) for i in range(2)]
    with torch.no_grad():
        out = m([i.to(DEV) for i in imgs])
    ofe, osz, opad = OM.extract_features(imgs, sd, 256, 320)
    oprops, _, _ = OM.rpn_proposals(ofe, osz, opad, sd)
    ref = OM.detect(ofe, oprops, osz, sd, 0.1)
    for i, (o, r) in enumerate(zip(out, ref)):
        partner = assert_same_set(r["boxes"], o["boxes"], r["labels"], o["labels"], r["scores"], o["scores"], max_flips=2,
                                  what=f"detections 256x320 image {i}")
        ok = partner >= 0
        assert_close(o["scores"].cpu()[partner[ok]], r["scores"][ok], rtol=1e-4)
        assert torch.equal(o["labels"].cpu()[partner[ok]], r["labels"][ok])


def test_image_model_forward_vs_oracle_with_real_detections(model_and_state):
    """Row a15: ``NewRoIHeads.forward`` eval branch (ref models/matchrcnn.py:451-468) on real detections: the image model's
    boxes / labels / scores / match_features / masks against ``oracle.model.video_matchrcnn_forward(video=False)``."""
    from seam_match_rcnn_amd.models.matchrcnn import matchrcnn_resnet50_fpn
    _, sd = model_and_state
    m1 = matchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m1.load_state_dict({k: v for k, v in sd.items() if "temporal_aggregator" not in k})
    m1 = m1.to(DEV).eval()
    m1.transform.min_size, m1.transform.max_size = 192, 256
    imgs = [torch.from_numpy(synth.frames(140 + i, 1, 192, 256)[0]) for i in range(3)]
    with torch.no_grad():
        out = m1([i.to(DEV) for i in imgs])
    # the oracle's forward with the same transform sizes (its extract_features takes them through OM.extract_features)
    ofe, osz, opad = OM.extract_features(imgs, sd, 192, 256)
    oprops, _, _ = OM.rpn_proposals(ofe, osz, opad, sd)
    ref = OM.detect(ofe, oprops, osz, sd, fallback_score=1.0)
    mask_props = [r["boxes"] for r in ref]
    orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], mask_props, osz, 14)       # ref :463 (second RoIAlign, same boxes)
    counts = [len(b) for b in mask_props]
    types = torch.tensor([0] * counts[0] + [1] * sum(counts[1:]), dtype=torch.int32)   # ref :455-461
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    ox3, _ = OH.match_predictor_forward(orf, types, mp)
    oprob = OD.maskrcnn_inference(OD.mask_head(orf, sd), [r["labels"] for r in ref])
    off = 0
    n_checked = 0
    for o, r, c, pr in zip(out, ref, counts, oprob):
        assert set(o) == {"boxes", "labels", "scores", "masks", "match_features", "w", "b"}       # no roi_features (ref :465-468)
        assert torch.equal(o["w"].cpu(), mp["last.weight"]) and torch.equal(o["b"].cpu(), mp["last.bias"])
        assert o["match_features"].shape == (len(o["scores"]), 256) and o["masks"].shape[1:] == (1, 192, 256)
        partner = assert_same_set(r["boxes"], o["boxes"], r["labels"], o["labels"], r["scores"], o["scores"], max_flips=2,
                                  what="image-model detections 192x256")               # identity scale: same pixels
        ok = partner >= 0
        assert_close(o["scores"].cpu()[partner[ok]], r["scores"][ok], rtol=1e-4)
        # descriptors of matched detections: boxes agree to ~1e-4 px, so RoIAlign + trunk agree to the usual tolerance
        assert_close(o["match_features"].cpu()[partner[ok]], ox3[off:off + c][ok], rtol=5e-3, atol_scale=2e-3)
        pasted = OD.paste_masks_in_image(pr, r["boxes"], (192, 256))
        gm, om = o["masks"].cpu()[partner[ok]], pasted[ok]
        assert float(((gm - om).abs() > 5e-2).float().mean()) < 5e-3            # paste resamples at box edges: a few edge pixels differ
        n_checked += int(ok.sum())
        off += c
    assert n_checked >= 30


def test_model_refuses_cpu_images(model_and_state):
    """CPU images must raise SeamNativeError on every preprocessing branch -- including same-storage views of one CPU clip
    tensor, which the one-launch batch branch would otherwise hand to the kernel as raw host pointers."""
    from seam_match_rcnn_amd._native import SeamNativeError
    m, _ = model_and_state
    clip = torch.from_numpy(synth.frames(150, 3, 64, 96))
    with torch.no_grad():
        with pytest.raises(SeamNativeError):
            m(list(clip.unbind(0)))                        # constant-stride CPU views (fp32 s2d batch branch)
        with pytest.raises(SeamNativeError):
            m([clip[0].clone()])                           # a single CPU image
        with pytest.raises(SeamNativeError):
            m([clip[0].to(DEV), clip[1]])                  # mixed
        m.set_compute_dtype(torch.float16)
        try:
            with pytest.raises(SeamNativeError):
                m(list(clip.unbind(0)))                    # NHWC8 batch branch
        finally:
            m.set_compute_dtype(torch.float32)


def test_91_classes_and_out_of_range_prefix():
    """COCO-sized head: 90 foreground classes x up to 1000 proposals = up to 90 000 class candidates per image, far beyond the NMS
    kernel's 16 384-box capacity: the prefix is what keeps the call legal, so a prefix changed to 0 or past the capacity after
    construction is clamped instead of reaching the kernel (which would answer hipErrorInvalidValue)."""
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn, TemporalRoIHeads
    sd = to_torch(synth.video_matchrcnn_state(5, 91))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=91)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.transform.min_size, m.transform.max_size = 256, 320
    imgs = [torch.from_numpy(synth.frames(160 + i, 1, 256, 320)[0]).to(DEV) for i in range(2)]
    with torch.no_grad():
        base = m(imgs)
        outs = []
        for prefix in (0, 50000, 16384, 100):
            m.roi_heads.nms_prefix = prefix
            outs.append(m(imgs))
    for out in outs:
        for o, b in zip(out, base):
            assert torch.equal(o["boxes"], b["boxes"]) and torch.equal(o["labels"], b["labels"]) and torch.equal(o["scores"], b["scores"])
    # vs the oracle's per-class NMS over ALL candidates
    cpu = [i.cpu() for i in imgs]
    ofe, osz, opad = OM.extract_features(cpu, sd, 256, 320)
    oprops, _, _ = OM.rpn_proposals(ofe, osz, opad, sd)
    ref = OM.detect(ofe, oprops, osz, sd, 0.1)
    for i, (o, r) in enumerate(zip(base, ref)):
        assert_same_set(r["boxes"], o["boxes"], r["labels"], o["labels"], r["scores"], o["scores"], max_flips=2,
                        what=f"91-class detections image {i}")
    try:
        TemporalRoIHeads.nms_prefix = 0
        with pytest.raises(ValueError):
            TemporalRoIHeads(num_classes=14)
    finally:
        TemporalRoIHeads.nms_prefix = 4096
