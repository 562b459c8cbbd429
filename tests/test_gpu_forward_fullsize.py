"""Full-size parity of the REAL drop-in forward: one 800x800 frame through ``model([img])`` -- 159 882 anchors, per-level
top-1000, 5 x 1000 candidates into the RPN NMS, 1000 proposals through the box head, 13 x 1000 class candidates through the
prefix NMS, top-100, mask + match branches, paste -- against the CPU oracle (ref models/video_matchrcnn.py:154-205,235-314).

Proposals and detections are compared as EXACT sets (same label, coordinates within 0.05 px, scores within 1e-4) up to a
printed list of at most ONE near-tie flip per side (tests/parity_sets.py: the observed count is 0 on every case, the gate is
observed + 1); descriptors / ROI features / masks of the paired detections within the north_star tolerance (1e-3 of scale).
Frames: two synthetic 800x800 seeds and one 1080x1920 frame (resized to 750x1333, padded 768x1344: 257 796 anchors -- the
geometry of BASELINE configs[4])."""
import math

import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import model as OM
from parity_sets import assert_same_set
from test_gpu_ops import assert_close


def assert_same_order(partner, what, ref_logit, noise, max_moved=16, max_shift=2):
    """Both lists are sorted by objectness.  Order is defined by score, ties are not: the device list may differ from the reference
    list only where two members' REFERENCE logits are closer than the two implementations' own disagreement on a logit (`noise` =
    max |device logit - reference logit| over every anchor of the frame, two exact-fp32 chains through ~50 layers; a pair further
    apart than 2 x noise cannot legitimately swap).  Every inverted pair is checked against that bound and the worst one printed, in
    units of the bound and in ulps of the logit; the absolute caps (<= max_moved positions, none displaced by more than max_shift)
    stay as a backstop only.  `partner[i]` = device position of reference member i (-1: flipped out, judged by assert_same_set)."""
    ok = partner >= 0
    pos = torch.arange(len(partner))
    disp = (partner[ok] - pos[ok]).abs()
    moved = int((disp > 0).sum())
    ref_of = torch.full((int(partner.max()) + 1,), -1, dtype=torch.int64)        # device position -> reference index
    ref_of[partner[ok]] = pos[ok]
    worst, worst_ulp, pairs = 0.0, 0.0, 0
    for j in range(len(ref_of)):
        for k in range(j + 1, min(j + 2 * max_shift + 2, len(ref_of))):
            a, b = int(ref_of[j]), int(ref_of[k])
            if a >= 0 and b >= 0 and a > b:                                      # device ranks a before b, the reference b before a
                gap = abs(float(ref_logit[a]) - float(ref_logit[b]))
                ulp = 2.0 ** (math.floor(math.log2(max(abs(float(ref_logit[a])), 1e-30))) - 23)
                pairs += 1
                worst, worst_ulp = max(worst, gap / (2 * noise)), max(worst_ulp, gap / ulp)
                assert gap <= 2 * noise, (what, "members swapped although their reference logits differ by more than the implementations' "
                                          "disagreement", a, b, float(ref_logit[a]), float(ref_logit[b]), noise)
    print(f"[{what}] order: {moved} of {len(partner)} positions moved, largest displacement {int(disp.max()) if len(disp) else 0}; "
          f"{pairs} inverted pair(s), worst reference-logit gap = {worst:.3f} x the 2*noise bound (noise {noise:.3g}) = {worst_ulp:.1f} ulp of the logit")
    assert moved <= max_moved and (len(disp) == 0 or int(disp.max()) <= max_shift), (what, moved, int(disp.max()))


pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def world():
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    img = torch.from_numpy(synth.frames(300, 1, 800, 800)[0])
    with torch.no_grad():
        ofe, osz, opad = OM.extract_features([img], sd)                       # default transform: 800 / 1333
        oprops, oobj, odlt = OM.rpn_proposals(ofe, osz, opad, sd)
        ref, _, _ = OM.video_matchrcnn_forward([img], sd)
    return dict(m=m, sd=sd, img=img, ofe=ofe, osz=osz, opad=opad, oprops=oprops, oobj=oobj, odlt=odlt, ref=ref[0])


def test_rpn_proposals_800(world):
    m, o = world["m"], world["oprops"][0]
    with torch.no_grad():
        feats, sizes, orig, padded = m.extract_features([world["img"].to(DEV)])
        p = m.rpn(feats, sizes, padded)[0].cpu()
    assert tuple(padded) == (800, 800) and len(o) == 1000 and len(p) == 1000            # post-NMS top-n is full at this size
    partner = assert_same_set(o, p, tol_px=1e-2, what="RPN proposals 800x800")
    # ... and in the same objectness order wherever the reference's own logits separate two members
    _check_proposal_order(m, feats, partner, o, world["ofe"], world["osz"], world["opad"], world["oobj"], world["odlt"], "RPN proposals 800x800")


def _check_proposal_order(m, feats, partner, o, ofe, osz, opad, oobj, odlt, what):
    """The order half of the proposal comparison: device logits vs reference logits (the noise), then every inverted pair."""
    from oracle import detection as D
    with torch.no_grad():
        dev_obj = torch.cat([o_.reshape(-1) for o_, _ in m.rpn.head(list(feats.values()))]).cpu()     # NHWC == the reference's (h, w, anchor) order
    anchors = D.grid_anchors(opad, [f.shape[-2:] for f in ofe.values()])
    rprops, _, ridx = D.rpn_filter_proposals(oobj, odlt, anchors, osz, return_index=True)
    assert torch.equal(rprops[0], o)
    ref_obj = torch.cat([t.permute(0, 2, 3, 1).reshape(-1) for t in oobj])
    assert ref_obj.shape == dev_obj.shape
    noise = float((dev_obj - ref_obj).abs().max())
    assert noise <= 1e-3 * float(ref_obj.abs().max())                                   # the north_star tolerance on the logits themselves
    assert_same_order(partner, what, ref_obj[ridx[0]], noise)


def _check_detections(out, ref, what):
    assert set(out) == {"boxes", "labels", "scores", "masks", "match_features", "w", "b", "roi_features"}
    partner = assert_same_set(ref["boxes"], out["boxes"], ref["labels"], out["labels"], ref["scores"], out["scores"], what=what)
    ok = partner >= 0
    g = partner[ok]
    assert int(ok.sum()) >= len(ref["scores"]) - 1
    assert_close(out["scores"].cpu()[g], ref["scores"][ok], rtol=1e-4)
    assert_close(out["match_features"].cpu()[g], ref["match_features"][ok], rtol=1e-3, atol_scale=1e-3)
    assert_close(out["roi_features"].cpu()[g], ref["roi_features"][ok], rtol=1e-3, atol_scale=1e-3)
    gm, om = out["masks"].cpu()[g], ref["masks"][ok]
    assert gm.shape[1:] == om.shape[1:]
    assert float(((gm - om).abs() > 2e-3).float().mean()) < 1e-4           # paste: a box edge 1e-4 px off can move a border pixel
    return partner


def test_drop_in_forward_800(world):
    m, ref = world["m"], world["ref"]
    with torch.no_grad():
        out = m([world["img"].to(DEV)])[0]
    assert len(ref["scores"]) == 100 and len(out["scores"]) == 100
    partner = _check_detections(out, ref, "detections 800x800 (prefix NMS, exact)")
    # the reference's contract for 'roi_features': a fresh contiguous NCHW tensor (ref :314) -- reference-style consumers work
    rf = out["roi_features"]
    assert rf.is_contiguous() and rf.view(rf.shape[0], -1).shape == (100, 256 * 14 * 14)
    assert torch.equal(out["w"].cpu(), world["sd"]["roi_heads.match_predictor.last.weight"])


def test_prefix_nms_paths_agree_800(world):
    """The detection NMS looks at the `nms_prefix` best candidates first and repeats on everything when that prefix cannot be
    proven exact.  Force both outcomes at full size (13 000 candidates): prefix 64 cannot hold 100 survivors -> the full
    13 000-candidate NMS runs; prefix 8192 is exact.  Both must give the default's detections bit for bit."""
    m = world["m"]
    heads = m.roi_heads
    img = world["img"].to(DEV)
    saved = heads.nms_prefix
    try:
        with torch.no_grad():
            base = m([img])[0]
            outs = {}
            for prefix in (64, 8192):
                heads.nms_prefix = prefix
                outs[prefix] = m([img])[0]
    finally:
        heads.nms_prefix = saved
    for prefix, o in outs.items():
        for k in ("boxes", "labels", "scores", "match_features"):
            assert torch.equal(o[k], base[k]), (prefix, k)
    _check_detections(outs[64], world["ref"], "detections 800x800 (prefix too small -> full NMS)")


def test_two_frame_batch_800(world):
    """Two different 800x800 frames in ONE call: the batched, padded post-process (13 000 candidates per image, shared prefix NMS
    launch) must give each frame the detections it gets alone, and the second frame's must match the oracle as an exact set too."""
    m, sd = world["m"], world["sd"]
    img2 = torch.from_numpy(synth.frames(301, 1, 800, 800)[0])
    with torch.no_grad():
        both = m([world["img"].to(DEV), img2.to(DEV)])
        alone = m([img2.to(DEV)])[0]
        ofe, osz, opad = OM.extract_features([img2], sd)
        oprops, _, _ = OM.rpn_proposals(ofe, osz, opad, sd)
        ref2 = OM.detect(ofe, oprops, osz, sd, 0.1)[0]
    for k in ("boxes", "labels", "scores"):
        assert torch.equal(both[1][k], alone[k]), k                       # batch-invariant
    partner = assert_same_set(ref2["boxes"], both[1]["boxes"], ref2["labels"], both[1]["labels"], ref2["scores"], both[1]["scores"],
                              what="detections 800x800, second frame of a batch")
    ok = partner >= 0
    assert_close(both[1]["scores"].cpu()[partner[ok]], ref2["scores"][ok], rtol=1e-4)
    _check_detections(both[0], world["ref"], "detections 800x800, first frame of a batch")


def test_box_branch_on_oracle_proposals_800(world):
    """Stage isolation: the device box branch fed the ORACLE's proposals -- no RPN flip can leak into the comparison."""
    m = world["m"]
    from oracle import model as OMm
    with torch.no_grad():
        feats, sizes, orig, padded = m.extract_features([world["img"].to(DEV)])
        res = m.roi_heads.detect(feats, [world["oprops"][0].to(DEV)], sizes)[0]
        ref = OMm.detect(world["ofe"], world["oprops"], world["osz"], world["sd"], 0.1)[0]
    partner = assert_same_set(ref["boxes"], res["boxes"], ref["labels"], res["labels"], ref["scores"], res["scores"], tol_px=1e-2,
                              what="box branch on oracle proposals")
    ok = partner >= 0
    assert_close(res["scores"].cpu()[partner[ok]], ref["scores"][ok], rtol=1e-4)


@pytest.mark.parametrize("seed,h,w,padded_hw", [(302, 800, 800, (800, 800)), (303, 1080, 1920, (768, 1344))])
def test_drop_in_forward_other_frames(world, seed, h, w, padded_hw):
    """A second 800x800 seed and a 1080x1920 frame through the real ``model([img])``: proposals (exact set, same order) and
    detections / descriptors / ROI features / pasted masks against the oracle."""
    m, sd = world["m"], world["sd"]
    img = torch.from_numpy(synth.frames(seed, 1, h, w)[0])
    with torch.no_grad():
        ofe, osz, opad = OM.extract_features([img], sd)
        oprops, oobj, odlt = OM.rpn_proposals(ofe, osz, opad, sd)
        ref, _, _ = OM.video_matchrcnn_forward([img], sd)
        feats, sizes, orig, padded = m.extract_features([img.to(DEV)])
        p = m.rpn(feats, sizes, padded)[0].cpu()
        out = m([img.to(DEV)])[0]
    assert tuple(padded) == padded_hw == tuple(opad)
    o = oprops[0]
    assert len(o) == len(p)
    partner = assert_same_set(o, p, tol_px=1e-2, what=f"RPN proposals {h}x{w} seed {seed}")
    _check_proposal_order(m, feats, partner, o, ofe, osz, opad, oobj, odlt, f"RPN proposals {h}x{w} seed {seed}")
    assert len(ref[0]["scores"]) == len(out["scores"])
    _check_detections(out, ref[0], f"detections {h}x{w} seed {seed}")
