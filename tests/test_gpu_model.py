"""GPU parity of the module-level API (reference signatures) against golden vectors and the oracle."""
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


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def heads():
    from seam_match_rcnn_amd.models.match_head import MatchPredictor, TemporalAggregationNLB
    mp = MatchPredictor()
    mp.load_state_dict(to_torch(synth.match_predictor_state(11)))
    ta = TemporalAggregationNLB()
    ta.load_state_dict(to_torch(synth.temporal_aggregator_state(12)))
    return mp.to(dev()).eval(), ta.to(dev()).eval()


def test_match_predictor_golden(heads, golden):
    mp, _ = heads
    x = torch.from_numpy(synth.roi_features(31, 6)).to(dev())
    with torch.no_grad():
        x3, x5 = mp(x, torch.IntTensor(golden["mp_types"]))       # CPU IntTensor, as the reference passes it
    assert_close(x3, torch.from_numpy(golden["mp_x3"]))
    assert_close(x5, torch.from_numpy(golden["mp_x5"]))


def test_temporal_aggregation_mode_a_golden(heads, golden):
    _, ta = heads
    x = torch.from_numpy(synth.roi_features(32, 17)).to(dev())
    with torch.no_grad():
        out = ta(x, torch.IntTensor(golden["ta_types"]), torch.LongTensor(golden["ta_ids"]), getatt=True)
    for nm, v in zip(("x3_1b", "x3_2", "x5", "x3_1_seq"), out[:4]):
        assert_close(v, torch.from_numpy(golden["taA_" + nm]))
    assert np.array_equal(out[4].cpu().numpy(), golden["taA_x3_1_mask"])
    assert np.array_equal(out[5].cpu().numpy(), golden["taA_x3_1_ids"])
    assert len(out[6]) == 3
    for i, a in enumerate(out[6]):
        assert_close(a, torch.from_numpy(golden[f"taA_att{i}"]))


def test_temporal_aggregation_mode_b_golden(heads, golden):
    _, ta = heads
    seq = torch.from_numpy(synth.normal(synth.stream_id(33, "seq"), (11, 4, 256)))
    lens = golden["taB_lens"].tolist()
    mask = torch.zeros((4, 11), dtype=torch.bool)
    seq[0] = 0
    for i, n in enumerate(lens):
        mask[i, n + 1:] = True
        seq[n + 1:, i] = 0
    gal = torch.from_numpy(synth.gallery(34, 16))
    # eval mode WITHOUT no_grad, exactly like evaluate_movingfashion.py:258
    out = ta(None, None, None, x3_1_seq=seq.to(dev()), x3_1_mask=mask.to(dev()), x3_2=gal.to(dev()), getatt=True)
    assert_close(out[0], torch.from_numpy(golden["taB_x3_1b"]))
    assert_close(out[2], torch.from_numpy(golden["taB_x5"]))
    for i, a in enumerate(out[6]):
        assert_close(a, torch.from_numpy(golden[f"taB_att{i}"]))
    assert out[5].shape == (1, 2)


def test_nlb_module_golden(heads, golden):
    _, ta = heads
    for t in (2, 3, 10):
        x = torch.from_numpy(synth.normal(synth.stream_id(21, f"nlb_x{t}"), (t, 256))).to(dev())
        with torch.no_grad():
            z = ta.newnlb(x.t()[None].contiguous())[0].t()
        assert_close(z, torch.from_numpy(golden[f"nlb_T{t}_z"]))


def test_heads_refuse_cpu_tensors(heads):
    mp, ta = heads
    x = torch.from_numpy(synth.roi_features(31, 2))
    with pytest.raises(Exception):
        mp(x, torch.IntTensor([0, 1]))                     # CPU tensor: no fallback
    mp.train()
    try:
        with pytest.raises(Exception):
            mp(x, torch.IntTensor([0, 1]))                 # ... in train mode either (tests/test_gpu_train.py covers row f2)
    finally:
        mp.eval()


@pytest.fixture(scope="module")
def model_and_state():
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    return m.to(dev()).eval(), sd


def test_backbone_fpn_rpn_head(model_and_state):
    m, sd = model_and_state
    imgs = [torch.from_numpy(synth.frames(i, 1, 160, 224)[0]) for i in range(2)]
    m.transform.min_size, m.transform.max_size = 192, 320          # exercises the bilinear resize
    with torch.no_grad():
        feats, sizes, orig, padded = m.extract_features([i.to(dev()) for i in imgs])
        rpn = m.rpn.head(list(feats.values()))
    batch, osz = OD.transform(imgs, 192, 320)
    assert [tuple(s) for s in osz] == [tuple(s) for s in sizes] and tuple(batch.shape[-2:]) == tuple(padded)
    ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
    for k in ofe:
        assert_close(feats[k].permute(0, 3, 1, 2), ofe[k])
    oobj, odl = OD.rpn_head(list(ofe.values()), sd)
    for (obj, dlt), oo, od in zip(rpn, oobj, odl):
        assert_close(obj.permute(0, 3, 1, 2), oo)
        assert_close(dlt.permute(0, 3, 1, 2), od)


def test_body_on_two_streams_is_bit_identical(model_and_state):
    """SEAM_BODY_STREAMS knob: the ResNet body over batch slices on two HIP streams == one stream, bit for bit."""
    import seam_match_rcnn_amd.models.detection as det
    model, _ = model_and_state
    imgs = [torch.from_numpy(synth.frames(90 + i, 1, 96, 128)[0]).to(dev()) for i in range(5)]
    saved = model.transform.min_size, model.transform.max_size, det.BODY_STREAMS
    model.transform.min_size, model.transform.max_size = 96, 128
    try:
        with torch.no_grad():
            det.BODY_STREAMS = 1
            one, *_ = model.extract_features(imgs)
            det.BODY_STREAMS = 2
            two, *_ = model.extract_features(imgs)
        torch.cuda.synchronize()
    finally:
        model.transform.min_size, model.transform.max_size, det.BODY_STREAMS = saved
    for k in one:
        assert torch.equal(one[k], two[k]), k


def test_level_streams_are_bit_identical(model_and_state):
    """SEAM_LEVEL_STREAMS: the FPN output convs / RPN-head launches of the small pyramid levels on a side stream == everything on
    one stream, bit for bit (features, RPN head outputs, proposals), in fp32 and fp16."""
    import seam_match_rcnn_amd.models.detection as det
    model, _ = model_and_state
    imgs = [torch.from_numpy(synth.frames(95 + i, 1, 128, 160)[0]).to(dev()) for i in range(3)]
    saved = model.transform.min_size, model.transform.max_size, det.LEVEL_STREAMS
    model.transform.min_size, model.transform.max_size = 128, 160
    try:
        for dt in (torch.float32, torch.float16):
            model.set_compute_dtype(dt)
            got = {}
            for flag in (False, True):
                det.LEVEL_STREAMS = flag
                with torch.no_grad():
                    feats, sizes, orig, padded = model.extract_features(imgs)
                    head = model.rpn.head.fused(list(feats.values()))
                    props = model.rpn(feats, sizes, padded)
                    rois = [torch.from_numpy(synth.fixed_rois(8, 128, 160)).to(dev())] * len(imgs)
                    res, _, _ = model.forward_fixed_rois(imgs, rois)      # mask branch next to the match trunk
                torch.cuda.synchronize()
                got[flag] = (feats, head, props, res)
            for k in got[False][0]:
                assert torch.equal(got[False][0][k], got[True][0][k]), (dt, k)
            for a, b in zip(got[False][1], got[True][1]):
                assert torch.equal(a, b)
            for a, b in zip(got[False][2], got[True][2]):
                assert torch.equal(a, b)
            for a, b in zip(got[False][3], got[True][3]):
                for key in ("masks", "match_features", "roi_features"):
                    assert torch.equal(a[key], b[key]), (dt, key)
    finally:
        model.set_compute_dtype(torch.float32)
        model.transform.min_size, model.transform.max_size, det.LEVEL_STREAMS = saved


def test_fixed_roi_forward_c1(model_and_state):
    """BASELINE config 1 shape, scaled: fixed ROIs, 16-product gallery."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 256, 320
    imgs = [torch.from_numpy(synth.frames(7, 1, 256, 320)[0])]
    rois = [torch.from_numpy(synth.fixed_rois(8, 256, 320))]
    with torch.no_grad():
        res, feats, _ = m.forward_fixed_rois([i.to(dev()) for i in imgs], rois)
    ref, ofe, _ = OM.video_matchrcnn_forward(imgs, sd, fixed_rois=rois, with_rpn=False)
    # oracle ran with default 800/1333 -> rerun with the same transform sizes
    batch, sizes = OD.transform(imgs, 256, 320)
    ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
    orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], rois, sizes, 14)
    assert_close(res[0]["roi_features"], orf)
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    assert_close(res[0]["match_features"], OH.match_trunk(orf, mp))
    probs = OD.maskrcnn_inference(OD.mask_head(orf, sd), [torch.ones(8, dtype=torch.int64)])[0]
    assert_close(res[0]["masks"], probs)
    assert torch.equal(res[0]["w"].cpu(), mp["last.weight"])


def test_full_forward_with_rpn(model_and_state):
    """Whole drop-in forward incl. RPN proposals, box head, NMS, mask paste vs the oracle."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 800, 1333
    imgs = [torch.from_numpy(synth.frames(20 + i, 1, 192, 256)[0]) for i in range(2)]
    m.transform.min_size, m.transform.max_size = 192, 256
    with torch.no_grad():
        out = m([i.to(dev()) for i in imgs])
    import oracle.model as OMm
    feats, sizes, padded = OMm.extract_features(imgs, sd, 192, 256)
    props, _, _ = OMm.rpn_proposals(feats, sizes, padded, sd)
    ref = OMm.detect(feats, props, sizes, sd, 0.1)
    assert len(out) == 2
    for o, r in zip(out, ref):
        assert set(o) >= {"boxes", "labels", "scores", "masks", "match_features", "w", "b", "roi_features"}
        # detections are a discrete selection (top-k / NMS): exact as a set up to a printed, bounded list of near-tie flips
        from parity_sets import assert_same_set
        partner = assert_same_set(r["boxes"], o["boxes"], r["labels"], o["labels"], r["scores"], o["scores"], max_flips=2,
                                  what="full forward 192x256")
        ok = partner >= 0
        assert_close(o["scores"].cpu()[partner[ok]], r["scores"][ok], rtol=1e-4)
        assert o["roi_features"].shape[1:] == (256, 14, 14)
        assert o["masks"].shape[-2:] == imgs[0].shape[-2:]


def test_eval_with_targets_prepends_gt_boxes(model_and_state):
    """ref models/video_matchrcnn.py:256-262: with targets in eval, GT boxes lead every image's detections (score 1)."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 192, 256                 # identity scale
    imgs = [torch.from_numpy(synth.frames(30 + i, 1, 192, 256)[0]) for i in range(2)]
    targets = [dict(boxes=torch.tensor([[10., 20., 120., 150.]]), labels=torch.tensor([3])),
               dict(boxes=torch.tensor([[30., 40., 200., 180.], [5., 5., 60., 70.]]), labels=torch.tensor([1, 2]))]
    with torch.no_grad():
        out = m([i.to(dev()) for i in imgs], targets)
        ref = m([i.to(dev()) for i in imgs])
    for o, r, t in zip(out, ref, targets):
        n = t["labels"].numel()
        assert len(o["scores"]) == n + len(r["scores"])
        assert torch.equal(o["scores"][:n].cpu(), torch.ones(n)) and torch.equal(o["labels"][:n].cpu(), t["labels"])
        assert_close(o["boxes"][:n], t["boxes"], rtol=1e-6)
        assert o["roi_features"].shape[0] == n + len(r["scores"]) == o["match_features"].shape[0]
        assert_close(o["match_features"][n:], r["match_features"])
    # the GT ROI's descriptor equals the oracle's trunk on the oracle's RoIAlign of that box
    feats, sizes, padded = OM.extract_features(imgs, sd, 192, 256)
    orf = OD.multiscale_roi_align([feats[k] for k in "0123"], [t["boxes"] for t in targets], sizes, 14)
    ox3 = OH.match_trunk(orf, OM.sub(sd, "roi_heads.match_predictor."))
    assert_close(torch.cat([out[0]["match_features"][:1], out[1]["match_features"][:2]]), ox3)


def test_empty_detection_fallback_and_image_model(model_and_state):
    """No detection above threshold -> the full-image fallback box (score 0.1 video / 1.0 image model, label 0):
    ref models/video_matchrcnn.py:246-253, models/matchrcnn.py:373-379; the image model emits no roi_features."""
    from seam_match_rcnn_amd.models.matchrcnn import matchrcnn_resnet50_fpn
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 128, 160
    imgs = [torch.from_numpy(synth.frames(40 + i, 1, 128, 160)[0]).to(dev()) for i in range(2)]
    old = m.roi_heads.score_thresh
    m.roi_heads.score_thresh = 2.0
    try:
        with torch.no_grad():
            out = m(imgs)
    finally:
        m.roi_heads.score_thresh = old
    for o in out:
        assert o["boxes"].cpu().tolist() == [[0.0, 0.0, 160.0, 128.0]]
        assert o["labels"].cpu().tolist() == [0] and abs(float(o["scores"][0]) - 0.1) < 1e-7
        assert o["roi_features"].shape == (1, 256, 14, 14) and o["masks"].shape == (1, 1, 128, 160)
    m1 = matchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m1.load_state_dict({k: v for k, v in sd.items() if "temporal_aggregator" not in k})
    m1 = m1.to(dev()).eval()
    m1.transform.min_size, m1.transform.max_size = 128, 160
    m1.roi_heads.score_thresh = 2.0
    with torch.no_grad():
        out1 = m1(imgs)
    for o, ov in zip(out1, out):
        assert "roi_features" not in o and abs(float(o["scores"][0]) - 1.0) < 1e-7
        assert_close(o["match_features"], ov["match_features"])


def test_mixed_size_batch_full_forward(model_and_state):
    """Images of different sizes in one batch (per-image resize, common zero-padded canvas, per-image clip in
    the RPN / box decode): features, proposals-independent ROI features and detections vs the oracle."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 160, 288
    imgs = [torch.from_numpy(synth.frames(60, 1, 150, 200)[0]), torch.from_numpy(synth.frames(61, 1, 180, 170)[0])]
    with torch.no_grad():
        feats, sizes, orig, padded = m.extract_features([i.to(dev()) for i in imgs])
        out = m([i.to(dev()) for i in imgs])
    ofe, osz, opad = OM.extract_features(imgs, sd, 160, 288)
    assert [tuple(s) for s in osz] == [tuple(s) for s in sizes] and tuple(opad) == tuple(padded)
    assert len({tuple(s) for s in sizes}) == 2
    for k in ofe:
        assert_close(feats[k].permute(0, 3, 1, 2), ofe[k])
    props, _, _ = OM.rpn_proposals(ofe, osz, opad, sd)
    ref = OM.detect(ofe, props, osz, sd, 0.1)
    for o, r, sz, og in zip(out, ref, osz, orig):
        n = min(len(o["scores"]), len(r["scores"]))
        assert abs(len(o["scores"]) - len(r["scores"])) <= 2 and n > 0
        assert_close(o["scores"][:n // 2], r["scores"][:n // 2], rtol=5e-3)
        # boxes come back in ORIGINAL image pixels and inside the image
        b = o["boxes"]
        assert float(b[:, 0::2].max()) <= og[1] + 1e-3 and float(b[:, 1::2].max()) <= og[0] + 1e-3 and float(b.min()) >= 0
        assert o["masks"].shape[-2:] == tuple(og)


def test_hip_graph_replay_matches_eager(model_and_state):
    """The fixed-ROI step is capturable as one HIP graph (no allocation / host sync / H2D copy in steady state);
    replays on new input reproduce the eager result bit for bit."""
    m, sd = model_and_state
    m.transform.min_size, m.transform.max_size = 128, 160
    ta = m.roi_heads.temporal_aggregator
    frames = [torch.from_numpy(synth.frames(70 + i, 1, 128, 160)[0]).to(dev()) for i in range(3)]
    static = [f.clone() for f in frames]
    rois = [torch.from_numpy(synth.fixed_rois(8, 128, 160)).to(dev())] * 3
    types = torch.zeros(24, dtype=torch.int32)
    ids = torch.arange(8).repeat(3)
    bank = torch.from_numpy(synth.gallery(3, 50)).to(dev())

    def step(inp):
        res, _, _ = m.forward_fixed_rois(inp, rois)
        out = ta(torch.cat([r["roi_features"] for r in res]), types, ids)
        return out[0], ta.pair(out[0], bank)

    with torch.no_grad():
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            step(static); step(static)
        torch.cuda.current_stream().wait_stream(cap)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gout = step(static)
        new = [torch.from_numpy(synth.frames(80 + i, 1, 128, 160)[0]).to(dev()) for i in range(3)]
        for s_, n_ in zip(static, new):
            s_.copy_(n_)
        g.replay()
        torch.cuda.synchronize()
        eager = step(new)
    assert torch.equal(gout[0], eager[0]) and torch.equal(gout[1], eager[1])


def test_one_sync_forward_equals_list_form(model_and_state):
    """VideoMatchRCNN.forward hands the RPN's proposals to its RoI heads in the padded form (one device synchronisation per
    forward, at the detection counts); the reference's list form (ONE_SYNC = False) must give the same outputs bit for bit,
    on a mixed batch (images with different proposal counts) and with a score threshold nothing passes (fallback boxes)."""
    m, _ = model_and_state
    m.transform.min_size, m.transform.max_size = 192, 256
    imgs = [torch.from_numpy(synth.frames(50 + i, 1, 192, 256)[0]).to(dev()) for i in range(3)]
    imgs[1] = imgs[1][:, :128, :160].contiguous()              # a smaller image: fewer anchors, fewer proposals
    old = m.roi_heads.score_thresh
    try:
        for thr in (old, 2.0):
            m.roi_heads.score_thresh = thr
            outs = []
            for one in (True, False):
                m.ONE_SYNC = one
                with torch.no_grad():
                    outs.append(m(imgs))
            for a, b in zip(*outs):
                assert a.keys() == b.keys()
                for k in a:
                    assert torch.equal(a[k], b[k]), k
    finally:
        del m.ONE_SYNC                # back to the class default (by post-NMS proposal count)
        m.roi_heads.score_thresh = old


def test_padded_proposals_match_the_list(model_and_state):
    """RegionProposalNetwork.forward(padded_out=True): row i = the list form's proposals of image i, zeros behind."""
    m, _ = model_and_state
    m.transform.min_size, m.transform.max_size = 192, 256
    imgs = [torch.from_numpy(synth.frames(60 + i, 1, 192, 256)[0]).to(dev()) for i in range(2)]
    with torch.no_grad():
        feats, sizes, _, padded = m.extract_features(imgs)
        lst = m.rpn(feats, sizes, padded)
        pb, pc = m.rpn(feats, sizes, padded, padded_out=True)
    assert pb.shape == (2, m.rpn.post_nms_top_n, 4) and pc.tolist() == [len(p) for p in lst]
    for i, p in enumerate(lst):
        assert torch.equal(pb[i, :len(p)], p) and not bool(pb[i, len(p):].any())
