"""ORACLE (test infrastructure, not product code) -- model-forward orchestration.

Restates the eval branch of ``TemporalRoIHeads.forward`` (ref
models/video_matchrcnn.py:207-316) and of ``NewRoIHeads.forward`` (ref
models/matchrcnn.py:451-468) on top of oracle/detection.py and oracle/heads.py.
The surrounding ``GeneralizedRCNN.forward`` (transform -> backbone -> rpn ->
roi_heads -> postprocess) is torchvision's; PARITY UNPINNED (see detection.py).
"""
from __future__ import annotations

import torch

from . import detection as D
from . import heads as H


def sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def extract_features(images, p, min_size=800, max_size=1333):
    batch, sizes = D.transform(images, min_size, max_size)
    feats = D.fpn(D.resnet50_body(batch, p), p)
    return feats, sizes, tuple(batch.shape[-2:])


def rpn_proposals(feats, sizes, padded_hw, p):
    fl = list(feats.values())
    obj, dlt = D.rpn_head(fl, p)
    anchors = D.grid_anchors(padded_hw, [f.shape[-2:] for f in fl])
    props, _ = D.rpn_filter_proposals(obj, dlt, anchors, sizes)
    return props, obj, dlt


def detect(feats, proposals, sizes, p, fallback_score=0.1):
    """box branch + post-process + empty-image fallback
    (ref models/video_matchrcnn.py:225-253; fallback score 1.0 in models/matchrcnn.py:373-379)."""
    fl = [feats[k] for k in ("0", "1", "2", "3")]
    bf = D.multiscale_roi_align(fl, proposals, sizes, 7)
    logits, deltas = D.box_head(bf, p)
    boxes, scores, labels = D.postprocess_detections(logits, deltas, proposals, sizes)
    res = []
    for i in range(len(boxes)):
        if boxes[i].numel() > 0:
            res.append(dict(boxes=boxes[i], labels=labels[i], scores=scores[i]))
        else:
            res.append(dict(boxes=torch.tensor([[0.0, 0.0, float(sizes[i][1]), float(sizes[i][0])]]),
                            labels=torch.tensor([0]), scores=torch.tensor([fallback_score])))
    return res


def video_matchrcnn_forward(images, p, targets=None, fixed_rois=None, with_masks=True,
                            with_rpn=True, video=True):
    """Eval forward -> list of per-image dicts with the reference's keys
    ``boxes, labels, scores, masks, match_features, w, b, roi_features``.

    fixed_rois: optional list (per image) of [k_i,4] boxes in resized-image pixels; when
    given, RPN proposals / box head / NMS are bypassed (the "fixed ROI" configs of
    BASELINE.json) and these boxes are the detections (label 1, score 1)."""
    orig = [tuple(im.shape[-2:]) for im in images]
    feats, sizes, padded = extract_features(images, p)
    extras = {}
    if fixed_rois is None:
        props, _, _ = rpn_proposals(feats, sizes, padded, p)
        result = detect(feats, props, sizes, p, 0.1 if video else 1.0)
    else:
        if with_rpn:
            fl = list(feats.values())
            extras["rpn"] = D.rpn_head(fl, p)
        result = [dict(boxes=torch.as_tensor(b, dtype=torch.float32),
                       labels=torch.ones(len(b), dtype=torch.int64),
                       scores=torch.ones(len(b))) for b in fixed_rois]
    if targets is not None:           # eval with GT boxes prepended (ref :256-262)
        for t, r in zip(targets, result):
            r["boxes"] = torch.cat([t["boxes"], r["boxes"]])
            r["labels"] = torch.cat([t["labels"], r["labels"]])
            r["scores"] = torch.cat([torch.ones(t["labels"].numel()), r["scores"]])
    mask_props = [r["boxes"] for r in result]
    fl = [feats[k] for k in ("0", "1", "2", "3")]
    roi_feats = D.multiscale_roi_align(fl, mask_props, sizes, 14)           # ref :277
    if with_masks:
        ml = D.mask_head(roi_feats, p)                                      # ref :278-279
        probs = D.maskrcnn_inference(ml, [r["labels"] for r in result])     # ref :291
        for pr, r in zip(probs, result):
            r["masks"] = pr
    counts = [len(b) for b in mask_props]
    types = torch.tensor([0] * counts[0] + [1] * sum(counts[1:]), dtype=torch.int32)   # ref :299-307
    mp = sub(p, "roi_heads.match_predictor.")
    x3, _ = H.match_predictor_forward(roi_feats, types, mp)                 # ref :309
    off = 0
    for r, c in zip(result, counts):                                        # ref :310-314
        r["match_features"] = x3[off:off + c]
        r["w"], r["b"] = mp["last.weight"], mp["last.bias"]
        if video:
            r["roi_features"] = roi_feats[off:off + c]
        off += c
    # transform.postprocess: boxes back to original-image pixels, masks pasted
    for r, sz, o in zip(result, sizes, orig):
        r["boxes_resized"] = r["boxes"]
        r["boxes"] = D.rescale_boxes(r["boxes"], sz, o)
        if with_masks:
            r["masks"] = D.paste_masks_in_image(r["masks"], r["boxes"], o)
    return result, feats, extras
