"""Synthetic MovingFashion-style retrieval datasets for the evaluator parity tests (test infrastructure).

The reference's ``evaluate(model, data_loader, device, ...)`` (evaluate_movingfashion.py:15-445) consumes a model that returns
one dict per image (``scores, boxes, match_features, roi_features, w, b``; :31-68) and a loader of ``(images, targets)`` pairs --
image 0 of each pair is the product's shop picture, the rest its street frames; ``targets[0]`` carries ``source`` / ``i``,
``targets[1:]`` the ground-truth ``tracklet`` box (:46-48).  This module builds such a loader with CANNED detections, so that the
retrieval logic after the detector (descriptor collection, tracklet linking, the seven rankings, the accuracy counters) can be
driven identically through (a) the imported reference -- tests/golden/make_eval_golden.py, this container only --, (b) the CPU
oracle and (c) the device evaluator.  Everything is derived from the repo's counter-based PRNG (seam-match-rcnn_amd/synth.py):
the fixture stores only what the reference computed.

Geometry of the canned ``match_features`` (so that fp16 tables -- the reference's, :82-92 -- and fp32 tables -- ours -- rank
alike): product p's shop descriptor sits at coordinate p (x SPACING) on axis 0; a street detection of p sits at p + drift + frac
with frac in [0.2, 0.4] (never equidistant from two products); distractor detections sit far out on axis 1; every other
coordinate carries N(0, 0.02^2) noise.  Distances are kept where fp16 softmax scores stay representable (a true product whose
score underflows to 0 in fp16 ties with every far product and the reference's rank becomes an artefact of argsort's tie order):
the generator asserts the smallest true-product score.  The pairwise classifier is ``w = [+c/2 u, -c/2 u]`` with u in [0.5,1.5]^256, i.e. the match logit
difference is -c * sum_k u_k (a_k - b_k)^2: a drift of D puts about 2 D other products ahead of the true one, which exercises
every k threshold of (1, 5, 10, 20).  ``roi_features`` are |N(0,1)| fields times a per-product channel gain, so that the REAL
aggregator trunk (conv_seq + pool + linear + BN, weights from synth) separates products.
"""
from __future__ import annotations

import numpy as np
import torch

import seam_match_rcnn_amd.synth as synth

C_PAIR = 0.4
SPACING = 0.5            # products sit SPACING apart on axis 0 (the coordinates below are in units of it)
C_AGGR = 0.02
AGG_SEED = 12


def aggregator_state():
    """The aggregator every side loads: synth.temporal_aggregator_state(AGG_SEED) -- trunk, non-local block, attention scorer --
    with its pairwise classifier ``last`` replaced by a distance-like one (the synthetic Kaiming ``last`` has random-sign weights:
    its scores are not a similarity, so no ranking under it would be decided).  numpy arrays, state-dict keys."""
    sd = synth.temporal_aggregator_state(AGG_SEED)
    u = synth.uniform(synth.stream_id(AGG_SEED, "eval_aggr_u"), (256,), 0.5, 1.5)
    sd["last.weight"] = np.stack([0.5 * C_AGGR * u, -0.5 * C_AGGR * u]).astype(np.float32)
    sd["last.bias"] = np.asarray([0.05, -0.05], np.float32)
    return sd


def classifier(seed):
    u = synth.uniform(synth.stream_id(seed, "eval_u"), (256,), 0.5, 1.5)
    u[:2] = 1.0
    w = np.stack([0.5 * C_PAIR * u, -0.5 * C_PAIR * u]).astype(np.float32)
    return w, np.asarray([0.05, -0.05], np.float32)


def _desc(seed, tag, a0, a1=0.0):
    v = synth.normal(synth.stream_id(seed, "eval_desc_" + tag), (256,), std=0.02)
    v[0], v[1] = a0, a1
    return v.astype(np.float32)


def _gain(seed, product):
    return synth.uniform(synth.stream_id(seed, f"eval_gain_{product}"), (256,), 0.1, 3.0)


def _roi(seed, tag, gain_product, mix_product=None, mix=0.0):
    """[256,14,14] |N(0,1)| field x the channel gain of `gain_product` (optionally blended towards another product's gain:
    the aggregated descriptor then lands between the two products -- how non-trivial AGGR-DESC ranks are made)."""
    f = np.abs(synth.normal(synth.stream_id(seed, "eval_roi_" + tag), (256, 14, 14)))
    g = _gain(seed, gain_product)
    if mix_product is not None:
        g = (1.0 - mix) * g + mix * _gain(seed, mix_product)
    return (f * g[:, None, None]).astype(np.float32)


def _gt_box(product, frame):
    j = (product * 7 + frame * 3) % 9 - 4
    return np.asarray([40 + 30 * frame + j, 50 + 10 * frame - j, 200 + 30 * frame + j, 330 + 10 * frame + j], np.float32)


def _jit(seed, tag, amp):
    return synth.uniform(synth.stream_id(seed, "eval_jit_" + tag), (4,), -amp, amp)


def _det(seed, tag, score, box, a0, a1, gain_product, mix_product=None, mix=0.0):
    return dict(score=np.float32(score), box=box.astype(np.float32), match=_desc(seed, tag, SPACING * a0, a1),
                roi=(seed, tag, gain_product, mix_product, mix))


FAR_BOX = np.asarray([500, 400, 620, 560], np.float32)


def _street_frames(seed, p, spec):
    """spec: list over frames of lists of (kind, score, a0) with kind in {"true", "far", "dup"}."""
    frames = []
    for f, dets in enumerate(spec):
        out = []
        for d, (kind, score, a0, mix_p, mix) in enumerate(dets):
            tag = f"p{p}_f{f}_d{d}"
            if kind == "true":
                out.append(_det(seed, tag, score, _gt_box(p, f) + _jit(seed, tag, 6.0), a0, 0.0, p, mix_p, mix))
            elif kind == "dup":      # a second plausible box of the same garment (lower confidence, slightly off)
                out.append(_det(seed, tag, score, _gt_box(p, f) + _jit(seed, tag, 25.0), a0, 0.35, p, mix_p, mix))
            else:                    # distractor: another object, far in descriptor space and in the image
                out.append(_det(seed, tag, score, FAR_BOX + _jit(seed, tag, 10.0), a0, 9.0, (p + 11) % 23))
        frames.append(out)
    return frames


def _shop(seed, p, extra=0, order=None, scores=None):
    """The shop picture's detections: the garment (largest box) + `extra` smaller boxes of other things."""
    dets = [_det(seed, f"p{p}_shop", 0.95, np.asarray([30, 20, 330, 420], np.float32) + _jit(seed, f"p{p}_shop", 5.0), float(p), 0.0, p)]
    for e in range(extra):
        dets.append(_det(seed, f"p{p}_shop_x{e}", 0.6 - 0.1 * e, np.asarray([400, 100, 480 + 20 * e, 220], np.float32), float(p) + 3.3, 3.0,
                         (p + 5 + e) % 23))
    if scores is not None:
        for d, s in zip(dets, scores):
            d["score"] = np.float32(s)
    if order is not None:
        dets = [dets[i] for i in order]
    return dets


def _frac(p, f, late=False):
    """Fractional part of a street detection's coordinate: 0.18-0.22, or 0.38-0.42 for the frames that sit one product further
    (so that the nearest product of an early frame and of a late frame are never at the same distance)."""
    return (0.4 if late else 0.2) + 0.01 * ((p + 2 * f) % 5 - 2)


def _sc(p, f, d=0):
    return 0.9 - 0.013 * f - 0.0007 * p - 0.21 * d     # distinct confidences


def scenario(name):
    """-> dict(products=[dict(shop=[det...], frames=[[det...]...], source, key, gts=[box...])], params=dict(...), seed)"""
    prods = []
    if name == "A":           # regular + hard sources, drifts for every k threshold, distractors (one of them the most confident box)
        seed, g, t = 71, 34, 4
        drift = [0, 0, 1, 0, 2, 3, 0, 5, 1, 10, 0, 2] + [0] * (g - 12)
        for p in range(g):
            spec = []
            for f in range(t):
                extra = (f % 2) if p % 3 == 0 else 0
                mixp, mix = ((p + 1) % g, 0.55) if p in (4, 7, 9) else (None, 0.0)
                dets = [("true", _sc(p, f), p + drift[p] + extra + _frac(p, f, bool(extra)), mixp, mix)]
                if p % 4 == 1 and f in (0, 2):
                    dets.insert(0, ("far", 0.97 - 0.01 * f if f == 0 else 0.41, p + 0.1 * f, None, 0.0))
                spec.append(dets)
            prods.append(dict(shop=_shop(seed, p, extra=p % 2), frames=_street_frames(seed, p, spec), source=1 if p % 2 == 0 else 2,
                              key=f"vid_{p:03d}", gts=[_gt_box(p, f) for f in range(t)]))
        params = dict(score_threshold=0.0, frames_per_product=t, tracking_threshold=0.3, first_n_withvideo=13)
    elif name == "B":         # ragged tracklets: an unlinked frame, a split track, clips of different lengths, duplicate boxes
        seed, g, t = 79, 20, 5
        for p in range(g):
            kind = p % 5
            tp = 3 if kind == 2 else t
            spec = []
            for f in range(tp):
                a0 = p + _frac(p, f)
                if kind == 0 and f == 3:
                    a0 = p + 9 + _frac(p, f)                  # too far from the others to be linked (self-score < 0.3)
                if kind == 1 and f >= 2:
                    a0 = p + 6 + _frac(p, f)                  # the track splits in two; the longer half wins on IoU
                if kind == 4:
                    a0 = p + (p // 5 + 1) * 2 + _frac(p, f)
                mixp, mix = ((p + 3) % g, 0.5) if kind in (1, 4) else (None, 0.0)
                dets = [("true", _sc(p, f), a0, mixp, mix)]
                if kind == 3:
                    dets.append(("dup", _sc(p, f, 1), a0 + 0.01, None, 0.0))
                spec.append(dets)
            prods.append(dict(shop=_shop(seed, p), frames=_street_frames(seed, p, spec), source=1 if p % 3 else 2, key=1000 + p,
                              gts=[_gt_box(p, f) for f in range(tp)]))
        params = dict(score_threshold=0.0, frames_per_product=t, tracking_threshold=0.3, first_n_withvideo=None)
    elif name == "C":         # score threshold 0.5: a product whose shop picture has no detection, a frame without one, filtered boxes
        seed, g, t = 73, 16, 3
        for p in range(g):
            spec = []
            for f in range(t):
                sc = _sc(p, f)
                if p == 5 and f == 1:
                    sc = 0.31                                  # the only box of this frame is below the threshold
                dets = [("true", sc, p + (p % 4) + _frac(p, f), None, 0.0)]
                if p % 3 == 0:
                    dets.append(("far", 0.45 - 0.01 * f, p + 0.2, None, 0.0))      # filtered out by the threshold
                if p % 5 == 1:
                    dets.append(("far", 0.55 + 0.01 * f, p + 0.4, None, 0.0))      # kept: its own tracklet
                spec.append(dets)
            if p == 2:
                shop = _shop(seed, p, extra=1, scores=[0.4, 0.3])                  # nothing passes: the product is skipped
            elif p == 7:
                # unsorted confidences with a rejected box in front: the reference picks the largest KEPT box but indexes the
                # UNFILTERED lists with its position (:39-45) -- restated as is
                shop = _shop(seed, p, extra=2, order=[1, 2, 0], scores=[0.9, 0.2, 0.7])
            else:
                shop = _shop(seed, p, extra=p % 3)
            prods.append(dict(shop=shop, frames=_street_frames(seed, p, spec), source=1 if p % 4 else 3, key=f"c{p}",
                              gts=[_gt_box(p, f) for f in range(t)]))
        params = dict(score_threshold=0.5, frames_per_product=t, tracking_threshold=0.3, first_n_withvideo=12)
    else:
        raise KeyError(name)
    return dict(products=prods, params=params, seed=seed, name=name)


class _Dataset:
    def __init__(self, keys):
        self.product_ids = keys


class Loader(list):
    """list of (images, targets) with the ``.dataset.product_ids`` the reference reads (:160)."""
    dataset: _Dataset


def build(name, device="cpu"):
    """-> (loader, canned) ; loader[i] = (images, targets); images are 1-element tensors holding the image id that
    ``canned[id]`` (a dict of torch tensors on `device`) answers for."""
    sc = scenario(name)
    w, b = classifier(sc["seed"])
    loader, canned = Loader(), {}
    loader.dataset = _Dataset([p["key"] for p in sc["products"]])
    nxt = 0
    for pi, p in enumerate(sc["products"]):
        images, targets = [], [dict(source=p["source"], i=pi)]
        for fi, dets in enumerate([p["shop"]] + p["frames"]):
            images.append(torch.tensor([float(nxt)]))
            canned[nxt] = dict(
                scores=torch.from_numpy(np.asarray([d["score"] for d in dets], np.float32)).to(device),
                boxes=torch.from_numpy(np.stack([d["box"] for d in dets])).to(device),
                labels=torch.ones(len(dets), dtype=torch.int64, device=device),
                match_features=torch.from_numpy(np.stack([d["match"] for d in dets])).to(device),
                roi_features=torch.from_numpy(np.stack([_roi(*d["roi"]) for d in dets])).to(device),
                w=torch.from_numpy(w).to(device), b=torch.from_numpy(b).to(device))
            if fi > 0:
                targets.append(dict(tracklet=p["gts"][fi - 1]))
            nxt += 1
        loader.append((images, targets))
    return loader, canned, sc["params"]


class CannedModel:
    """Stands where the detector stands: ``model(images)`` returns the canned dicts; ``model.roi_heads.temporal_aggregator`` is a
    REAL aggregator (the reference's, the oracle's or the device one)."""

    class _Heads:
        pass

    def __init__(self, canned, temporal_aggregator):
        self.canned = canned
        self.roi_heads = CannedModel._Heads()
        self.roi_heads.temporal_aggregator = temporal_aggregator

    def __call__(self, images, targets=None):
        return [dict(self.canned[int(round(float(im.reshape(-1)[0])))]) for im in images]

    def eval(self):
        return self
