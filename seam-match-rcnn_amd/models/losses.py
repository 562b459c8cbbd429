"""Weakly-supervised match losses, mirror of the loss classes of the reference's ``models/match_head.py``
that its training loops import (``stuffs/engine.py:11-12``): same class names, constructor and call
signatures, same target-building rules; the criterion (``nn.CrossEntropyLoss(weight=...)`` over [n,2]
logits) is the fused HIP kernel ``seam_ce2_fwd_bwd_f32``.

Target building is bookkeeping over a few dozen detections: it runs on the host from ONE device->host copy
of the positive-class logits (the reference does ``.max()`` / ``.argmax()`` / ``int()`` per image on device
tensors, one sync each).
"""
from __future__ import annotations

import numpy as np
import torch

from ..autograd import WeightedCE2Function


class WeightedCrossEntropy2:
    """``nn.CrossEntropyLoss(weight=w)`` for two classes on the device (mean reduction)."""

    def __init__(self, weight: torch.Tensor):
        self.weight = weight.to(torch.float32)

    def __call__(self, logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return WeightedCE2Function.apply(logits.reshape(-1, 2), target.reshape(-1), self.weight)


def _host(v) -> np.ndarray:
    return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)


def _rank_within_type(types: np.ndarray) -> np.ndarray:
    """Index of every detection among the detections of its own type == its row (street) / column (shop)
    in the [n_street, n_shop] logits (ref models/match_head.py:223-226,272-275)."""
    rev = np.zeros(len(types), dtype=np.int64)
    for t in (0, 1):
        sel = np.flatnonzero(types == t)
        rev[sel] = np.arange(len(sel))
    return rev


def _best_box_per_street_image(l1, types, prod_ids, img_ids, rev, threshold, products=None):
    """For every street image: the detection whose positive logit against the shop box of its product is the
    largest, kept when that logit exceeds ``threshold`` (ref :227-242 / :278-297).  Yields (det, shop_dets)."""
    out = []
    groups = [None] if products is None else products
    for pi in groups:
        imgs = np.unique(img_ids) if pi is None else np.unique(img_ids[prod_ids == pi])
        for ii in imgs:
            dets = np.flatnonzero(img_ids == ii)
            if types[dets[0]] == 1:
                continue
            prod = prod_ids[dets[0]] if pi is None else pi
            shop = np.flatnonzero((prod_ids == prod) & (types == 1))
            vals = l1[rev[dets], rev[shop]].reshape(-1)          # broadcasts like the reference's paired fancy index
            if vals.max() > threshold:
                out.append((int(dets[int(vals.argmax())]), shop))
    return out


class MatchLossWeak(object):
    """ref models/match_head.py:210-246."""

    def __init__(self, device, match_threshold=-10.0):
        self.criterion = WeightedCrossEntropy2(torch.tensor([1.0, 1.0]).to(device))
        self.match_threshold = match_threshold

    def __call__(self, logits, types, prod_ids, img_ids):
        ty, pr, im = _host(types), _host(prod_ids), _host(img_ids)
        rev = _rank_within_type(ty)
        l1 = logits.detach()[..., 1].cpu().numpy()
        gts = np.zeros(l1.shape, dtype=np.int64)
        for det, shop in _best_box_per_street_image(l1, ty, pr, im, rev, self.match_threshold):
            gts[rev[det], rev[shop]] = 1
        return self.criterion(logits.view(-1, 2), torch.from_numpy(gts).view(-1).to(logits.device))


class NEWBalancedAggregationMatchLossWeak(object):
    """ref models/match_head.py:252-360."""

    def __init__(self, device, temporal_aggregator, match_threshold=-10.0):
        self.criterion = WeightedCrossEntropy2(torch.tensor([1.0, 0.3]).to(device))
        self.match_threshold = match_threshold
        self.temporal_aggregator = temporal_aggregator

    def __call__(self, match_logits, types, prod_ids, img_ids, roi_features):
        ty, pr, im = _host(types), _host(prod_ids), _host(img_ids)
        rev = _rank_within_type(ty)
        l1 = match_logits.detach()[..., 1].cpu().numpy()
        cands = np.asarray([d for d, _ in _best_box_per_street_image(l1, ty, pr, im, rev, self.match_threshold,
                                                                      products=np.unique(pr))], dtype=np.int64)
        if cands.size == 0:            # not enough aggregation candidates (:298-300)
            return torch.tensor(0, dtype=torch.float32).to(match_logits.device)
        street, seq_ids, valid = [], [], []
        for pi in np.unique(pr[cands]):
            mine = cands[pr[cands] == pi]
            if mine.size < self.temporal_aggregator.n_frames:
                continue
            seq_ids += [len(valid)] * mine.size
            valid.append(pi)
            street.append(mine)
        if not valid:                  # not enough valid frames (:325-327)
            return torch.tensor(0, dtype=torch.float32).to(roi_features.device)
        shop = []
        for pi in valid:
            s = np.flatnonzero((pr == pi) & (ty == 1))
            if s.size != 1:
                raise ValueError("only one element tensors can be converted to Python scalars")   # torch.tensor([...]) at :336
            shop.append(int(s[0]))
        street = np.concatenate(street)
        feature_inds = torch.from_numpy(np.concatenate([street, np.asarray(shop, dtype=np.int64)]))
        seq = torch.tensor(seq_ids + [len(valid) + i for i in range(len(shop))], dtype=torch.int64)
        types_t = types if torch.is_tensor(types) else torch.as_tensor(ty)
        new_feats = roi_features[feature_inds.to(roi_features.device)]
        agg_logits = self.temporal_aggregator(new_feats, types_t[feature_inds], seq)[2]
        # one sequence per valid product, one column per valid product's shop box: the targets are the identity (:345-356)
        gts = torch.eye(len(valid), dtype=torch.int64)
        return self.criterion(agg_logits.view(-1, 2), gts.view(-1).to(agg_logits.device))


class MatchLossDF2(object):
    """ref models/match_head.py:363-379."""

    def __init__(self, device):
        self.criterion = WeightedCrossEntropy2(torch.tensor([1.0, 1.0]).to(device))

    def __call__(self, logits, types, raw_gt):
        ty, gt = _host(types), _host(raw_gt)
        gts = (gt[ty == 1][None, :] == gt[ty == 0][:, None]).astype(np.int64)
        return self.criterion(logits.view(-1, 2), torch.from_numpy(gts).view(-1).to(logits.device))


class AggregationMatchLossDF2(object):
    """ref models/match_head.py:382-438 (sequences need >= 3 street boxes, :405)."""

    def __init__(self, device, temporal_aggregator):
        self.criterion = WeightedCrossEntropy2(torch.tensor([1.0, 0.3]).to(device))
        self.temporal_aggregator = temporal_aggregator

    def __call__(self, types, roi_features, raw_gt):
        ty, gt = _host(types), _host(raw_gt)
        street_inds, shop_inds = np.flatnonzero(ty == 0), np.flatnonzero(ty == 1)
        street, seq_ids, valid = [], [], []
        for pi in np.unique(gt):
            if pi <= 0:
                continue
            mine = street_inds[gt[street_inds] == pi]
            if mine.size < 3:
                continue
            seq_ids += [len(valid)] * mine.size
            valid.append(pi)
            street.append(mine)
        if not valid:
            raise RuntimeError("torch.cat(): expected a non-empty list of Tensors")      # what the reference hits at :411
        feature_inds = torch.from_numpy(np.concatenate(street + [shop_inds]))
        seq = torch.tensor(seq_ids + [len(valid) + i for i in range(len(shop_inds))], dtype=torch.int64)
        types_t = types if torch.is_tensor(types) else torch.as_tensor(ty)
        agg_logits = self.temporal_aggregator(roi_features[feature_inds.to(roi_features.device)], types_t[feature_inds], seq)[2]
        gts = (gt[shop_inds][None, :] == np.asarray(valid)[:, None]).astype(np.int64)
        return self.criterion(agg_logits.view(-1, 2), torch.from_numpy(gts).view(-1).to(agg_logits.device))
