"""ORACLE (test infrastructure, not product code) -- the weak match losses and one training step on the CPU.

Loop-form torch-CPU restatement of the loss classes the training loop uses
(ref models/match_head.py:210-246 ``MatchLossWeak``, :252-360 ``NEWBalancedAggregationMatchLossWeak``,
:363-379 ``MatchLossDF2``, :382-438 ``AggregationMatchLossDF2``) and of the grad-enabled pass of
ref stuffs/engine.py:158-185 (both heads in train mode, ``losses.backward()``), over the functional heads of
``oracle/heads.py``.  Gradients come from torch autograd on the CPU.

PINNED: tests/test_oracle_golden.py checks ``train_step`` (losses, logits, parameter gradients, BatchNorm
buffers) against fixtures captured by running the reference's own modules and loss classes
(tests/golden/make_train_golden.py).  Only tests/ may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import heads


def _reverse_index(types):
    rev = torch.zeros_like(types, dtype=torch.int64)
    for t in (0, 1):
        sel = (types == t).nonzero().view(-1)
        rev[sel] = torch.arange(sel.shape[0])
    return rev


def match_loss_weak_targets(logits, types, prod_ids, img_ids, threshold=-10.0):
    """gts [n_street, n_shop] of ``MatchLossWeak`` (ref :215-242)."""
    img_ids, prod_ids = torch.as_tensor(img_ids), torch.as_tensor(prod_ids)
    gts = torch.zeros(logits.shape[0], logits.shape[1], dtype=torch.int64)
    rev = _reverse_index(types)
    for ii in torch.unique(img_ids):
        if int(types[img_ids == ii][0]) == 1:
            continue
        prod = int(prod_ids[img_ids == ii][0])
        dets = (img_ids == ii).nonzero().view(-1)
        shop = ((prod_ids == prod) & (types == 1)).nonzero().view(-1)
        vals = logits[rev[dets], rev[shop], 1].view(-1)
        if vals.max() > threshold:
            gts[rev[dets[vals.argmax()]], rev[shop]] = 1
    return gts


def match_loss_weak(logits, types, prod_ids, img_ids, threshold=-10.0):
    gts = match_loss_weak_targets(logits.detach(), types, prod_ids, img_ids, threshold)
    return F.cross_entropy(logits.view(-1, 2), gts.view(-1), weight=torch.tensor([1.0, 1.0]))


def aggregation_plan(match_logits, types, prod_ids, img_ids, n_frames, threshold=-10.0):
    """(feature_inds, seq_ids, gts) of ``NEWBalancedAggregationMatchLossWeak`` (ref :264-356), or None."""
    img_ids, prod_ids = torch.as_tensor(img_ids), torch.as_tensor(prod_ids)
    rev = _reverse_index(types)
    cands = []
    for pi in torch.unique(prod_ids):
        mine = (prod_ids == pi).nonzero().view(-1)
        for ii in torch.unique(img_ids[mine]):
            if int(types[img_ids == ii][0]) == 1:
                continue
            dets = (img_ids == ii).nonzero().view(-1)
            shop = ((prod_ids == pi) & (types == 1)).nonzero().view(-1)
            vals = match_logits[rev[dets], rev[shop], 1].view(-1)
            if vals.max() > threshold:
                cands.append(int(dets[vals.argmax()]))
    if not cands:
        return None
    cands = torch.tensor(cands)
    valid, street, seq_ids = [], [], []
    for pi in torch.unique(prod_ids[cands]):
        c = cands[prod_ids[cands] == pi]
        if c.numel() < n_frames:
            continue
        seq_ids += [len(valid)] * c.numel()
        valid.append(int(pi))
        street.append(c)
    if not valid:
        return None
    shop = [int(((prod_ids == pi) & (types == 1)).nonzero().view(-1)) for pi in valid]
    feature_inds = torch.cat(street + [torch.tensor(shop)])
    seq_ids = torch.tensor(seq_ids + [len(valid) + i for i in range(len(shop))])
    n_street = sum(c.numel() for c in street)
    gts = torch.zeros(len(valid), len(shop), dtype=torch.int64)
    for i, sid in enumerate(seq_ids[:n_street].unique()):
        rows = (seq_ids == sid).nonzero().view(-1)
        prod = prod_ids[feature_inds[rows]][0]
        gts[i, valid.index(int(prod))] = 1
    return feature_inds, seq_ids, gts


def aggregation_loss(match_logits, types, prod_ids, img_ids, roi_features, ta_params, n_frames, bn_train=True):
    plan = aggregation_plan(match_logits.detach(), types, prod_ids, img_ids, n_frames)
    if plan is None:
        return torch.tensor(0.0), None
    feature_inds, seq_ids, gts = plan
    out = heads.temporal_aggregation_forward(roi_features[feature_inds], types[feature_inds], seq_ids, ta_params,
                                             bn_train=bn_train)
    return F.cross_entropy(out[2].view(-1, 2), gts.view(-1), weight=torch.tensor([1.0, 0.3])), out[2]


def match_loss_df2(logits, types, raw_gt):
    raw_gt = torch.as_tensor(raw_gt)
    gts = (raw_gt[types == 1].unsqueeze(0) == raw_gt[types == 0].unsqueeze(1)).view(-1).to(torch.int64)
    return F.cross_entropy(logits.view(-1, 2), gts, weight=torch.tensor([1.0, 1.0]))


def aggregation_loss_df2(types, roi_features, raw_gt, ta_params, bn_train=True):
    raw_gt = torch.as_tensor(raw_gt)
    street_inds, shop_inds = (types == 0).nonzero().view(-1), (types == 1).nonzero().view(-1)
    valid, street, seq_ids = [], [], []
    for pi in raw_gt.unique():
        if pi <= 0:
            continue
        c = street_inds[raw_gt[street_inds] == pi]
        if c.numel() < 3:
            continue
        seq_ids += [len(valid)] * c.numel()
        valid.append(int(pi))
        street.append(c)
    feature_inds = torch.cat(street + [shop_inds])
    seq = torch.tensor(seq_ids + [len(valid) + i for i in range(shop_inds.numel())])
    out = heads.temporal_aggregation_forward(roi_features[feature_inds], types[feature_inds], seq, ta_params, bn_train=bn_train)
    gts = (raw_gt[shop_inds].unsqueeze(0) == torch.tensor(valid).unsqueeze(1)).view(-1).to(torch.int64)
    return F.cross_entropy(out[2].view(-1, 2), gts, weight=torch.tensor([1.0, 0.3]))


def train_step(roi_features, types, prod_ids, img_ids, mp_params, ta_params, n_frames=3, weight_aggr=1.0):
    """The grad-enabled pass of ref stuffs/engine.py:158-185 with both heads in train mode.
    mp_params / ta_params: state dicts (torch CPU tensors); BatchNorm running buffers are updated IN PLACE.
    -> dict(logits, match_loss, aggregation_loss, agg_logits, grads_mp, grads_ta)."""
    buf = ("running_mean", "running_var", "num_batches_tracked")
    mp = {k: (v if k.endswith(buf) else v.clone().requires_grad_(True)) for k, v in mp_params.items()}
    ta = {k: (v if k.endswith(buf) else v.clone().requires_grad_(True)) for k, v in ta_params.items()}
    _, logits = heads.match_predictor_forward(roi_features, types, mp, bn_train=True)
    l_match = match_loss_weak(logits, types, prod_ids, img_ids)
    l_aggr, agg_logits = aggregation_loss(logits, types, prod_ids, img_ids, roi_features, ta, n_frames)
    total = l_match + weight_aggr * l_aggr
    total.backward()
    zero = lambda v: torch.zeros_like(v)  # noqa: E731
    return dict(logits=logits.detach(), match_loss=l_match.detach(), aggregation_loss=l_aggr.detach(),
                agg_logits=None if agg_logits is None else agg_logits.detach(),
                grads_mp={k: (v.grad if v.grad is not None else zero(v)) for k, v in mp.items() if not k.endswith(buf)},
                grads_ta={k: (v.grad if v.grad is not None else zero(v)) for k, v in ta.items() if not k.endswith(buf)})
