#!/usr/bin/env python3
"""Golden vectors of the GRAD-ENABLED pass, captured by running the reference's own
``MatchPredictor`` / ``TemporalAggregationNLB`` in ``.train()`` with its own ``MatchLossWeak`` /
``NEWBalancedAggregationMatchLossWeak`` (the step of ref stuffs/engine.py:158-185) on the CPU.

Run:  python tests/golden/make_train_golden.py          (needs /root/reference; this container only)

Inputs and weights come from seam-match-rcnn_amd/synth.py (rebuilt from the seed by the tests); only outputs
are stored: logits, the two losses, parameter gradients (full when <= 4096 values, otherwise a fixed strided
sample + float64 sum / abs-sum) and the BatchNorm buffers after the step.  No reference source text is stored.
"""
import os
import sys
import types as pytypes

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import seam_match_rcnn_amd.synth as synth  # noqa: E402

sys.path.insert(0, "/root/reference")
pc, pm = pytypes.ModuleType("pycocotools"), pytypes.ModuleType("pycocotools.mask")
pc.mask = pm
sys.modules["pycocotools"], sys.modules["pycocotools.mask"] = pc, pm
from models.match_head import (MatchPredictor, TemporalAggregationNLB, MatchLossWeak,          # noqa: E402
                               NEWBalancedAggregationMatchLossWeak, MatchLossDF2, AggregationMatchLossDF2)

OUT = os.path.dirname(os.path.abspath(__file__))
SEED_MP, SEED_TA, SEED_X = 11, 12, 41
FULL = 4096


def scenario():
    """3 products: one shop box each (tag 1) + street frames with 1-2 boxes (tag 0); img ids are batch positions."""
    types, prod, img = [], [], []
    i = 0
    for p, boxes_per_frame in enumerate(([2, 1, 2], [1, 1, 1, 2], [1, 2, 1])):
        types.append(1); prod.append(10 + p); img.append(i); i += 1
        for nb in boxes_per_frame:
            for _ in range(nb):
                types.append(0); prod.append(10 + p); img.append(i)
            i += 1
    return types, prod, img


def to_torch(sd):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def pack(g, prefix, grads):
    for k, v in grads.items():
        a = v.detach().numpy().reshape(-1)
        if a.size <= FULL:
            g[f"{prefix}{k}"] = a.copy()
        else:
            step = a.size // FULL
            g[f"{prefix}{k}@{step}"] = a[::step].copy()
        g[f"{prefix}{k}#sum"] = np.asarray([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])


def main():
    torch.manual_seed(0)
    dev = torch.device("cpu")
    mp = MatchPredictor()
    mp.load_state_dict(to_torch(synth.match_predictor_state(SEED_MP)))
    ta = TemporalAggregationNLB()
    ta.load_state_dict(to_torch(synth.temporal_aggregator_state(SEED_TA)))
    ta.n_frames = 3
    mp.train(); ta.train()
    ty, prod, img = scenario()
    x = torch.from_numpy(synth.roi_features(SEED_X, len(ty)))
    types = torch.IntTensor(ty)
    g = {"types": np.asarray(ty, np.int32), "prod_ids": np.asarray(prod, np.int64), "img_ids": np.asarray(img, np.int64)}

    match_loss = MatchLossWeak(dev)
    aggr_loss = NEWBalancedAggregationMatchLossWeak(dev, ta)
    _, logits = mp(x, types)
    l1 = match_loss(logits, types, prod, img)
    l2 = aggr_loss(logits, types, prod, img, x)
    (l1 + 1.0 * l2).backward()
    g["logits"] = logits.detach().numpy()
    g["match_loss"], g["aggregation_loss"] = l1.detach().numpy(), l2.detach().numpy()
    pack(g, "mp.", {k: p.grad for k, p in mp.named_parameters()})
    pack(g, "ta.", {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in ta.named_parameters()})
    for nm, m in (("mp", mp), ("ta", ta)):
        bn = m.linear[1]
        g[f"{nm}.bn_mean"], g[f"{nm}.bn_var"] = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
        g[f"{nm}.bn_n"] = np.asarray(int(bn.num_batches_tracked))

    # DeepFashion2-style losses (ref :363-438) on the same ROIs: raw_gt = product id, 0 = unlabeled
    mp2 = MatchPredictor(); mp2.load_state_dict(to_torch(synth.match_predictor_state(SEED_MP))); mp2.train()
    ta2 = TemporalAggregationNLB(); ta2.load_state_dict(to_torch(synth.temporal_aggregator_state(SEED_TA))); ta2.train()
    raw_gt = [p - 9 for p in prod]
    raw_gt[4] = 0
    _, logits2 = mp2(x, types)
    d1 = MatchLossDF2(dev)(logits2, types, raw_gt)
    d2 = AggregationMatchLossDF2(dev, ta2)(types, x, raw_gt)
    (d1 + d2).backward()
    g["df2_raw_gt"] = np.asarray(raw_gt, np.int64)
    g["df2_match_loss"], g["df2_aggregation_loss"] = d1.detach().numpy(), d2.detach().numpy()
    pack(g, "df2.mp.", {k: p.grad for k, p in mp2.named_parameters() if k.startswith(("last", "linear"))})
    pack(g, "df2.ta.", {k: p.grad for k, p in ta2.named_parameters() if k.startswith(("last", "attention", "newnlb.W", "newnlb.concat"))})

    np.savez_compressed(os.path.join(OUT, "train_golden.npz"), **g)
    print("wrote train_golden.npz:", len(g), "arrays,", sum(v.nbytes for v in g.values()) // 1024, "KiB raw")
    print("losses", float(l1), float(l2), float(d1), float(d2), "logit range", float(logits[..., 1].min()), float(logits[..., 1].max()))


if __name__ == "__main__":
    main()
