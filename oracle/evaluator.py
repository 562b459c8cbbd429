"""ORACLE (test infrastructure, not product code) -- the evaluator's retrieval section on the CPU.

A NumPy fp32 restatement of evaluate_movingfashion.py:94-334 (closures ``compute_ranking`` /
``compute_distances`` / ``compute_selfdist``, the greedy tracklet builder and the seven rankings), kept in the
reference's own loop-per-product / full-argsort form so that it checks the device implementation
(``seam-match-rcnn_amd/evaluator.py``), which is organised differently (rank counting, batched frames).

PARITY UNPINNED: ``evaluate`` cannot be imported here (module-level imports of cv2 / torchvision /
pycocotools datasets, and the code under test is nested inside one 430-line function), so this file is
checked only against hand-computed cases in tests/test_evaluator.py.  Deviations on purpose: fp32 tables
instead of fp16 (:82-92); ties broken towards the lower index (NumPy's reversed unstable argsort is not
reproducible).  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import torch

from . import heads


def _scores(q: np.ndarray, g: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """softmax((g - q)^2 @ w.T + b)[..., 1]   (:94-98 / :102-107 / :115-121)."""
    sq = (g[np.newaxis].astype(np.float32) - q[:, np.newaxis].astype(np.float32)) ** 2
    raw = sq @ w.T.astype(np.float32) + b.astype(np.float32)
    raw = raw - raw.max(2, keepdims=True)
    e = np.exp(raw)
    return (e / e.sum(2, keepdims=True))[:, :, 1]


def _ranking(score_row: np.ndarray) -> np.ndarray:
    """descending order, ties -> lower index first (the reference: np.argsort(x)[::-1], :99)."""
    return np.argsort(-score_row, kind="stable")


def _rank_of(score_row: np.ndarray, target: int) -> int:
    return int(np.flatnonzero(_ranking(score_row) == target)[0])


def box_iou(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """torchvision.ops.box_iou [TV] as used at :207."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(a[:, None, :2], b[None, :, :2])
    rb = np.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter)


def track_product(simmat, all_inds, imgs_of, scores_of, threshold):
    """:166-202, list-based like the reference; returns (tracklets_inds, tracklets_imgs) with GLOBAL indices."""
    all_inds = list(all_inds)
    unique_imgs = sorted(set(imgs_of[i] for i in all_inds))
    taken, tr_inds, tr_imgs = [], [], []
    while len(taken) < len(all_inds):
        remaining = [i for i in all_inds if i not in taken]
        start = remaining[int(np.argmax([scores_of[i] for i in remaining]))]
        t_inds, t_imgs = [start], [imgs_of[start]]
        to_check = [f for f in unique_imgs if f != imgs_of[start]]
        while to_check:
            cols = [c for c, i in enumerate(all_inds) if imgs_of[i] in to_check and i not in taken]
            if not cols:
                break
            rows = [r for r, i in enumerate(all_inds) if i in t_inds]
            sub = simmat[rows][:, cols]
            r, c = np.unravel_index(sub.argmax(), sub.shape)
            if sub[r, c] > threshold:
                t_inds.append(all_inds[cols[c]])
                t_imgs.append(imgs_of[all_inds[cols[c]]])
                to_check = [f for f in to_check if f not in t_imgs]
            else:
                break
        taken += t_inds
        tr_inds.append(t_inds)
        tr_imgs.append(t_imgs)
    return tr_inds, tr_imgs


def evaluate_tables(tab: dict, agg_params: dict, k_thresholds=(1, 5, 10, 20), frames_per_product=3,
                    tracking_threshold=0.3) -> dict:
    """tab: numpy fp32 arrays under the names of seam-match-rcnn_amd/evaluator.DescriptorTables.
    agg_params: TemporalAggregationNLB state dict (torch CPU tensors) for the AGGR DESC ranking."""
    ks = list(k_thresholds)
    names = ["frame", "max_per_image", "aggr_desc", "avg_desc", "avg_dist", "max_dist", "max_score"]
    out = {n + s: np.zeros(len(ks), dtype=np.int64) for n in names for s in ("", "_reg", "_hard")}
    out.update(count_reg=0, count_hard=0, track_lens=[], frame_ranks=[])
    w, b = tab["w"], tab["b"]
    aggr_w, aggr_b = agg_params["last.weight"].numpy(), agg_params["last.bias"].numpy()
    shop, street = tab["shop_mat"], tab["street_mat"]

    def bump(name, rank, sub, with_sub=True):
        for j, k in enumerate(ks):
            if rank < k:
                out[name][j] += 1
                if with_sub:
                    out[name + sub][j] += 1

    for p in range(tab["count_street"]):
        if p not in tab["shop_prods"]:
            continue
        shop_index = int(np.flatnonzero(tab["shop_prods"] == p)[0])
        sub = "_reg" if tab["shop_sources"][shop_index] == 1 else "_hard"
        out["count_reg" if sub == "_reg" else "count_hard"] += 1
        inds = np.flatnonzero(tab["street_prods"] == p)
        unique_imgs = np.unique(tab["street_imgs"][inds])
        simmat = _scores(street[inds], street[inds], w, b)
        tr_inds, tr_imgs = track_product(simmat, inds, tab["street_imgs"], tab["street_scores"], tracking_threshold)
        ious = []
        for ti, tim in zip(tr_inds, tr_imgs):
            ious.append(box_iou(tab["street_boxes"][ti], tab["tracklets_gt"][tim]).max(-1).sum())
        tid = int(np.argmax(np.stack(ious)))
        out["track_lens"].append(len(tr_inds[tid]))
        track_inds, track_imgs = np.asarray(tr_inds[tid]), np.asarray(tr_imgs[tid])

        ranks, best_inds, distances, scores = [], [], [], []
        for ii in unique_imgs:
            if (track_imgs == ii).sum() > 0:
                q = track_inds[np.flatnonzero(track_imgs == ii)]
                row = _scores(street[q], shop, w, b)
                r = _rank_of(row[0], shop_index)
                best_inds.append(q[0])
                ranks.append(r)
                bump("frame", r, sub)
                distances.append(row[0])
                scores.append(tab["street_scores"][q])
        out["frame_ranks"] += ranks
        bump("max_per_image", int(np.min(ranks)), sub, with_sub=False)
        best_inds = np.asarray(best_inds)

        # AGGR DESC (:250-276)
        seq = torch.zeros((1 + len(best_inds), 1, 256))
        seq[1:, 0] = torch.from_numpy(tab["street_aggr"][best_inds])
        mask = torch.zeros((1, 1 + len(best_inds)), dtype=torch.bool)
        desc = heads.temporal_aggregation_forward(None, None, None, agg_params, x3_1_seq=seq, x3_1_mask=mask,
                                                  x3_2=torch.from_numpy(tab["shop_aggr"][shop_index:shop_index + 1]))[0][0].numpy()
        bump("aggr_desc", _rank_of(_scores(desc[None], tab["shop_aggr"], aggr_w, aggr_b)[0], shop_index), sub)

        # AVG DESC (:279-291)
        avg = street[best_inds].mean(0)
        bump("avg_desc", _rank_of(_scores(avg[None], shop, w, b)[0], shop_index), sub)

        # AVG & MAX DISTANCE (:293-315)
        distances = np.stack(distances)
        bump("avg_dist", _rank_of(distances.mean(0), shop_index), sub)
        bump("max_dist", _rank_of(distances.max(0), shop_index), sub)

        # MAX CONFIDENCE SCORE (:317-328)
        pick = best_inds[int(np.asarray(scores).argmax())]
        bump("max_score", _rank_of(_scores(street[pick][None], shop, w, b)[0], shop_index), sub)
    return out
