"""ORACLE (test infrastructure, not product code) -- the evaluator's retrieval section on the CPU.

A NumPy fp32 restatement of evaluate_movingfashion.py:26-334 (descriptor collection, closures ``compute_ranking`` /
``compute_distances`` / ``compute_selfdist``, the greedy tracklet builder and the seven rankings), kept in the
reference's own loop-per-product / full-argsort form so that it checks the device implementation
(``seam-match-rcnn_amd/evaluator.py``), which is organised differently (rank counting, batched frames).

PINNED against the reference's own ``evaluate()``: tests/golden/make_eval_golden.py imports
evaluate_movingfashion.py in the build container (stand-ins for the uninstalled modules it never calls, the real
``TemporalAggregationNLB``), runs it over the canned-detector datasets of tests/eval_scenarios.py and stores every
counter, the track lengths, the per-frame ranks and the descriptor tables in tests/golden/eval_golden.npz;
tests/test_evaluator.py::test_oracle_matches_reference_evaluate asserts this file reproduces them all.
Deviations on purpose: fp32 tables instead of fp16 (:82-92); ties broken towards the lower index (NumPy's reversed
unstable argsort is not reproducible) -- the pinned datasets are built so that neither can change a rank, which
``MARGIN_LOG`` measures.  Only tests/ (and the fixture generator) may import this module.
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


MARGIN_LOG = None      # set to a list to record (kind, relative gap between the true product's score and the nearest other score)


def _rank_of(score_row: np.ndarray, target: int, kind: str = "rank") -> int:
    if MARGIN_LOG is not None and score_row.size > 1:
        s = score_row.astype(np.float64)
        gap = np.abs(np.delete(s, target) - s[target]).min()
        MARGIN_LOG.append((kind, float(gap / max(abs(s[target]), 1e-30))))
        MARGIN_LOG.append(("true_score", float(s[target])))
    return int(np.flatnonzero(_ranking(score_row) == target)[0])


def collect_tables(model, data_loader, agg_params: dict, score_threshold: float = 0.0, first_n_withvideo=None, step: int = 11) -> dict:
    """evaluate_movingfashion.py:17-92: run `model` over the (shop picture, street frames...) batches and build the descriptor
    tables.  `model(images)` -> list of dicts with scores / boxes / match_features / roi_features / w / b; the aggregator
    descriptors come from ``heads.temporal_aggregation_forward`` with `agg_params` (:43-45 shop, :73-78 street)."""
    count_products = count_street = 0
    shop, street, aggr, gts = [], [], [], []
    w = b = None
    for images, targets in data_loader:
        count_products += 1
        output = [o for x in range(0, len(images), step) for o in model(images[x:x + step])]
        o0 = {k: v.detach().cpu() for k, v in output[0].items()}
        keep = o0["scores"] >= score_threshold
        if not bool(keep.any()):
            continue                                                              # (:34-35)
        if w is None:
            w, b = o0["w"].numpy(), o0["b"].numpy()
        bs = o0["boxes"][keep]
        maxind = int(((bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])).argmax())    # position among the KEPT boxes ...
        desc = heads.temporal_aggregation_forward(o0["roi_features"][maxind][None], torch.IntTensor([1]), torch.LongTensor([0]),
                                                  agg_params)[1]                  # ... used on the unfiltered list (:42,46), as is
        shop.append((o0["match_features"][maxind].numpy(), count_products - 1, desc.numpy().reshape(-1),
                     targets[0]["source"], targets[0]["i"]))
        gts += [np.asarray(t["tracklet"], np.float32).reshape(-1)[:4] for t in targets[1:]]
        if first_n_withvideo is not None and count_products >= first_n_withvideo:
            continue                                                              # gallery-only product (:50-51)
        count_street += 1
        feats = []
        for i, o in enumerate(output[1:]):
            o = {k: v.detach().cpu() for k, v in o.items()}
            for j in (o["scores"] >= score_threshold).nonzero().view(-1).tolist():
                street.append((o["match_features"][j].numpy(), count_products - 1, i, float(o["scores"][j]), o["boxes"][j].numpy()))
                feats.append(o["roi_features"][j][None])
        feats = torch.cat(feats, 0)
        n = feats.shape[0]
        seq = heads.temporal_aggregation_forward(feats, torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int64),
                                                 agg_params)[3][1:]
        aggr.append(seq.reshape(-1, seq.shape[-1]).numpy())
    return dict(shop_mat=np.stack([x[0] for x in shop]), shop_prods=np.asarray([x[1] for x in shop]),
                shop_aggr=np.stack([x[2] for x in shop]), shop_sources=np.asarray([x[3] for x in shop]),
                shop_datais=np.asarray([x[4] for x in shop]),
                street_mat=np.stack([x[0] for x in street]), street_prods=np.asarray([x[1] for x in street]),
                street_imgs=np.asarray([x[2] for x in street]), street_scores=np.asarray([x[3] for x in street], np.float32),
                street_boxes=np.stack([x[4] for x in street]), street_aggr=np.concatenate(aggr),
                tracklets_gt=np.stack(gts), w=w, b=b, count_street=count_street, count_products=count_products)


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
            if MARGIN_LOG is not None:
                MARGIN_LOG.append(("track_threshold", float(abs(sub[r, c] - threshold) / threshold)))
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
                r = _rank_of(row[0], shop_index, "frame")
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
        bump("aggr_desc", _rank_of(_scores(desc[None], tab["shop_aggr"], aggr_w, aggr_b)[0], shop_index, "aggr_desc"), sub)

        # AVG DESC (:279-291)
        avg = street[best_inds].mean(0)
        bump("avg_desc", _rank_of(_scores(avg[None], shop, w, b)[0], shop_index, "avg_desc"), sub)

        # AVG & MAX DISTANCE (:293-315)
        distances = np.stack(distances)
        bump("avg_dist", _rank_of(distances.mean(0), shop_index, "avg_dist"), sub)
        bump("max_dist", _rank_of(distances.max(0), shop_index, "max_dist"), sub)

        # MAX CONFIDENCE SCORE (:317-328)
        pick = best_inds[int(np.asarray(scores).argmax())]
        bump("max_score", _rank_of(_scores(street[pick][None], shop, w, b)[0], shop_index), sub)
    return out
