"""Evaluator-side retrieval on the device in fp32.

Counterpart of ``evaluate()`` in the reference's evaluate_movingfashion.py:15-445: descriptor collection
(:26-80), the per-product tracking + the seven rankings (:157-334) and the accuracy tables (:338-445).
The reference does this with NumPy fp16 matrices, a full ``argsort`` over the gallery per query and
O(P^2) Python list scans on the CPU; here every arithmetic step is one kernel call on device-resident fp32
tables and only the tiny per-product decisions (greedy tracklet linking over <= a few dozen boxes) run on
the host:

    compute_selfdist / compute_distances  -> seam_pair_logits_f32 + seam_match_scores_f32
    compute_ranking(...) == shop_index    -> seam_rank_of_f32        (rank of the true product, no argsort)
    AVG DESC mean / AVG,MAX DISTANCE      -> seam_score_reduce_f32 + seam_rank_of_scores_f32
    AGGR DESC                             -> TemporalAggregationNLB Mode B (seam_nlb_attnpool_f32) + seam_rank_of_f32
    box_iou                               -> seam_box_iou_f32

Deliberate deviation: fp32 instead of the reference's fp16 tables (:82-92) -- SURVEY 8f f1 asks for it;
tie rule "lower index first" instead of NumPy's unstable reversed argsort.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops

K_THRESHOLDS = (1, 5, 10, 20)          # ref evaluate_movingfashion.py:15


@dataclass
class DescriptorTables:
    """What evaluate_movingfashion.py:82-92 builds from the per-detection tuples; descriptor matrices live on
    the device (fp32), the bookkeeping columns on the host."""
    shop_mat: torch.Tensor              # [Ns,256] match_features of each product's shop box
    shop_aggr: torch.Tensor             # [Ns,256] temporal_aggregator descriptor of the same box (:43-45)
    shop_prods: np.ndarray              # [Ns] product index
    shop_sources: np.ndarray            # [Ns] 1 = "regular", else "hard" (:216-219)
    street_mat: torch.Tensor            # [Nq,256] match_features of every kept street detection
    street_aggr: torch.Tensor           # [Nq,256] temporal_aggregator trunk descriptors (:73-78)
    street_prods: np.ndarray            # [Nq]
    street_imgs: np.ndarray             # [Nq] frame index inside its clip
    street_scores: np.ndarray           # [Nq] detection confidence
    street_boxes: torch.Tensor          # [Nq,4]
    tracklets_gt: torch.Tensor          # [n,4] ground-truth boxes, indexed by frame index as the reference does (:205)
    w: torch.Tensor                     # [2,256] match_predictor.last.weight
    b: torch.Tensor                     # [2]
    count_street: int = 0
    product_keys: Optional[Sequence] = None


@torch.no_grad()
def collect_descriptors(model, data_loader, device, score_threshold: float = 0.0,
                        first_n_withvideo: Optional[int] = None, step: int = 11) -> DescriptorTables:
    """evaluate_movingfashion.py:26-92: run the model over (shop image, street frames...) batches, keep the
    largest shop box and every street detection above the score threshold, plus their aggregator descriptors."""
    agg = model.roi_heads.temporal_aggregator
    shop_mat, shop_aggr, shop_prods, shop_src, keys = [], [], [], [], []
    s_mat, s_aggr, s_prod, s_img, s_score, s_box, gts = [], [], [], [], [], [], []
    w = b = None
    count_products = count_street = 0
    for images, targets in data_loader:
        count_products += 1
        images = [im.to(device) for im in images]
        output = [o for x in range(0, len(images), step) for o in model(images[x:x + step])]
        keep0 = output[0]["scores"] >= score_threshold
        if not bool(keep0.any()):
            continue
        if w is None:
            w, b = output[0]["w"].detach(), output[0]["b"].detach()
        bs = output[0]["boxes"][keep0]
        maxind = int(((bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1])).argmax())
        one_i = torch.ones(1, dtype=torch.int32)
        shop_aggr.append(agg(output[0]["roi_features"][maxind].unsqueeze(0), one_i, torch.zeros(1, dtype=torch.int64))[1])
        shop_mat.append(output[0]["match_features"][maxind].unsqueeze(0))
        shop_prods.append(count_products - 1)
        shop_src.append(int(targets[0].get("source", 1)))
        keys.append(targets[0].get("i", count_products - 1))
        gts += [torch.as_tensor(t["tracklet"], dtype=torch.float32).view(-1)[:4] for t in targets[1:]]
        if first_n_withvideo is not None and count_products >= first_n_withvideo:
            continue
        count_street += 1
        feats = []
        for i, o in enumerate(output[1:]):
            sel = (o["scores"] >= score_threshold).nonzero().view(-1)
            if sel.numel() == 0:
                continue
            s_mat.append(o["match_features"][sel])
            s_box.append(o["boxes"][sel])
            s_score.append(o["scores"][sel])
            s_prod += [count_products - 1] * sel.numel()
            s_img += [i] * sel.numel()
            feats.append(o["roi_features"][sel])
        feats = torch.cat(feats, 0)
        n = feats.shape[0]
        seq = agg(feats, torch.zeros(n, dtype=torch.int32), torch.zeros(n, dtype=torch.int64))[3][1:]
        s_aggr.append(seq.reshape(-1, seq.shape[-1]))
    return DescriptorTables(
        shop_mat=torch.cat(shop_mat), shop_aggr=torch.cat(shop_aggr).reshape(len(shop_mat), -1),
        shop_prods=np.asarray(shop_prods), shop_sources=np.asarray(shop_src),
        street_mat=torch.cat(s_mat), street_aggr=torch.cat(s_aggr), street_prods=np.asarray(s_prod),
        street_imgs=np.asarray(s_img), street_scores=torch.cat(s_score).cpu().numpy(), street_boxes=torch.cat(s_box),
        tracklets_gt=torch.stack(gts).to(device) if gts else torch.zeros((0, 4), device=device),
        w=w, b=b, count_street=count_street, product_keys=keys)


def build_tracklets(simmat: np.ndarray, imgs: np.ndarray, scores: np.ndarray, threshold: float) -> List[List[int]]:
    """Greedy tracklet linking of one product's detections (evaluate_movingfashion.py:166-202).

    simmat [n,n] = compute_selfdist of the detections, imgs [n] their frame index, scores [n] the detection
    confidences.  Repeatedly: seed a tracklet with the most confident unused detection; while frames without
    a member remain, take the unused detection in such a frame with the highest similarity to ANY current
    member; link it if that similarity exceeds ``threshold`` (it then closes its frame), else stop.
    Returns tracklets as lists of LOCAL detection indices, in creation order."""
    n = len(imgs)
    imgs = np.asarray(imgs, dtype=np.int64)
    free = np.ones(n, dtype=bool)
    present = np.zeros(int(imgs.max()) + 1 if n else 0, dtype=bool)     # frames that have a detection at all
    present[imgs] = True
    tracks: List[List[int]] = []
    while free.any():
        cand = np.flatnonzero(free)
        start = int(cand[np.argmax(scores[cand])])
        members = [start]
        open_f = present.copy()                     # frames without a member yet
        open_f[imgs[start]] = False
        while open_f.any():
            # `free` still holds the members of the tracklet under construction: they are only retired when it is
            # closed, but their frames are closed, so they can never be picked again.  (Vectorised in round 6: the pool in
            # ascending detection order and the row-major argmax pick the same element as the reference's Python loops.)
            pool = np.flatnonzero(free & open_f[imgs])
            if pool.size == 0:
                break
            rows = np.sort(np.asarray(members))     # rows in detection order, like the reference
            sub = simmat[np.ix_(rows, pool)]
            r, c = divmod(int(np.argmax(sub)), sub.shape[1])
            if not sub[r, c] > threshold:
                break
            members.append(int(pool[c]))
            open_f[imgs[members[-1]]] = False       # (the earlier members' frames are closed already)
        free[members] = False
        tracks.append(members)
    return tracks


def build_tracklets_batch(blocks: np.ndarray, seg: np.ndarray, imgs_all: np.ndarray, scores_all: np.ndarray,
                          threshold: float) -> List[List[List[int]]]:
    """``build_tracklets`` for every product of a pass in ONE native call (``seam_host_build_tracklets``, csrc/seam_tracklets.hip;
    host code): ``blocks`` = the products' self-similarity blocks concatenated, ``seg`` their detection offsets."""
    from . import _native
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    n_seg, ndet = len(seg) - 1, int(seg[-1])
    blocks = np.ascontiguousarray(blocks, dtype=np.float32)
    imgs_all = np.ascontiguousarray(imgs_all, dtype=np.int64)
    scores_all = np.ascontiguousarray(scores_all, dtype=np.float64)
    members = np.empty(max(ndet, 1), dtype=np.int32)
    lens = np.empty(max(ndet, 1), dtype=np.int32)
    ntr = np.empty(max(n_seg, 1), dtype=np.int32)
    rc = _native.lib().seam_host_build_tracklets(blocks.ctypes.data, seg.ctypes.data, imgs_all.ctypes.data, scores_all.ctypes.data, n_seg,
                                                 float(threshold), members.ctypes.data, lens.ctypes.data, ntr.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"seam_host_build_tracklets failed: {rc}")
    out = []
    for s in range(n_seg):
        o, tracks, mo = int(seg[s]), [], 0
        for k in range(int(ntr[s])):
            ln = int(lens[o + k])
            tracks.append(members[o + mo:o + mo + ln].tolist())
            mo += ln
        out.append(tracks)
    return out


@dataclass
class RetrievalReport:
    k_thresholds: Sequence[int]
    counts: Dict[str, np.ndarray] = field(default_factory=dict)     # name -> hits per k threshold
    count_street: int = 0
    count_reg: int = 0
    count_hard: int = 0
    frames_per_product: int = 0
    track_lens: List[int] = field(default_factory=list)
    frame_ranks: List[int] = field(default_factory=list)            # all_ranks_list (:241)
    per_product: Dict = field(default_factory=dict)                 # accs_per_product (:224,333-334)

    def accuracy(self, name: str, subset: str = "") -> np.ndarray:
        """The numbers the reference prints (:338-437): per-frame tables divide by #products x frames_per_product,
        per-product tables by #products (of the subset)."""
        n = {"": self.count_street, "_reg": self.count_reg, "_hard": self.count_hard}[subset]
        denom = n * self.frames_per_product if name == "frame" else n
        return self.counts[name + subset] / max(denom, 1)

    def summary(self):
        """(ret1, ret2, ret3) as ``evaluate`` returns them (:341,350,355,445)."""
        return (float(self.accuracy("frame")[0]), float(self.accuracy("avg_desc")[0]), float(self.accuracy("aggr_desc")[0]))

    # ---- what the reference prints and writes at the end of evaluate() (:336-443) ---------------------------------------
    _TITLES = (("frame", ""), ("max_per_image", " Product Max"), ("avg_desc", " Product Avg Desc"), ("aggr_desc", " Product Aggr Desc"),
               ("avg_dist", " Product Avg Dist"), ("max_dist", " Product Max Dist"), ("max_score", " Product Max Score"))

    def tables_text(self) -> str:
        """The accuracy tables exactly as the reference prints them: all products, then "Regular ONLY", then "Hard ONLY"
        (those two without the "Product Max" table, which the reference keeps for all products only), the rank quartiles and the
        average track length.  (A subset without products prints nan where the reference raises ZeroDivisionError.)"""
        lines = []
        for sub, head in (("", None), ("_reg", "\n\n\n Regular ONLY"), ("_hard", "\n\n\n Hard ONLY")):
            if head is not None:
                lines.append(head)
            n = {"": self.count_street, "_reg": self.count_reg, "_hard": self.count_hard}[sub]
            for name, title in self._TITLES:
                if sub and name == "max_per_image":
                    continue
                denom = n * self.frames_per_product if name == "frame" else n
                for k, hits in zip(self.k_thresholds, self.counts[name + sub]):
                    lines.append("Top-%d Retrieval Accuracy%s: %1.4f" % (k, title, int(hits) / denom if denom else float("nan")))
                lines.append("*" * 50)
        ranks = np.asarray(self.frame_ranks)
        lines.append(f"Rank median: {np.median(ranks)}; rank 1st quartile: {np.percentile(ranks, 25)}; "
                     f"rank 3rd quartile: {np.percentile(ranks, 75)}")
        lines.append(f"Average Track Length: {float(np.asarray(self.track_lens).mean())}")
        return "\n".join(lines) + "\n"

    def perf_rows(self) -> np.ndarray:
        """The 8 x len(k) block the reference writes to logs_mf/<time>.csv (:439-443): rows 0-3 = per-frame, product-max,
        avg-desc and aggr-desc accuracies in percent, rows 4-7 zero."""
        perf = np.zeros((8, len(self.k_thresholds)))
        tq = self.count_street * self.frames_per_product
        perf[0] = np.asarray(self.counts["frame"], dtype=np.float32) / tq
        for row, name in ((1, "max_per_image"), (2, "avg_desc"), (3, "aggr_desc")):
            perf[row] = np.asarray(self.counts[name], dtype=np.float32) / self.count_street
        return perf * 100

    def save_artifacts(self, directory: str = ".") -> None:
        """The files the reference leaves behind: ``accs_per_product.pth`` (:336) and ``logs_mf/<time>.csv`` (:441-443)."""
        import os
        import time
        torch.save(self.per_product, os.path.join(directory, "accs_per_product.pth"))
        os.makedirs(os.path.join(directory, "logs_mf"), exist_ok=True)
        np.savetxt(os.path.join(directory, "logs_mf", str(time.time()) + ".csv"), self.perf_rows(), fmt="%02.2f", delimiter="\t")


_TABLES = ("frame", "max_per_image", "aggr_desc", "avg_desc", "avg_dist", "max_dist", "max_score")


@torch.no_grad()
def evaluate_tables(t: DescriptorTables, temporal_aggregator, k_thresholds: Sequence[int] = K_THRESHOLDS,
                    frames_per_product: int = 3, tracking_threshold: float = 0.3, max_dets_per_pass: int = 2048,
                    max_pairs_per_pass: int = 1 << 25) -> RetrievalReport:
    """evaluate_movingfashion.py:123-334 on device-resident tables, BATCHED over products: every stage is one launch over all
    products of a pass (self-similarity, tracklet IoUs, per-frame ranks, AVG DESC, AVG / MAX DISTANCE, AGGR DESC as Mode B with
    S = #products) and one device -> host copy; only the greedy tracklet linking (a few dozen boxes per product) runs per
    product, on the host.  A pass takes as many products as fit ``max_dets_per_pass`` detections (the self-similarity stage
    computes their all-pairs matrix and keeps its diagonal blocks) and ``max_pairs_per_pass`` (query, shop) pairs.
    Same arithmetic per product as ``evaluate_tables_per_product`` (the kernels work row by row / segment by segment):
    identical reports, tested."""
    ks = np.asarray(k_thresholds)
    rep = RetrievalReport(k_thresholds=tuple(k_thresholds), count_street=t.count_street, frames_per_product=frames_per_product)
    for name in _TABLES:
        for sub in ("", "_reg", "_hard"):
            rep.counts[name + sub] = np.zeros(len(ks), dtype=np.int64)
    dev = t.shop_mat.device
    aggr_w, aggr_b = temporal_aggregator.last.weight.detach(), temporal_aggregator.last.bias.detach()
    G = t.shop_mat.shape[0]

    def hit(name, rank, sub):
        h = (rank < ks).astype(np.int64)
        rep.counts[name] += h
        if name != "max_per_image":
            rep.counts[name + sub] += h
        return h

    # products in order, with their shop entry and their detections (ascending street index, as np.flatnonzero gives them)
    first_shop = {}
    for i, pr in enumerate(t.shop_prods.tolist()):
        first_shop.setdefault(pr, i)
    order = np.argsort(t.street_prods, kind="stable")
    sp_sorted = t.street_prods[order]
    todo = []
    for p in range(t.count_street):
        if p not in first_shop:
            continue
        lo, hi = np.searchsorted(sp_sorted, p, "left"), np.searchsorted(sp_sorted, p, "right")
        if hi == lo:        # before any device work; the reference fails here too (np.stack of an empty list, evaluate_movingfashion.py:211)
            raise ValueError(f"evaluate: product {p} has no street detections (the model's empty-image fallback box normally prevents this)")
        todo.append((p, first_shop[p], order[lo:hi]))

    def idx(a):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int64, device=dev)

    pos = 0
    while pos < len(todo):
        # ---- the products of this pass
        batch, ndet = [], 0
        while pos < len(todo) and (not batch or (ndet + len(todo[pos][2]) <= max_dets_per_pass
                                                  and (ndet + len(todo[pos][2])) * max(G, 1) <= max_pairs_per_pass)):
            batch.append(todo[pos])
            ndet += len(todo[pos][2])
            pos += 1
        offs = np.cumsum([0] + [len(d) for _, _, d in batch])
        dets_all = np.concatenate([d for _, _, d in batch]) if ndet else np.zeros((0,), dtype=np.int64)
        dets_d = idx(dets_all)
        mine_all = t.street_mat[dets_d]

        # ---- tracking (:166-214): self-similarity of every product's detections -- the diagonal blocks only, one launch
        # (round 6; the all-pairs matrix of a pass is ndet^2 = up to 2048^2 pairs for sum n_i^2 useful ones) -- to the host (copy 1)
        blocks = ops.pair_scores_blockdiag(mine_all.contiguous(), offs, t.w, t.b).cpu().numpy()
        tracks_all = build_tracklets_batch(blocks, offs, t.street_imgs[dets_all], t.street_scores[dets_all], tracking_threshold)
        # IoU of every tracklet member with the ground truth of every member's frame (copy 2); the reference indexes the GT list
        # by the frame index inside the clip (:205)
        imgs_all = t.street_imgs[dets_all]
        n_gt = int(imgs_all.max()) + 1 if ndet else 0
        iou_full = ops.box_iou(t.street_boxes[dets_d].contiguous(), t.tracklets_gt[:n_gt].contiguous())      # [ndet, n_gt]
        rows, cols, shapes = [], [], []
        for (p, _, dets), o, tracks in zip(batch, offs[:-1], tracks_all):
            im = t.street_imgs[dets]
            for members in tracks:
                mm = np.asarray(members)
                rows.append(np.repeat(o + mm, len(mm)))
                cols.append(np.tile(im[mm], len(mm)))
                shapes.append(len(mm))
        vals = iou_full[idx(np.concatenate(rows)), idx(np.concatenate(cols))].cpu().numpy()
        chosen, vo, ti = [], 0, 0
        for (p, _, dets), tracks in zip(batch, tracks_all):
            best, best_iou = 0, -np.inf
            for k, members in enumerate(tracks):
                m = shapes[ti]
                iou = float(vals[vo:vo + m * m].reshape(m, m).max(-1).sum())
                vo += m * m
                ti += 1
                if iou > best_iou:
                    best, best_iou = k, iou
            if not tracks:      # the reference fails here too (np.stack of an empty list, evaluate_movingfashion.py:211): say why
                raise ValueError(f"evaluate: product {p} has no street detections (the model's empty-image fallback box normally prevents this)")
            members = np.asarray(tracks[best])
            chosen.append(members[np.argsort(t.street_imgs[dets][members], kind="stable")])    # frames in unique_imgs order (:225)

        # ---- one query per tracked frame of every product (:225-239), all products at once
        seg = np.cumsum([0] + [len(m) for m in chosen])
        P, Q = len(batch), int(seg[-1])
        q_local = np.concatenate([o + m for o, m in zip(offs[:-1], chosen)])               # rows of mine_all
        q_d = idx(q_local)
        shop_idx = np.asarray([si for _, si, _ in batch])
        target_p = idx(shop_idx)
        target_q = idx(np.repeat(shop_idx, np.diff(seg)))
        queries = mine_all[q_d].contiguous()
        logits = ops.pair_logits(queries, t.shop_mat, t.w, t.b)
        frame_rank_d = ops.rank_of(logits, target_q)
        distances = ops.match_scores(logits)                                             # [Q, G]
        del logits
        seg_d = torch.as_tensor(seg, dtype=torch.int32, device=dev)
        # AVG DESC (:279-291), AVG & MAX DISTANCE (:293-315): segment reductions, then one rank launch each
        avg_rank_d = ops.rank_of(ops.pair_logits(ops.score_reduce_segments(queries, seg_d, "mean"), t.shop_mat, t.w, t.b), target_p)
        both = torch.cat([ops.score_reduce_segments(distances, seg_d, "mean"), ops.score_reduce_segments(distances, seg_d, "max")])
        dist_rank_d = ops.rank_of_scores(both, target_p.repeat(2))
        # AGGR DESC (:250-276): Mode B over the tracked frames' aggregator descriptors, S = #products sequences
        tmax = int(np.diff(seg).max())
        seq = torch.zeros((1 + tmax, P, t.street_aggr.shape[1]), device=dev)
        trow = np.concatenate([1 + np.arange(len(m)) for m in chosen])
        tcol = np.repeat(np.arange(P), np.diff(seg))
        seq[idx(trow), idx(tcol)] = t.street_aggr[dets_d[q_d]]
        mask = torch.as_tensor(np.arange(1 + tmax)[None, :] > np.diff(seg)[:, None], device=dev)
        desc = temporal_aggregator(None, None, None, x3_1_seq=seq, x3_1_mask=mask, x3_2=t.shop_aggr[:1])[0][:P]
        aggr_rank_d = ops.rank_of(ops.pair_logits(desc.contiguous(), t.shop_aggr, aggr_w, aggr_b), target_p)
        ranks = torch.cat([frame_rank_d, avg_rank_d, dist_rank_d, aggr_rank_d]).cpu().numpy()      # copy 3
        frame_rank_all, avg_rank = ranks[:Q], ranks[Q:Q + P]
        dist_rank, aggr_rank = ranks[Q + P:Q + 3 * P], ranks[Q + 3 * P:]

        # ---- the counters, product by product in the reference's order
        for j, ((p, shop_index, dets), members) in enumerate(zip(batch, chosen)):
            sub = "_reg" if t.shop_sources[shop_index] == 1 else "_hard"
            if sub == "_reg":
                rep.count_reg += 1
            else:
                rep.count_hard += 1
            rep.track_lens.append(len(members))
            frame_rank = frame_rank_all[seg[j]:seg[j + 1]]
            per = {"sfmr": np.zeros(len(ks)), "seamrcnn": np.zeros(len(ks))}
            for r in frame_rank:
                per["sfmr"] += hit("frame", r, sub)
            rep.frame_ranks += [int(r) for r in frame_rank]
            hit("max_per_image", frame_rank.min(), sub)                              # (:243-247)
            per["seamrcnn"] += hit("aggr_desc", int(aggr_rank[j]), sub)
            hit("avg_desc", int(avg_rank[j]), sub)
            hit("avg_dist", int(dist_rank[j]), sub)
            hit("max_dist", int(dist_rank[P + j]), sub)
            hit("max_score", int(frame_rank[int(np.argmax(t.street_scores[dets][members]))]), sub)   # (:317-328)
            key = t.product_keys[shop_index] if t.product_keys is not None else shop_index
            rep.per_product[key] = {"sfmr": per["sfmr"] / frames_per_product, "seamrcnn": per["seamrcnn"] / 1.0}
    return rep


@torch.no_grad()
def evaluate_tables_per_product(t: DescriptorTables, temporal_aggregator, k_thresholds: Sequence[int] = K_THRESHOLDS,
                    frames_per_product: int = 3, tracking_threshold: float = 0.3) -> RetrievalReport:
    """evaluate_movingfashion.py:123-334 on device-resident tables, ONE PRODUCT AT A TIME (seven device -> host copies per
    product: the round-2 form, sync-bound at ~1 ms per product).  Kept as the reference the batched ``evaluate_tables`` is tested
    against: both must produce identical reports."""
    ks = np.asarray(k_thresholds)
    rep = RetrievalReport(k_thresholds=tuple(k_thresholds), count_street=t.count_street, frames_per_product=frames_per_product)
    for name in _TABLES:
        for sub in ("", "_reg", "_hard"):
            rep.counts[name + sub] = np.zeros(len(ks), dtype=np.int64)
    dev = t.shop_mat.device
    aggr_w, aggr_b = temporal_aggregator.last.weight.detach(), temporal_aggregator.last.bias.detach()

    def hit(name, rank, sub):
        h = (rank < ks).astype(np.int64)
        rep.counts[name] += h
        if name != "max_per_image":
            rep.counts[name + sub] += h
        return h

    for p in range(t.count_street):
        where = np.flatnonzero(t.shop_prods == p)
        if where.size == 0:
            continue
        shop_index = int(where[0])
        sub = "_reg" if t.shop_sources[shop_index] == 1 else "_hard"
        if sub == "_reg":
            rep.count_reg += 1
        else:
            rep.count_hard += 1
        dets = np.flatnonzero(t.street_prods == p)
        imgs, scores = t.street_imgs[dets], t.street_scores[dets]
        dets_d = torch.as_tensor(dets, device=dev)
        mine = t.street_mat[dets_d]

        # ---- tracking (:166-214)
        simmat = ops.match_scores(ops.pair_logits(mine, mine, t.w, t.b)).cpu().numpy()
        tracks = build_tracklets(simmat, imgs, scores, tracking_threshold)
        boxes = t.street_boxes[dets_d]
        best, best_iou = 0, -np.inf
        for ti, members in enumerate(tracks):
            gt = t.tracklets_gt[torch.as_tensor(imgs[members], device=dev)]
            iou = float(ops.box_iou(boxes[torch.as_tensor(members, device=dev)], gt).cpu().numpy().max(-1).sum())
            if iou > best_iou:
                best, best_iou = ti, iou
        if not tracks:          # the reference fails here too (np.stack of an empty list, evaluate_movingfashion.py:211): say why
            raise ValueError("evaluate: a product has no street detections (the model's empty-image fallback box normally prevents this)")
        members = np.asarray(tracks[best])
        rep.track_lens.append(len(members))
        members = members[np.argsort(imgs[members], kind="stable")]          # frames visited in unique_imgs order (:225)

        # ---- one query per tracked frame (:225-239)
        m_d = torch.as_tensor(members, device=dev)
        target = torch.full((len(members),), shop_index, dtype=torch.int64, device=dev)
        logits = ops.pair_logits(mine[m_d], t.shop_mat, t.w, t.b)
        frame_rank = ops.rank_of(logits, target).cpu().numpy()
        distances = ops.match_scores(logits)                                     # [n_frames, G]
        per = {"sfmr": np.zeros(len(ks)), "seamrcnn": np.zeros(len(ks))}
        for r in frame_rank:
            per["sfmr"] += hit("frame", r, sub)
        rep.frame_ranks += [int(r) for r in frame_rank]
        hit("max_per_image", frame_rank.min(), sub)                              # (:243-247)

        # ---- AGGR DESC (:250-276): Mode B over the tracked frames' aggregator descriptors
        seq = torch.zeros((1 + len(members), 1, t.street_aggr.shape[1]), device=dev)
        seq[1:, 0] = t.street_aggr[dets_d[m_d]]
        mask = torch.zeros((1, 1 + len(members)), dtype=torch.bool, device=dev)
        desc = temporal_aggregator(None, None, None, x3_1_seq=seq, x3_1_mask=mask, x3_2=t.shop_aggr[shop_index:shop_index + 1])[0][:1]
        r = int(ops.rank_of(ops.pair_logits(desc.contiguous(), t.shop_aggr, aggr_w, aggr_b), target[:1]).cpu())
        per["seamrcnn"] += hit("aggr_desc", r, sub)

        # ---- AVG DESC (:279-291)
        avg = ops.score_reduce(mine[m_d].contiguous(), "mean").unsqueeze(0)
        hit("avg_desc", int(ops.rank_of(ops.pair_logits(avg, t.shop_mat, t.w, t.b), target[:1]).cpu()), sub)

        # ---- AVG & MAX DISTANCE (:293-315)
        both = torch.stack([ops.score_reduce(distances, "mean"), ops.score_reduce(distances, "max")])
        r2 = ops.rank_of_scores(both, target[:1].repeat(2)).cpu().numpy()
        hit("avg_dist", int(r2[0]), sub)
        hit("max_dist", int(r2[1]), sub)

        # ---- MAX CONFIDENCE SCORE (:317-328)
        hit("max_score", int(frame_rank[int(np.argmax(scores[members]))]), sub)

        key = t.product_keys[shop_index] if t.product_keys is not None else shop_index
        rep.per_product[key] = {"sfmr": per["sfmr"] / frames_per_product, "seamrcnn": per["seamrcnn"] / 1.0}
    return rep


@torch.no_grad()
def evaluate(model, data_loader, device, score_threshold: float = 0.0, k_thresholds: Sequence[int] = K_THRESHOLDS,
             frames_per_product: int = 3, tracking_threshold: float = 0.3, first_n_withvideo: Optional[int] = None,
             return_report: bool = False, verbose: bool = True, artifacts_dir: Optional[str] = None):
    """Same signature and return value (ret1, ret2, ret3) as the reference's ``evaluate``
    (evaluate_movingfashion.py:15-16,445).  ``verbose`` prints the accuracy tables as the reference does (:338-437);
    ``artifacts_dir`` (e.g. ".") also writes ``accs_per_product.pth`` and ``logs_mf/<time>.csv`` there (:336,441-443) -- the
    reference always writes them into the working directory; here that is opt-in."""
    tables = collect_descriptors(model, data_loader, device, score_threshold, first_n_withvideo)
    ids = getattr(getattr(data_loader, "dataset", None), "product_ids", None)
    if ids is not None:                     # the reference keys accs_per_product by dataset.product_ids[targets[0]["i"]] (:160)
        tables.product_keys = [ids[i] for i in tables.product_keys]
    rep = evaluate_tables(tables, model.roi_heads.temporal_aggregator, k_thresholds, frames_per_product, tracking_threshold)
    if verbose:
        print(rep.tables_text(), end="")
    if artifacts_dir is not None:
        rep.save_artifacts(artifacts_dir)
    return (rep.summary(), rep) if return_report else rep.summary()
