#!/usr/bin/env python3
"""Golden vectors of the evaluator's retrieval section, captured by running the reference's own ``evaluate()``
(/root/reference/evaluate_movingfashion.py:15-445) in this container.

Run:  python tests/golden/make_eval_golden.py          (needs /root/reference; this container only)

``evaluate_movingfashion`` imports cv2 / pycocotools / torchvision / tensorboard at module level (through datasets/, models/
video_matchrcnn.py, stuffs/): none of them is installed here, and ``evaluate()`` itself uses exactly one symbol of them,
``torchvision.ops.box_iou`` (:207).  So: the modules ``evaluate()`` never touches are replaced by empty stand-ins in
``sys.modules``; ``torchvision.ops.box_iou`` is given torchvision's public definition (area / intersection-over-union of xyxy
boxes) [TV]; ``models.match_head`` / ``models.nlb`` are the REAL reference modules, so ``model.roi_heads.temporal_aggregator`` is
the reference's ``TemporalAggregationNLB`` with the repo's synthetic weights.  The detector is replaced by canned per-image
outputs (tests/eval_scenarios.py) -- the part under test is everything after it: descriptor collection, fp16 tables, tracklet
linking, the seven rankings, the counters.

What is stored (outputs only, no reference source text): for each scenario the 21 hit-counter vectors, count_reg / count_hard /
count_street, track_lens, all_ranks_list, (ret1, ret2, ret3), accs_per_product, what it printed (the accuracy tables), the
`logs_mf/*.csv` row block it wrote, and the descriptor tables the reference built
(shop / street aggregator descriptors as it stored them, fp16) -- read out of ``evaluate``'s frame when it returns
(``sys.setprofile``), since the function itself only prints and returns three numbers.

Before a scenario is accepted the script checks that it is DECIDED, i.e. that the reference's fp16 arithmetic and an fp32
evaluation cannot rank differently: every score the oracle compares the true product's score with must differ from it by more
than 0.5 % (10 half-ulps of fp16), and no true product's score may fall below 1e-6 (fp16 flushes < 3e-8 to zero, where it would tie
with every far product), see ``oracle.evaluator.MARGIN_LOG``.
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import seam_match_rcnn_amd.synth as synth          # noqa: E402
import eval_scenarios as ES                         # noqa: E402

REF = "/root/reference"


def _box_iou(a, b):
    """torchvision.ops.box_iou [TV, public definition]: IoU of every xyxy box of `a` with every box of `b`."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter)


def import_reference_evaluate():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    pm = mod("pycocotools.mask")
    mod("pycocotools", mask=pm)
    ops = mod("torchvision.ops", box_iou=_box_iou)
    mod("torchvision", ops=ops)
    sys.path.insert(0, REF)
    import models                                               # the real package (models/__init__.py is empty of imports we lack?)
    # the three imports below exist only for the script's __main__ section (:449-506); evaluate() never touches them
    mod("datasets.MFDataset", MovingFashionDataset=None, get_dataloader=None)
    mod("datasets", MFDataset=sys.modules["datasets.MFDataset"])
    mod("models.video_matchrcnn", videomatchrcnn_resnet50_fpn=None)
    mod("stuffs.transform")
    mod("stuffs", transform=sys.modules["stuffs.transform"])
    import evaluate_movingfashion as EM
    from models.match_head import TemporalAggregationNLB
    return EM, TemporalAggregationNLB


COUNTERS = ["k_accs", "k_accs_avg", "k_accs_avg_desc", "k_accs_aggr_desc", "k_accs_avg_dist", "k_accs_max_dist", "k_accs_max_score",
            "k_accs_reg", "k_accs_hard", "k_accs_avg_desc_reg", "k_accs_avg_desc_hard", "k_accs_aggr_desc_reg", "k_accs_aggr_desc_hard",
            "k_accs_max_dist_reg", "k_accs_max_dist_hard", "k_accs_avg_dist_reg", "k_accs_avg_dist_hard", "k_accs_max_score_reg",
            "k_accs_max_score_hard"]
SCALARS = ["count_reg", "count_hard", "count_street", "count_products", "total_querys"]
TABLES = ["shop_prods", "shop_sources", "street_prods", "street_imgs", "street_scores", "street_boxes", "street_aggr_feats",
          "shop_aggregated_descrs"]         # shop_mat / street_mat are the canned inputs themselves (cast to fp16): not stored


def run_reference(EM, agg, name):
    loader, canned, params = ES.build(name)
    model = ES.CannedModel(canned, agg)
    grabbed = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code.co_name == "evaluate" and frame.f_code.co_filename.endswith("evaluate_movingfashion.py"):
            grabbed.update(frame.f_locals)

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                                            # evaluate() writes accs_per_product.pth and logs_mf/*.csv
        try:
            sys.setprofile(prof)
            with contextlib.redirect_stdout(io.StringIO()) as out, np.errstate(all="ignore"):
                ret = EM.evaluate(model, loader, torch.device("cpu"), **params)
        finally:
            sys.setprofile(None)
            os.chdir(cwd)
        per_product = torch.load(os.path.join(tmp, "accs_per_product.pth"), weights_only=False)
        csvs = sorted(os.listdir(os.path.join(tmp, "logs_mf")))
        csv_text = open(os.path.join(tmp, "logs_mf", csvs[0])).read()
    g = {"ret": np.asarray(ret, np.float64), "perf_csv": np.asarray(csv_text)}
    for k in COUNTERS:
        g[k] = np.asarray(grabbed[k], np.int64)
    for k in SCALARS:
        g[k] = np.asarray(int(grabbed[k]), np.int64)
    g["track_lens"] = np.asarray(grabbed["track_lens"], np.int64)
    g["all_ranks_list"] = np.asarray(grabbed["all_ranks_list"], np.int64).reshape(-1)
    for k in TABLES:
        a = np.asarray(grabbed[k])
        g["tab_" + k] = a.astype(np.float32) if a.dtype.kind == "f" else a
    keys = list(per_product)
    g["per_product_keys"] = np.asarray([str(k) for k in keys])
    g["per_product_sfmr"] = np.stack([np.asarray(per_product[k]["sfmr"], np.float64) for k in keys])
    g["per_product_seamrcnn"] = np.stack([np.asarray(per_product[k]["seamrcnn"], np.float64) for k in keys])
    return g, out.getvalue(), (loader, canned, params)


def oracle_margins(built):
    """The CPU oracle over the same loader, logging how decided every ranking is."""
    from conftest import to_torch
    from oracle import evaluator as OE
    loader, canned, params = built
    agg_sd = to_torch(ES.aggregator_state())
    OE.MARGIN_LOG = []
    try:
        tab = OE.collect_tables(ES.CannedModel(canned, None), loader, agg_sd, params["score_threshold"], params["first_n_withvideo"])
        out = OE.evaluate_tables(tab, agg_sd, frames_per_product=params["frames_per_product"],
                                 tracking_threshold=params["tracking_threshold"])
        margins = list(OE.MARGIN_LOG)
    finally:
        OE.MARGIN_LOG = None
    return out, margins


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(8)
    EM, TA = import_reference_evaluate()
    agg = TA().eval()
    agg.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in ES.aggregator_state().items()})
    store = {}
    for name in ("A", "B", "C"):
        g, text, built = run_reference(EM, agg, name)
        out, margins = oracle_margins(built)
        least_score = min(m for k, m in margins if k == "true_score")
        margins = [(k, m) for k, m in margins if k != "true_score"]
        worst = min(m for _, m in margins)
        print(f"scenario {name}: ret = {g['ret']}, count_street {int(g['count_street'])} reg/hard {int(g['count_reg'])}/{int(g['count_hard'])}, "
              f"track_lens {g['track_lens'].tolist()}, frame ranks {g['all_ranks_list'].tolist()}")
        print(f"  k_accs {g['k_accs'].tolist()}  aggr {g['k_accs_aggr_desc'].tolist()}  avg_desc {g['k_accs_avg_desc'].tolist()}  "
              f"avg_dist {g['k_accs_avg_dist'].tolist()}  max_dist {g['k_accs_max_dist'].tolist()}  max_score {g['k_accs_max_score'].tolist()}")
        print(f"  least decided comparison: {worst:.4f} relative ({len(margins)} rankings; kinds: "
              f"{ {k: round(min(m for kk, m in margins if kk == k), 4) for k in sorted(set(k for k, _ in margins))} })")
        print(f"  smallest score of a true product: {least_score:.3e}")
        assert worst > 5e-3, f"scenario {name} is not decided under fp16: least margin {worst}"
        assert least_score > 1e-6, f"scenario {name}: a true product's score ({least_score}) is not representable in fp16"
        g["stdout"] = np.asarray(text)              # every line evaluate() printed (the accuracy tables), verbatim
        for k, v in g.items():
            store[f"{name}_{k}"] = v
    path = os.path.join(HERE, "eval_golden.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, f"{os.path.getsize(path)} bytes, {len(store)} arrays")


if __name__ == "__main__":
    main()
