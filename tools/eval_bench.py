#!/usr/bin/env python3
"""Row f1 timing: the evaluator's retrieval section on device tables vs the NumPy oracle (the reference's form).
usage: eval_bench.py [n_products] [n_shop] [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import evaluator as EV
from seam_match_rcnn_amd.models.match_head import TemporalAggregationNLB
from test_evaluator import make_tables
from conftest import to_torch
from oracle import evaluator as OE

P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
Fr = int(sys.argv[3]) if len(sys.argv) > 3 else 10
tab = make_tables(9, n_products=P, n_shop=G, frames=Fr)
dev = torch.device("cuda:0")
sd = to_torch(synth.temporal_aggregator_state(12))
ta = TemporalAggregationNLB(); ta.load_state_dict(sd); ta = ta.to(dev).eval()
d = lambda k: torch.from_numpy(tab[k]).to(dev)
t = EV.DescriptorTables(shop_mat=d("shop_mat"), shop_aggr=d("shop_aggr"), shop_prods=tab["shop_prods"], shop_sources=tab["shop_sources"],
                        street_mat=d("street_mat"), street_aggr=d("street_aggr"), street_prods=tab["street_prods"],
                        street_imgs=tab["street_imgs"], street_scores=tab["street_scores"], street_boxes=d("street_boxes"),
                        tracklets_gt=d("tracklets_gt"), w=d("w"), b=d("b"), count_street=P)
EV.evaluate_tables(t, ta, frames_per_product=Fr)
torch.cuda.synchronize(); t0 = time.perf_counter()
rep = EV.evaluate_tables(t, ta, frames_per_product=Fr)
torch.cuda.synchronize(); gpu = time.perf_counter() - t0
t0 = time.perf_counter()
ref = OE.evaluate_tables(tab, sd, frames_per_product=Fr)
cpu = time.perf_counter() - t0
EV.evaluate_tables_per_product(t, ta, frames_per_product=Fr)
torch.cuda.synchronize(); t0 = time.perf_counter()
rep1 = EV.evaluate_tables_per_product(t, ta, frames_per_product=Fr)
torch.cuda.synchronize(); gpu1 = time.perf_counter() - t0
print(f"per-product form (round 2-3): {gpu1*1e3:.0f} ms ({gpu1/P*1e3:.2f} ms/product); batched form: {gpu*1e3:.0f} ms ({gpu/P*1e3:.3f} ms/product); "
      f"identical reports: {rep1.tables_text() == rep.tables_text() and rep1.frame_ranks == rep.frame_ranks}")
same = all((rep.counts[k] == ref[k]).all() for k in rep.counts)
print(f"{P} products x {Fr} frames ({len(tab['street_prods'])} street boxes) vs {G} shop entries: device {gpu*1e3:.0f} ms "
      f"({gpu/P*1e3:.2f} ms/product), NumPy oracle (fp32, {os.cpu_count()} cores visible) {cpu*1e3:.0f} ms ({cpu/P*1e3:.2f} ms/product), "
      f"{cpu/gpu:.1f}x; counters identical: {same}")
