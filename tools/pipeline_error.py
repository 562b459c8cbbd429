#!/usr/bin/env python3
"""End-to-end numerical error of the fp32 pipeline against the CPU oracle on BASELINE configs[0] (one 800x800 frame,
8 fixed ROIs, 16-product gallery): max |got - ref| / max |ref| and relative L2 per stage, for the Winograd forms in use
(SEAM_WINOGRAD=0 / SEAM_WINOGRAD24=0 select the others).  GPU box; the oracle is the checker, nothing here is timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
from seam_match_rcnn_amd import ops
from oracle import detection as OD, heads as OH, model as OM

DEV = "cuda:0"
sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.video_matchrcnn_state(5).items()}
m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
m.load_state_dict(sd)
m = m.to(DEV).eval()
img = torch.from_numpy(synth.frames(0, 1, 800, 800)[0])
rois = torch.from_numpy(synth.fixed_rois(8, 800, 800))
bank = torch.from_numpy(synth.gallery(7, 16))
with torch.no_grad():
    res, feats, _ = m.forward_fixed_rois([img.to(DEV)], [rois])
    ta = m.roi_heads.temporal_aggregator
    x = res[0]["roi_features"]
    out = ta(x, torch.zeros(8, dtype=torch.int32), torch.arange(8))
    x5 = ta.pair(out[0], bank.to(DEV))
batch, sizes = OD.transform([img], 800, 1333)
ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], [rois], sizes, 14)
mp = OM.sub(sd, "roi_heads.match_predictor.")
tap = OM.sub(sd, "roi_heads.temporal_aggregator.")
oo = OH.temporal_aggregation_forward(orf, torch.zeros(8, dtype=torch.int32), torch.arange(8), tap)
ox5 = OH.pair_logits(oo[0], bank, tap["last.weight"], tap["last.bias"])
probs = OD.maskrcnn_inference(OD.mask_head(orf, sd), [torch.ones(8, dtype=torch.int64)])[0]


def err(name, got, ref):
    got, ref = got.detach().float().cpu().double(), ref.double()
    print(f"{name:28s} max/scale {float((got - ref).abs().max() / ref.abs().max()):.2e}   rel L2 {float((got - ref).norm() / ref.norm()):.2e}")


print("forms: WINOGRAD", ops.WINOGRAD, "WINOGRAD24", ops.WINOGRAD24)
for k in ofe:
    err("FPN level " + k, feats[k].permute(0, 3, 1, 2), ofe[k])
err("roi_features", x, orf)
err("match_features (x3)", res[0]["match_features"], OH.match_trunk(orf, mp))
err("x3_1b (aggregated)", out[0], oo[0])
err("match logits x5", x5, ox5)
err("mask probabilities", res[0]["masks"], probs)
