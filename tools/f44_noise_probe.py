#!/usr/bin/env python3
"""End-to-end noise of a Winograd F(4x4,3x3) step, EMULATED in torch fp32 (deliverable 2; GPU box).

The fp32 MFMA roof is fixed; the only lever above conv3x3_wino24pc is fewer multiplies: F(4x4,3x3) issues 36 per 16 outputs (2.25 per
output) instead of 24 per 8 (3).  Before anyone writes that kernel this probe answers the admissibility question: does the extra
rounding of the F(4,3) transform in BOTH directions stay inside the parity budget of the path?

Three runs of BASELINE configs[1] on one clip (10 frames 800x800), identical inputs and weights:
  exact    every 3x3 layer on the implicit GEMM (plain fp32 fma chains)                      -- the yardstick
  shipped  the dispatch of the product path (F(2x4) / F(2x2) Winograd kernels)
  f44      the stride-1 3x3 layers on maps of >= 50x50 pixels (the 200^2 / 100^2 / 50^2 layers F(4x4) would take) through
           an F(4x4,3x3) emulation in torch fp32: U = G g G^T in fp64 rounded once (as the kernels pack their weights), V = B^T d B
           as two separable fp32 passes, 36 position GEMMs in fp32, Y = A^T M A in fp32; everything else as shipped
and for `shipped` and `f44` against `exact`: max error / scale of the FPN levels, ROI features, descriptors, match logits; top-20
equality; proposal and detection set flips of the drop-in forward (tests/parity_sets.py rule: same label, coordinates within tol).
usage: f44_noise_probe.py [frames]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import torch.nn.functional as F
import seam_match_rcnn_amd.synth as synth
from seam_match_rcnn_amd import ops
from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
from seam_match_rcnn_amd.models.detection import resized_size
from parity_sets import pair_boxes

DEV = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R, G, TOPK = 32, 1000, 20

BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], dtype=torch.float32, device=DEV)
GM = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                  dtype=torch.float64, device=DEV)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32, device=DEV)
_U = {}


def f44_conv(x, wt, pad):
    """x NHWC fp32, wt [K,C,3,3] fp32 -> NHWC [N,Ho,Wo,K]; F(4x4,3x3), fp32 arithmetic apart from the one-off weight transform."""
    n, h, w, c = x.shape
    k = wt.shape[0]
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    ty, tx = (ho + 3) // 4, (wo + 3) // 4
    key = wt.data_ptr()
    if key not in _U:
        u = torch.einsum("ir,kcrs,js->ijck", GM, wt.double(), GM).float()           # [6,6,C,K], fp64 -> one rounding
        _U[key] = u.reshape(36, c, k).contiguous()
    u = _U[key]
    xp = F.pad(x, (0, 0, pad, 4 * tx + 2 - w - pad, pad, 4 * ty + 2 - h - pad))      # NHWC: pad W then H
    out = torch.empty((n, ty * 4, tx * 4, k), dtype=torch.float32, device=x.device)
    for i in range(n):                                                               # image by image: bounded workspace
        d = xp[i].unfold(0, 6, 4).unfold(1, 6, 4)                                    # [ty, tx, C, 6(y), 6(x)]
        v = torch.einsum("...yx,jx->...yj", d, BT)                                   # columns:  d B
        v = torch.einsum("iy,...yj->...ij", BT, v)                                   # rows:     B^T (d B)        [ty,tx,C,6,6]
        v = v.permute(3, 4, 0, 1, 2).reshape(36, ty * tx, c)
        m = torch.bmm(v, u)                                                          # [36, tiles, K]
        m = m.reshape(6, 6, ty, tx, k).permute(2, 3, 4, 0, 1)                        # [ty,tx,K,6,6]
        y = torch.einsum("...ij,bj->...ib", m, AT)
        y = torch.einsum("ai,...ib->...ab", AT, y)                                   # [ty,tx,K,4,4]
        out[i] = y.permute(0, 3, 1, 4, 2).reshape(ty * 4, tx * 4, k)
    return out[:, :ho, :wo, :]


_orig_conv2d = ops.conv2d
_WEIGHTS = {}       # id(PackedConv) -> OIHW weight (registered by the patched pack_conv)
_orig_pack = ops.pack_conv
F44 = {"on": False, "layers": 0}


def pack_conv_keep(weight, *a, **kw):
    pc = _orig_pack(weight, *a, **kw)
    if pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.dtype == torch.float32 and weight.dim() == 4:
        _WEIGHTS[id(pc)] = weight.detach().float().contiguous()
    return pc


def conv2d_f44(x, pc, relu=False, residual=None, out=None, out_f32=False, out_hw=None):
    wt = _WEIGHTS.get(id(pc))
    if not F44["on"] or wt is None or x.shape[1] * x.shape[2] < 2500 or out_hw is not None or wt.shape[1] != x.shape[3]:
        return _orig_conv2d(x, pc, relu, residual, out, out_f32, out_hw)
    F44["layers"] += 1
    y = f44_conv(x, wt, pc.pad)
    if pc.scale is not None:
        y = y * pc.scale
    if pc.shift is not None:
        y = y + pc.shift
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    y = y.contiguous()
    if out is not None:
        out.copy_(y)
        return out
    return y


ops.pack_conv = pack_conv_keep
ops.conv2d = conv2d_f44
import seam_match_rcnn_amd.models.detection as det        # noqa: E402
import seam_match_rcnn_amd.models.match_head as mh        # noqa: E402
for mod in (det, mh):
    if hasattr(mod, "ops"):
        assert mod.ops is ops

sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in synth.video_matchrcnn_state(5).items()}
model = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
model.load_state_dict(sd)
model = model.to(DEV).eval()
model.roi_heads.roi_features_contiguous = False
ta = model.roi_heads.temporal_aggregator
frames = torch.from_numpy(synth.frames(0, T, 800, 800)).to(DEV)
rh, rw, _ = resized_size(800, 800)
rois = [torch.from_numpy(synth.fixed_rois(R, rh, rw)).to(DEV) for _ in range(T)]
bank = torch.from_numpy(synth.gallery(7, G)).to(DEV)
types = torch.zeros(T * R, dtype=torch.int32)
ids = torch.arange(R, dtype=torch.int64).repeat(T)


def run(mode):
    ops.WINOGRAD = mode != "exact"
    F44["on"], F44["layers"] = mode == "f44", 0
    with torch.no_grad():
        res, feats, _ = model.forward_fixed_rois(list(frames.unbind(0)), rois, run_rpn_head=True)
        rf = torch.cat([r["roi_features"] for r in res])
        out = ta(rf, types, ids)
        x5 = ta.pair(out[0], bank)
        idx, score = ops.rank_topk(x5, TOPK)
        f2, sizes, orig, padded = model.extract_features([frames[0]])
        props = model.rpn(f2, sizes, padded)[0]
        dets = model([frames[0]])[0]
    torch.cuda.synchronize()
    n44 = F44["layers"]
    F44["on"] = False
    return dict(feats={k: v.clone() for k, v in feats.items()}, rf=rf.clone(), x3=out[0].clone(), x5=x5.clone(), idx=idx.clone(),
                props=props.clone(), boxes=dets["boxes"].clone(), labels=dets["labels"].clone(), scores=dets["scores"].clone(), n44=n44)


def err(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))


exact = run("exact")
rows = []
for mode in ("shipped", "f44"):
    r = run(mode)
    e = {"FPN " + k: err(r["feats"][k], exact["feats"][k]) for k in exact["feats"]}
    e.update(roi_features=err(r["rf"], exact["rf"]), x3_1b=err(r["x3"], exact["x3"]), match_logits=err(r["x5"], exact["x5"]))
    worst = max(e.values())
    same_rows = int((r["idx"] == exact["idx"]).all(1).sum())
    pp, pextra, _ = pair_boxes(exact["props"], r["props"], tol_px=1e-2)
    dp, dextra, _ = pair_boxes(exact["boxes"], r["boxes"], exact["labels"], r["labels"], tol_px=5e-2)
    print(f"== {mode}: layers through the F(4x4) emulation per pass: {r['n44']}")
    for k, v in e.items():
        print(f"   {k:14s} max err / scale {v:.3e}")
    print(f"   max_err_of_scale {worst:.3e}   top-{TOPK} rows identical {same_rows}/{r['idx'].shape[0]}")
    print(f"   proposals: {int((pp < 0).sum())} missing / {len(pextra)} extra of {len(exact['props'])};  "
          f"detections: {int((dp < 0).sum())} missing / {len(dextra)} extra of {len(exact['boxes'])}")
    rows.append((mode, worst, same_rows, int((pp < 0).sum()), len(pextra), int((dp < 0).sum()), len(dextra)))
print("SUMMARY mode max_err_of_scale top20_rows props_missing props_extra dets_missing dets_extra")
for r in rows:
    print("SUMMARY", *r)
