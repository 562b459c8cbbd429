"""ORACLE (test infrastructure, not product code) -- torchvision-owned stages on the CPU.

PARITY UNPINNED vs. the real torchvision: the reference contains zero lines of
backbone / FPN / RPN / RoIAlign / NMS arithmetic (it only configures torchvision
objects -- ref models/video_matchrcnn.py:6-9,337-338; models/matchrcnn.py:2-3,11-28),
torchvision is not installed in this image and its version is unpinned upstream
(ref README.md:15-16).  Every function restates torchvision's *published* algorithm
("classic" <=0.12 layout, FrozenBN eps=1e-5) on top of ATen
CPU ops -- the same conv/linear/interpolate kernels the reference would dispatch --
and is pinned by analytic known-answer tests (tests/test_oracle_detection.py).

Parameters: flat dict keyed by torchvision's state-dict names.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

IMAGE_MEAN = (0.485, 0.456, 0.406)
IMAGE_STD = (0.229, 0.224, 0.225)
FROZEN_BN_EPS = 1e-5
BBOX_XFORM_CLIP = math.log(1000.0 / 16)
RESNET50_LAYERS = ((3, 64, 1), (4, 128, 2), (6, 256, 2), (3, 512, 2))
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)


# --------------------------------------------------------------------------- a2
def resized_size(h: int, w: int, min_size: int = 800, max_size: int = 1333):
    """GeneralizedRCNNTransform.resize: scale=min(min_size/min(h,w), max_size/max(h,w));
    output = floor(in*scale) (``recompute_scale_factor=True``)."""
    scale = min(float(min_size) / float(min(h, w)), float(max_size) / float(max(h, w)))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale)), scale


def transform(images, min_size: int = 800, max_size: int = 1333, size_divisible: int = 32):
    """normalise -> bilinear resize -> zero-pad batch to a multiple of 32.
    images: list of [3,H,W] in [0,1].  Returns (batch[N,3,H',W'], image_sizes)."""
    mean = torch.tensor(IMAGE_MEAN)[:, None, None]
    std = torch.tensor(IMAGE_STD)[:, None, None]
    outs, sizes = [], []
    for img in images:
        x = (img - mean) / std
        h, w = x.shape[-2:]
        _, _, scale = resized_size(h, w, min_size, max_size)
        x = F.interpolate(x[None], scale_factor=scale, mode="bilinear",
                          recompute_scale_factor=True, align_corners=False)[0]
        outs.append(x)
        sizes.append((x.shape[-2], x.shape[-1]))
    hm = max(s[0] for s in sizes)
    wm = max(s[1] for s in sizes)
    hp = int(math.ceil(hm / size_divisible) * size_divisible)
    wp = int(math.ceil(wm / size_divisible) * size_divisible)
    batch = torch.zeros((len(outs), 3, hp, wp))
    for i, x in enumerate(outs):
        batch[i, :, :x.shape[1], :x.shape[2]] = x
    return batch, sizes


# --------------------------------------------------------------------------- a3
def frozen_bn(x, p, k):
    """FrozenBatchNorm2d: y = x*scale + (b - rm*scale), scale = w*rsqrt(rv+eps)."""
    scale = p[k + ".weight"] * (p[k + ".running_var"] + FROZEN_BN_EPS).rsqrt()
    shift = p[k + ".bias"] - p[k + ".running_mean"] * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def resnet50_body(x, p, prefix="backbone.body."):
    """torchvision ResNet-50 v1.5 (stride on the 3x3) with FrozenBN -> C2..C5."""
    x = F.relu(frozen_bn(F.conv2d(x, p[prefix + "conv1.weight"], None, 2, 3), p, prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, (nblk, planes, stride) in enumerate(RESNET50_LAYERS, start=1):
        for bi in range(nblk):
            k = f"{prefix}layer{li}.{bi}."
            s = stride if bi == 0 else 1
            idt = x
            o = F.relu(frozen_bn(F.conv2d(x, p[k + "conv1.weight"]), p, k + "bn1"))
            o = F.relu(frozen_bn(F.conv2d(o, p[k + "conv2.weight"], None, s, 1), p, k + "bn2"))
            o = frozen_bn(F.conv2d(o, p[k + "conv3.weight"]), p, k + "bn3")
            if bi == 0:
                idt = frozen_bn(F.conv2d(x, p[k + "downsample.0.weight"], None, s), p, k + "downsample.1")
            x = F.relu(o + idt)
        feats.append(x)
    return feats


# --------------------------------------------------------------------------- a4
def fpn(feats, p, prefix="backbone.fpn."):
    """FeaturePyramidNetwork + LastLevelMaxPool -> OrderedDict '0','1','2','3','pool'."""
    def inner(i, x):
        return F.conv2d(x, p[f"{prefix}inner_blocks.{i}.weight"], p[f"{prefix}inner_blocks.{i}.bias"])

    def layer(i, x):
        return F.conv2d(x, p[f"{prefix}layer_blocks.{i}.weight"], p[f"{prefix}layer_blocks.{i}.bias"], 1, 1)

    last = inner(3, feats[3])
    outs = [None, None, None, layer(3, last)]
    for i in (2, 1, 0):
        lat = inner(i, feats[i])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        outs[i] = layer(i, last)
    od = OrderedDict((str(i), o) for i, o in enumerate(outs))
    od["pool"] = F.max_pool2d(outs[3], 1, 2, 0)
    return od


# --------------------------------------------------------------------------- a5
def rpn_head(features, p, prefix="rpn.head."):
    """RPNHead shared over levels: 3x3+ReLU, 1x1->A, 1x1->4A."""
    obj, dlt = [], []
    for f in features:
        t = F.relu(F.conv2d(f, p[prefix + "conv.weight"], p[prefix + "conv.bias"], 1, 1))
        obj.append(F.conv2d(t, p[prefix + "cls_logits.weight"], p[prefix + "cls_logits.bias"]))
        dlt.append(F.conv2d(t, p[prefix + "bbox_pred.weight"], p[prefix + "bbox_pred.bias"]))
    return obj, dlt


def base_anchors(size, ratios=ANCHOR_RATIOS):
    r = torch.tensor(ratios, dtype=torch.float32)
    h_r = torch.sqrt(r)
    w_r = 1.0 / h_r
    ws = (w_r[:, None] * torch.tensor([float(size)])[None, :]).view(-1)
    hs = (h_r[:, None] * torch.tensor([float(size)])[None, :]).view(-1)
    return (torch.stack([-ws, -hs, ws, hs], 1) / 2).round()


def grid_anchors(padded_hw, feat_hws, sizes=ANCHOR_SIZES):
    """AnchorGenerator: stride = padded_image_size // feature_size (integer division
    => 4,8,16,32,61 at 800^2); order (y, x, anchor)."""
    out = []
    for (fh, fw), s in zip(feat_hws, sizes):
        sh, sw = padded_hw[0] // fh, padded_hw[1] // fw
        xs = torch.arange(0, fw, dtype=torch.float32) * sw
        ys = torch.arange(0, fh, dtype=torch.float32) * sh
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), 1)
        out.append((shifts.view(-1, 1, 4) + base_anchors(s).view(1, -1, 4)).reshape(-1, 4))
    return out


def decode_boxes(deltas, boxes, weights=(1.0, 1.0, 1.0, 1.0)):
    """BoxCoder.decode_single; deltas[N,4k], boxes[N,4] -> [N,4k]."""
    wx, wy, ww, wh = weights
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + 0.5 * widths
    cy = boxes[:, 1] + 0.5 * heights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = torch.clamp(deltas[:, 2::4] / ww, max=BBOX_XFORM_CLIP)
    dh = torch.clamp(deltas[:, 3::4] / wh, max=BBOX_XFORM_CLIP)
    pcx = dx * widths[:, None] + cx[:, None]
    pcy = dy * heights[:, None] + cy[:, None]
    pw = torch.exp(dw) * widths[:, None]
    ph = torch.exp(dh) * heights[:, None]
    x1 = pcx - 0.5 * pw
    y1 = pcy - 0.5 * ph
    x2 = pcx + 0.5 * pw
    y2 = pcy + 0.5 * ph
    return torch.stack((x1, y1, x2, y2), 2).flatten(1)


def clip_boxes(boxes, hw):
    h, w = hw
    b = boxes.clone()
    b[..., 0::2] = b[..., 0::2].clamp(min=0, max=float(w))
    b[..., 1::2] = b[..., 1::2].clamp(min=0, max=float(h))
    return b


def small_box_keep(boxes, min_size):
    return ((boxes[:, 2] - boxes[:, 0]) >= min_size) & ((boxes[:, 3] - boxes[:, 1]) >= min_size)


def nms(boxes, scores, thr):
    """Greedy NMS: sort by score desc (stable: ties keep the lower index first), keep a
    box, suppress later boxes with IoU > thr (strict); areas without +1."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    supp = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if supp[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
            yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
            xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
            yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
            inter = (xx2 - xx1).clamp(min=0) * (yy2 - yy1).clamp(min=0)
            iou = inter / (area[i] + area[i + 1:] - inter)
            supp[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.int64)]


def batched_nms(boxes, scores, idxs, thr):
    """Coordinate-offset trick: offset = idx*(max_coord+1), then plain NMS."""
    if boxes.numel() == 0:
        return torch.zeros((0,), dtype=torch.int64)
    off = idxs.to(boxes) * (boxes.max() + 1.0)
    return nms(boxes + off[:, None], scores, thr)


def rpn_filter_proposals(objectness, deltas, anchors, image_sizes, pre_nms_top_n=1000,
                         post_nms_top_n=1000, nms_thresh=0.7, min_size=1e-3, return_index=False):
    """RegionProposalNetwork.filter_proposals (eval).  objectness: list per level
    [N,A,H,W]; deltas: list per level [N,4A,H,W]; anchors: list per level [HWA,4]."""
    n = objectness[0].shape[0]
    obj = torch.cat([o.permute(0, 2, 3, 1).reshape(n, -1) for o in objectness], 1)
    dl = torch.cat([d.view(n, -1, 4, d.shape[-2], d.shape[-1]).permute(0, 3, 4, 1, 2).reshape(n, -1, 4)
                    for d in deltas], 1)
    anc = torch.cat(anchors, 0)
    per_level = [a.shape[0] for a in anchors]
    levels = torch.cat([torch.full((c,), i, dtype=torch.int64) for i, c in enumerate(per_level)])
    results, result_scores, result_index = [], [], []
    for b in range(n):
        props = decode_boxes(dl[b], anc)
        idx, off = [], 0
        for c in per_level:
            k = min(pre_nms_top_n, c)
            # ties: lower index first (torch.topk leaves it unspecified)
            top = torch.argsort(obj[b, off:off + c], descending=True, stable=True)[:k]
            idx.append(top + off)
            off += c
        idx = torch.cat(idx)
        sc = torch.sigmoid(obj[b, idx])
        bx = clip_boxes(props[idx], image_sizes[b])
        lv = levels[idx]
        keep = small_box_keep(bx, min_size)
        bx, sc, lv, idx = bx[keep], sc[keep], lv[keep], idx[keep]
        keep = batched_nms(bx, sc, lv, nms_thresh)[:post_nms_top_n]
        results.append(bx[keep])
        result_scores.append(sc[keep])
        result_index.append(idx[keep])
    if return_index:        # tests only: the flat anchor index (level-major, then y, x, anchor) of every kept proposal
        return results, result_scores, result_index
    return results, result_scores


# --------------------------------------------------------------------------- a7
def infer_scales(feat_hws, image_sizes):
    hm = max(s[0] for s in image_sizes)
    out = []
    for fh, _ in feat_hws:
        out.append(2.0 ** float(torch.tensor(float(fh) / float(hm)).log2().round()))
    return out


def map_levels(boxes, k_min=2, k_max=5, canonical_scale=224, canonical_level=4, eps=1e-6):
    """LevelMapper: floor(4 + log2(sqrt(area)/224) + 1e-6) clamped to [2,5], minus 2."""
    s = torch.sqrt((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    lvl = torch.floor(canonical_level + torch.log2(s / canonical_scale) + torch.tensor(eps, dtype=s.dtype))
    return (torch.clamp(lvl, min=k_min, max=k_max).to(torch.int64) - k_min)


def _axis_samples(start, extent, pooled, sr, size):
    """positions, validity, low/high index and high-weight along one axis."""
    binsz = extent / pooled
    g = torch.arange(pooled * sr, dtype=torch.float32)
    pos = start + (g // sr) * binsz + ((g % sr) + 0.5) * binsz / sr
    valid = ~((pos < -1.0) | (pos > size))
    pos = pos.clamp(min=0.0)
    lo = pos.to(torch.int64)
    edge = lo >= size - 1
    lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
    hi = torch.where(edge, lo, lo + 1)
    pos = torch.where(edge, lo.to(torch.float32), pos)
    frac = pos - lo.to(torch.float32)
    return valid, lo, hi, frac


def roi_align(feat, rois, scale, pooled, sampling_ratio=2):
    """torchvision ``roi_align(aligned=False)``.  feat[N,C,H,W]; rois[K,5]
    (batch_idx,x1,y1,x2,y2) -> [K,C,P,P].  Vectorised per ROI."""
    k = rois.shape[0]
    c, h, w = feat.shape[1:]
    sr = sampling_ratio
    out = torch.zeros((k, c, pooled, pooled), dtype=feat.dtype)
    for i in range(k):
        b = int(rois[i, 0])
        x1, y1, x2, y2 = (rois[i, j].to(torch.float32) * scale for j in (1, 2, 3, 4))
        rw = torch.clamp(x2 - x1, min=1.0)
        rh = torch.clamp(y2 - y1, min=1.0)
        vy, yl, yh, ly = _axis_samples(y1, rh, pooled, sr, h)
        vx, xl, xh, lx = _axis_samples(x1, rw, pooled, sr, w)
        f = feat[b]
        hy, hx = 1.0 - ly, 1.0 - lx
        v = (f[:, yl][:, :, xl] * (hy[:, None] * hx[None, :]) + f[:, yl][:, :, xh] * (hy[:, None] * lx[None, :])
             + f[:, yh][:, :, xl] * (ly[:, None] * hx[None, :]) + f[:, yh][:, :, xh] * (ly[:, None] * lx[None, :]))
        v = v * (vy[:, None] & vx[None, :]).to(v.dtype)
        out[i] = v.view(c, pooled, sr, pooled, sr).sum((2, 4)) / float(sr * sr)
    return out


def roi_align_scalar(feat, roi, scale, pooled, sampling_ratio=2):
    """Pure-Python scalar restatement of one ROI (small cases only): cross-checks
    the vectorised version above, following the published kernel line by line."""
    c, h, w = feat.shape[1:]
    b = int(roi[0])
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)  # noqa: E731
    x1, y1, x2, y2 = [f32(float(roi[j])) * f32(scale) for j in range(1, 5)]
    rw = max(x2 - x1, f32(1.0))
    rh = max(y2 - y1, f32(1.0))
    bw, bh = rw / pooled, rh / pooled
    out = torch.zeros((c, pooled, pooled))
    for ph in range(pooled):
        for pw in range(pooled):
            acc = torch.zeros(c)
            for iy in range(sampling_ratio):
                y = y1 + ph * bh + (iy + 0.5) * bh / sampling_ratio
                for ix in range(sampling_ratio):
                    x = x1 + pw * bw + (ix + 0.5) * bw / sampling_ratio
                    yy, xx = float(y), float(x)
                    if yy < -1.0 or yy > h or xx < -1.0 or xx > w:
                        continue
                    yy, xx = max(yy, 0.0), max(xx, 0.0)
                    yl, xl = int(yy), int(xx)
                    if yl >= h - 1:
                        yl = yh = h - 1
                        yy = float(yl)
                    else:
                        yh = yl + 1
                    if xl >= w - 1:
                        xl = xh = w - 1
                        xx = float(xl)
                    else:
                        xh = xl + 1
                    ly, lx = f32(yy) - yl, f32(xx) - xl
                    hy, hx = 1.0 - ly, 1.0 - lx
                    acc += (hy * hx * feat[b, :, yl, xl] + hy * lx * feat[b, :, yl, xh]
                            + ly * hx * feat[b, :, yh, xl] + ly * lx * feat[b, :, yh, xh])
            out[:, ph, pw] = acc / (sampling_ratio * sampling_ratio)
    return out


def multiscale_roi_align(features, boxes, image_sizes, pooled, sampling_ratio=2):
    """MultiScaleRoIAlign(['0','1','2','3'], pooled, 2).  features: list of 4 maps
    [N,C,H,W]; boxes: list (per image) of [k_i,4] -> [sum k_i, C, P, P]."""
    rois = torch.cat([torch.cat([torch.full((b.shape[0], 1), float(i)), b.to(torch.float32)], 1)
                      for i, b in enumerate(boxes)], 0)
    scales = infer_scales([f.shape[-2:] for f in features], image_sizes)
    k_min = int(-math.log2(scales[0]))
    k_max = int(-math.log2(scales[-1]))
    lv = map_levels(rois[:, 1:], k_min, k_max)
    out = torch.zeros((rois.shape[0], features[0].shape[1], pooled, pooled))
    for l, (f, s) in enumerate(zip(features, scales)):
        sel = torch.nonzero(lv == l).squeeze(1)
        if sel.numel():
            out[sel] = roi_align(f, rois[sel], s, pooled, sampling_ratio)
    return out


# --------------------------------------------------------------------------- a6
def box_head(x, p, prefix="roi_heads."):
    """TwoMLPHead + FastRCNNPredictor: [K,256,7,7] -> (logits[K,nc], deltas[K,4nc])."""
    x = x.flatten(1)
    x = F.relu(F.linear(x, p[prefix + "box_head.fc6.weight"], p[prefix + "box_head.fc6.bias"]))
    x = F.relu(F.linear(x, p[prefix + "box_head.fc7.weight"], p[prefix + "box_head.fc7.bias"]))
    return (F.linear(x, p[prefix + "box_predictor.cls_score.weight"], p[prefix + "box_predictor.cls_score.bias"]),
            F.linear(x, p[prefix + "box_predictor.bbox_pred.weight"], p[prefix + "box_predictor.bbox_pred.bias"]))


def postprocess_detections(class_logits, box_regression, proposals, image_shapes,
                           score_thresh=0.05, nms_thresh=0.5, detections_per_img=100):
    """ref models/video_matchrcnn.py:154-205 (in-tree).  box_coder weights (10,10,5,5)."""
    num_classes = class_logits.shape[-1]
    counts = [len(b) for b in proposals]
    pred_boxes = decode_boxes(box_regression, torch.cat(proposals, 0), (10.0, 10.0, 5.0, 5.0))
    pred_boxes = pred_boxes.view(-1, num_classes, 4)
    pred_scores = F.softmax(class_logits, -1)
    all_b, all_s, all_l = [], [], []
    for boxes, scores, shape in zip(pred_boxes.split(counts, 0), pred_scores.split(counts, 0), image_shapes):
        boxes = clip_boxes(boxes, shape)
        labels = torch.arange(num_classes).view(1, -1).expand_as(scores)
        boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].flatten(), labels[:, 1:].flatten()
        inds = torch.nonzero(scores > score_thresh).squeeze(1)
        boxes, scores, labels = boxes[inds], scores[inds], labels[inds]
        keep = small_box_keep(boxes, 1e-2)
        boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
        keep = batched_nms(boxes, scores, labels, nms_thresh)[:detections_per_img]
        all_b.append(boxes[keep])
        all_s.append(scores[keep])
        all_l.append(labels[keep])
    return all_b, all_s, all_l


# --------------------------------------------------------------------------- a8
def mask_head(x, p, prefix="roi_heads."):
    """MaskRCNNHeads (4x conv3x3 p1 + ReLU) + MaskRCNNPredictor (ConvT 2x2 s2 + ReLU,
    1x1 -> num_classes): [K,256,14,14] -> logits [K,nc,28,28]."""
    for i in range(1, 5):
        k = f"{prefix}mask_head.mask_fcn{i}"
        x = F.relu(F.conv2d(x, p[k + ".weight"], p[k + ".bias"], 1, 1))
    k = prefix + "mask_predictor."
    x = F.relu(F.conv_transpose2d(x, p[k + "conv5_mask.weight"], p[k + "conv5_mask.bias"], 2))
    return F.conv2d(x, p[k + "mask_fcn_logits.weight"], p[k + "mask_fcn_logits.bias"])


def maskrcnn_inference(mask_logits, labels):
    """sigmoid, pick channel = label per ROI -> list of [k_i,1,28,28]."""
    prob = mask_logits.sigmoid()
    counts = [len(l) for l in labels]
    lab = torch.cat(labels)
    prob = prob[torch.arange(prob.shape[0]), lab][:, None]
    return list(prob.split(counts, 0))


def paste_masks_in_image(masks, boxes, img_hw, padding=1):
    """transform.postprocess mask paste: pad 1px, expand box, int box, bilinear resize
    of the prob map to the integer box size, paste; no threshold. -> [K,1,H,W]."""
    m = masks.shape[-1]
    scale = float(m + 2 * padding) / m
    pm = F.pad(masks, (padding,) * 4)
    wh = (boxes[:, 2] - boxes[:, 0]) * 0.5 * scale
    hh = (boxes[:, 3] - boxes[:, 1]) * 0.5 * scale
    xc = (boxes[:, 2] + boxes[:, 0]) * 0.5
    yc = (boxes[:, 3] + boxes[:, 1]) * 0.5
    eb = torch.stack((xc - wh, yc - hh, xc + wh, yc + hh), 1).to(torch.int64)
    im_h, im_w = img_hw
    res = []
    for mk, b in zip(pm, eb):
        b = b.tolist()
        w = max(b[2] - b[0] + 1, 1)
        h = max(b[3] - b[1] + 1, 1)
        r = F.interpolate(mk[None], size=(h, w), mode="bilinear", align_corners=False)[0, 0]
        im = torch.zeros((im_h, im_w), dtype=r.dtype)
        x0, x1 = max(b[0], 0), min(b[2] + 1, im_w)
        y0, y1 = max(b[1], 0), min(b[3] + 1, im_h)
        if x1 > x0 and y1 > y0:
            im[y0:y1, x0:x1] = r[(y0 - b[1]):(y1 - b[1]), (x0 - b[0]):(x1 - b[0])]
        res.append(im[None])
    if not res:
        return masks.new_zeros((0, 1, im_h, im_w))
    return torch.stack(res, 0)


def rescale_boxes(boxes, from_hw, to_hw):
    rh = float(to_hw[0]) / float(from_hw[0])
    rw = float(to_hw[1]) / float(from_hw[1])
    b = boxes.clone()
    b[:, 0::2] = boxes[:, 0::2] * rw
    b[:, 1::2] = boxes[:, 1::2] * rh
    return b
