"""Deterministic synthetic weights and inputs (host logic; no torch RNG).

Everything here comes from a repo-owned counter-based integer generator
(splitmix64 finaliser over ``(stream, index)``), so every host produces the
same bits regardless of the numpy/torch version.  Normal variates are an
Irwin-Hall sum of 12 uniforms, computed in exact integer arithmetic before a
single conversion to float (no libm calls).

Shapes / distributions follow SURVEY.md section 8(d):
  * frames           U[0,1)  ``[T,3,H,W]``            (what ``ToTensor`` yields,
                                                       ref stuffs/transform.py:46-49)
  * fixed ROI sets   8 / 32 / 64 boxes covering all 4 FPN levels
  * head weights     Kaiming-uniform scale; ``newnlb.W`` non-zero (it is
                     zero-initialised in the reference, models/nlb.py:48-49, so
                     a zero W would make the non-local block an identity)
  * BatchNorm        running_mean ~ N(0,0.1^2), running_var ~ U[0.5,1.5]
State-dict key names are the reference's own.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def stream_id(seed: int, name: str = "") -> int:
    """Stable 64-bit stream id from an integer seed and a tensor name."""
    h = zlib.crc32(name.encode("utf-8")) & 0xFFFFFFFF
    return ((int(seed) & 0xFFFFFFFF) << 32) | h


def raw_u64(stream: int, n: int, offset: int = 0) -> np.ndarray:
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _mix64(np.full(1, np.uint64(stream & 0xFFFFFFFFFFFFFFFF)) * _GOLD + np.uint64(0x632BE59BD9B4E019))
        return _mix64((idx + np.uint64(1)) * _GOLD ^ base)


def uniform(stream: int, shape, lo: float = 0.0, hi: float = 1.0, offset: int = 0) -> np.ndarray:
    """float32 U[lo,hi) with 24-bit resolution."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    out = np.empty(n, dtype=np.float32)
    step = 1 << 22
    for s in range(0, n, step):
        e = min(n, s + step)
        u24 = (raw_u64(stream, e - s, offset + s) >> np.uint64(40)).astype(np.float64)
        out[s:e] = (lo + (hi - lo) * (u24 / 16777216.0)).astype(np.float32)
    return out.reshape(shape)


def normal(stream: int, shape, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """float32 approx-N(mean,std^2): Irwin-Hall(12) - 6, exact integer sum."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    out = np.empty(n, dtype=np.float32)
    step = 1 << 20
    for s in range(0, n, step):
        e = min(n, s + step)
        acc = np.zeros(e - s, dtype=np.int64)
        for j in range(12):
            acc += (raw_u64(stream ^ (0x5851F42D4C957F2D * (j + 1) & 0xFFFFFFFFFFFFFFFF), e - s, s)
                    >> np.uint64(40)).astype(np.int64)
        out[s:e] = (mean + std * (acc.astype(np.float64) / 16777216.0 - 6.0)).astype(np.float32)
    return out.reshape(shape)


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------

def _kaiming_uniform(seed, name, shape, fan_in, gain=math.sqrt(2.0)):
    b = gain * math.sqrt(3.0 / fan_in)
    return uniform(stream_id(seed, name), shape, -b, b)


def _bias(seed, name, n, fan_in):
    b = 1.0 / math.sqrt(fan_in)
    return uniform(stream_id(seed, name), (n,), -b, b)


def match_predictor_state(seed: int, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """The 17 tensors of ``MatchPredictor`` (ref models/match_head.py:47-64)."""
    sd = OrderedDict()
    chans = [(256, 256), (256, 256), (256, 256), (256, 1024)]
    for i, (ci, co) in zip((0, 2, 4, 6), chans):
        k = f"{prefix}conv_seq.{i}"
        sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (co, ci, 3, 3), ci * 9)
        sd[k + ".bias"] = _bias(seed, k + ".bias", co, ci * 9)
    sd[prefix + "linear.0.weight"] = _kaiming_uniform(seed, prefix + "linear.0.weight", (256, 1024), 1024, 1.0)
    sd[prefix + "linear.0.bias"] = _bias(seed, prefix + "linear.0.bias", 256, 1024)
    sd[prefix + "linear.1.weight"] = uniform(stream_id(seed, prefix + "linear.1.weight"), (256,), 0.5, 1.5)
    sd[prefix + "linear.1.bias"] = normal(stream_id(seed, prefix + "linear.1.bias"), (256,), 0.0, 0.1)
    sd[prefix + "linear.1.running_mean"] = normal(stream_id(seed, prefix + "linear.1.running_mean"), (256,), 0.0, 0.1)
    sd[prefix + "linear.1.running_var"] = uniform(stream_id(seed, prefix + "linear.1.running_var"), (256,), 0.5, 1.5)
    sd[prefix + "linear.1.num_batches_tracked"] = np.zeros((), dtype=np.int64)
    sd[prefix + "last.weight"] = _kaiming_uniform(seed, prefix + "last.weight", (2, 256), 256, 1.0)
    sd[prefix + "last.bias"] = _bias(seed, prefix + "last.bias", 2, 256)
    return sd


def temporal_aggregator_state(seed: int, prefix: str = "") -> "OrderedDict[str, np.ndarray]":
    """``TemporalAggregationNLB`` = MatchPredictor + scorer + NLB
    (ref models/match_head.py:79-88, models/nlb.py:34-59)."""
    sd = match_predictor_state(seed, prefix)
    sd[prefix + "attention_scorer.weight"] = _kaiming_uniform(seed, prefix + "attention_scorer.weight", (1, 256), 256, 1.0)
    sd[prefix + "attention_scorer.bias"] = _bias(seed, prefix + "attention_scorer.bias", 1, 256)
    for nm in ("g", "theta", "phi"):
        k = f"{prefix}newnlb.{nm}"
        sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (128, 256, 1), 256, 1.0)
        sd[k + ".bias"] = _bias(seed, k + ".bias", 128, 256)
    # zero-initialised upstream (models/nlb.py:48-49): use non-zero values
    sd[prefix + "newnlb.W.weight"] = normal(stream_id(seed, prefix + "newnlb.W.weight"), (256, 128, 1), 0.0, 0.05)
    sd[prefix + "newnlb.W.bias"] = normal(stream_id(seed, prefix + "newnlb.W.bias"), (256,), 0.0, 0.05)
    sd[prefix + "newnlb.concat_project.0.weight"] = _kaiming_uniform(
        seed, prefix + "newnlb.concat_project.0.weight", (1, 256, 1, 1), 256, 1.0)
    return sd


def _frozen_bn(sd, seed, k, c, wlo=0.5, whi=1.5):
    sd[k + ".weight"] = uniform(stream_id(seed, k + ".weight"), (c,), wlo, whi)
    sd[k + ".bias"] = normal(stream_id(seed, k + ".bias"), (c,), 0.0, 0.05)
    sd[k + ".running_mean"] = normal(stream_id(seed, k + ".running_mean"), (c,), 0.0, 0.1)
    sd[k + ".running_var"] = uniform(stream_id(seed, k + ".running_var"), (c,), 0.5, 1.5)


RESNET50_LAYERS = ((3, 64, 1), (4, 128, 2), (6, 256, 2), (3, 512, 2))


def detector_state(seed: int, num_classes: int = 14) -> "OrderedDict[str, np.ndarray]":
    """torchvision-classic key layout of the MaskRCNN-R50-FPN trunk
    (SURVEY.md Appendix C): backbone.body/fpn, rpn.head, roi_heads.box_*, mask_*."""
    sd = OrderedDict()
    b = "backbone.body."
    sd[b + "conv1.weight"] = _kaiming_uniform(seed, b + "conv1.weight", (64, 3, 7, 7), 3 * 49)
    _frozen_bn(sd, seed, b + "bn1", 64)
    inpl = 64
    for li, (nblk, planes, stride) in enumerate(RESNET50_LAYERS, start=1):
        for bi in range(nblk):
            p = f"{b}layer{li}.{bi}."
            sd[p + "conv1.weight"] = _kaiming_uniform(seed, p + "conv1.weight", (planes, inpl, 1, 1), inpl)
            _frozen_bn(sd, seed, p + "bn1", planes)
            sd[p + "conv2.weight"] = _kaiming_uniform(seed, p + "conv2.weight", (planes, planes, 3, 3), planes * 9)
            _frozen_bn(sd, seed, p + "bn2", planes)
            sd[p + "conv3.weight"] = _kaiming_uniform(seed, p + "conv3.weight", (planes * 4, planes, 1, 1), planes)
            _frozen_bn(sd, seed, p + "bn3", planes * 4, 0.15, 0.35)   # keeps the residual sum bounded
            if bi == 0:
                sd[p + "downsample.0.weight"] = _kaiming_uniform(
                    seed, p + "downsample.0.weight", (planes * 4, inpl, 1, 1), inpl, 1.0)
                _frozen_bn(sd, seed, p + "downsample.1", planes * 4, 0.5, 1.0)
            inpl = planes * 4
    f = "backbone.fpn."
    for i, c in enumerate((256, 512, 1024, 2048)):
        k = f"{f}inner_blocks.{i}"
        sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (256, c, 1, 1), c, 1.0)
        sd[k + ".bias"] = _bias(seed, k + ".bias", 256, c)
        k = f"{f}layer_blocks.{i}"
        sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (256, 256, 3, 3), 2304, 1.0)
        sd[k + ".bias"] = _bias(seed, k + ".bias", 256, 2304)
    r = "rpn.head."
    sd[r + "conv.weight"] = _kaiming_uniform(seed, r + "conv.weight", (256, 256, 3, 3), 2304)
    sd[r + "conv.bias"] = _bias(seed, r + "conv.bias", 256, 2304)
    sd[r + "cls_logits.weight"] = _kaiming_uniform(seed, r + "cls_logits.weight", (3, 256, 1, 1), 256, 1.0)
    sd[r + "cls_logits.bias"] = _bias(seed, r + "cls_logits.bias", 3, 256)
    sd[r + "bbox_pred.weight"] = _kaiming_uniform(seed, r + "bbox_pred.weight", (12, 256, 1, 1), 256, 0.3)
    sd[r + "bbox_pred.bias"] = _bias(seed, r + "bbox_pred.bias", 12, 256)
    h = "roi_heads."
    sd[h + "box_head.fc6.weight"] = _kaiming_uniform(seed, h + "box_head.fc6.weight", (1024, 12544), 12544)
    sd[h + "box_head.fc6.bias"] = _bias(seed, h + "box_head.fc6.bias", 1024, 12544)
    sd[h + "box_head.fc7.weight"] = _kaiming_uniform(seed, h + "box_head.fc7.weight", (1024, 1024), 1024)
    sd[h + "box_head.fc7.bias"] = _bias(seed, h + "box_head.fc7.bias", 1024, 1024)
    sd[h + "box_predictor.cls_score.weight"] = _kaiming_uniform(
        seed, h + "box_predictor.cls_score.weight", (num_classes, 1024), 1024, 1.0)
    sd[h + "box_predictor.cls_score.bias"] = _bias(seed, h + "box_predictor.cls_score.bias", num_classes, 1024)
    sd[h + "box_predictor.bbox_pred.weight"] = _kaiming_uniform(
        seed, h + "box_predictor.bbox_pred.weight", (4 * num_classes, 1024), 1024, 0.3)
    sd[h + "box_predictor.bbox_pred.bias"] = _bias(seed, h + "box_predictor.bbox_pred.bias", 4 * num_classes, 1024)
    for i in range(1, 5):
        k = f"{h}mask_head.mask_fcn{i}"
        sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (256, 256, 3, 3), 2304)
        sd[k + ".bias"] = _bias(seed, k + ".bias", 256, 2304)
    k = h + "mask_predictor.conv5_mask"
    sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (256, 256, 2, 2), 1024)   # ConvTranspose2d [Cin,Cout,2,2]
    sd[k + ".bias"] = _bias(seed, k + ".bias", 256, 1024)
    k = h + "mask_predictor.mask_fcn_logits"
    sd[k + ".weight"] = _kaiming_uniform(seed, k + ".weight", (num_classes, 256, 1, 1), 256, 1.0)
    sd[k + ".bias"] = _bias(seed, k + ".bias", num_classes, 256)
    return sd


def video_matchrcnn_state(seed: int, num_classes: int = 14) -> "OrderedDict[str, np.ndarray]":
    """Full ``VideoMatchRCNN`` state dict (ref models/video_matchrcnn.py:320-343)."""
    sd = detector_state(seed, num_classes)
    sd.update(match_predictor_state(seed + 1, "roi_heads.match_predictor."))
    sd.update(temporal_aggregator_state(seed + 2, "roi_heads.temporal_aggregator."))
    return sd


# ----------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------

C1_ROIS = np.array([
    [100, 100, 180, 200], [300, 50, 360, 110], [50, 400, 200, 560], [400, 400, 600, 560],
    [100, 100, 400, 420], [420, 60, 760, 380], [20, 20, 780, 780], [0, 0, 800, 800]], dtype=np.float32)


def fixed_rois(n: int, height: int = 800, width: int = 800) -> np.ndarray:
    """SURVEY.md 8(d) fixed ROI sets (xyxy px).  n=8: the C1 list; n=32 / 64:
    ROI r centred on an 8x4 (8x8) grid, side in {64,128,256,512}[r%4], aspect
    in {1,1/2,2}[r%3], clipped to the image."""
    if n == 8:
        r = C1_ROIS.copy()
        r[:, [0, 2]] *= width / 800.0
        r[:, [1, 3]] *= height / 800.0
        return r
    gx = 8
    gy = n // gx
    out = np.zeros((n, 4), dtype=np.float32)
    sides = (64.0, 128.0, 256.0, 512.0)
    aspects = (1.0, 0.5, 2.0)
    for r in range(n):
        cx = (r % gx + 0.5) * width / gx
        cy = (r // gx + 0.5) * height / gy
        s = sides[r % 4]
        a = aspects[r % 3]                 # a = w/h
        w = s * math.sqrt(a)
        h = s / math.sqrt(a)
        out[r] = (max(cx - w / 2, 0.0), max(cy - h / 2, 0.0), min(cx + w / 2, width), min(cy + h / 2, height))
    return out


def frames(clip: int, t: int, height: int = 800, width: int = 800) -> np.ndarray:
    """``[t,3,H,W]`` float32 U[0,1); seed = 1000 + clip."""
    return uniform(stream_id(1000 + clip, "frames"), (t, 3, height, width))


def roi_features(seed: int, k: int) -> np.ndarray:
    """Head-only configs start from ``[k,256,14,14]`` ~ |N(0,1)| (post-ReLU-like)."""
    return np.abs(normal(stream_id(seed, "roi_features"), (k, 256, 14, 14)))


def gallery(seed: int, g: int) -> np.ndarray:
    """Product-descriptor bank ``x3_2[g,256]`` ~ N(0,1) (BatchNorm-output scale)."""
    return normal(stream_id(seed, "gallery"), (g, 256))
