"""The torchvision-owned stages of the path, rebuilt on the gfx950 kernels.

The reference contains no arithmetic for these (it configures ``torchvision`` objects:
ref models/video_matchrcnn.py:6-9,337-338; models/matchrcnn.py:2-3,11-28); this module
supplies them with torchvision's classic (<=0.12) module/attribute/state-dict names
 so reference checkpoints load unchanged:

  GeneralizedRCNNTransform   normalise / resize / pad / batch            (row a2)
  ResNet50Body + FPN         ``backbone.body.*``, ``backbone.fpn.*``         (rows a3, a4)
  RegionProposalNetwork      ``rpn.head.*`` + anchors + filter_proposals   (row a5)
  MultiScaleRoIAlign         FPN level mapper + RoIAlign                 (row a7)
  TwoMLPHead / FastRCNNPredictor / MaskRCNNHeads / MaskRCNNPredictor     (rows a6, a8)

Activations are NHWC fp32 device tensors internally.  Modules hold parameters only
(``nn.Conv2d`` etc. as containers); every ``forward`` launches HIP kernels through
``ops`` -- CPU tensors raise, there is no eager fallback.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops

RESNET50_LAYERS = ((3, 64, 1), (4, 128, 2), (6, 256, 2), (3, 512, 2))
BBOX_XFORM_CLIP = math.log(1000.0 / 16)
NMS_MAX_BOXES = 16384     # per-image capacity of seam_nms_sorted_f32 (csrc/seam_detect.hip)


def _key(params, dtype=None) -> tuple:
    return tuple((p.data_ptr(), p._version) for p in params) + (dtype,)


def cdt(module):
    """Operand precision of a module's contractions: torch.float32 (exact fp32 MFMA, default), torch.float16 (the
    fp16-MFMA / fp32-accumulate path of BASELINE config 5) or ops.BX3 = "bf16x3" (fp32 activations, split-bf16 operands,
    3 bf16 MFMAs per product, fp32 accumulate: ~1e-5 relative).  Set for a whole model with ``set_compute_dtype``."""
    return getattr(module, "compute_dtype", torch.float32)


def adt(module) -> torch.dtype:
    """Storage type of a module's activations: fp16 only on the fp16 path."""
    return torch.float16 if cdt(module) == torch.float16 else torch.float32


def set_compute_dtype(model: nn.Module, dtype) -> nn.Module:
    if dtype not in (torch.float32, torch.float16, ops.BX3):
        raise ValueError('compute dtype must be torch.float32, torch.float16 or "bf16x3"')
    for m in model.modules():
        m.compute_dtype = dtype
    return model


# ------------------------------------------------------------------------------ transform (a2)
def resized_size(h: int, w: int, min_size: int = 800, max_size: int = 1333) -> Tuple[int, int, float]:
    """scale = min(min_size/min(h,w), max_size/max(h,w)); out = floor(in*scale)
    (torchvision ``_resize_image_and_masks`` with ``recompute_scale_factor=True``)."""
    scale = min(float(min_size) / float(min(h, w)), float(max_size) / float(max(h, w)))
    return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale)), scale


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size=800, max_size=1333, size_divisible=32):
        super().__init__()
        self.min_size, self.max_size, self.size_divisible = min_size, max_size, size_divisible

    def forward(self, images: Sequence[torch.Tensor]):
        """list of [3,H,W] in [0,1] -> (NHWC batch [N,Hp,Wp,4] fp32 | [N,Hp,Wp,8] fp16, image_sizes,
        original_sizes)."""
        # float [3,H,W] in [0,1] (the reference's input) or, as an extension, raw uint8 [H,W,3] frames
        orig = [(int(i.shape[0]), int(i.shape[1])) if i.dtype == torch.uint8 else (int(i.shape[-2]), int(i.shape[-1]))
                for i in images]
        sizes = [resized_size(h, w, self.min_size, self.max_size)[:2] for h, w in orig]
        d = self.size_divisible
        hp = int(math.ceil(max(s[0] for s in sizes) / d) * d)
        wp = int(math.ceil(max(s[1] for s in sizes) / d) * d)
        # fp32 / fp16 paths with float images: space-to-depth layout for the 4x4 / stride-1 form of the stem (STEM_S2D)
        s2d = STEM_S2D and cdt(self) in (torch.float32, torch.float16) and all(i.dtype == torch.float32 for i in images)
        # fp16 path: the frame comes with the zero cells the streaming stem kernel wants around it (2 before, 1 after); the caller
        # tells the backbone (``s2d_padded``) by comparing the frame's height with the padded size returned here
        pad = (2, 1) if (s2d and STEM_SWH and adt(self) == torch.float16) else (0, 0)
        return ops.preprocess(images, sizes, hp, wp, adt(self), s2d=s2d, s2d_pad=pad), sizes, orig, (hp, wp)

    @staticmethod
    def rescale_boxes(boxes: torch.Tensor, from_hw, to_hw) -> torch.Tensor:
        rh, rw = float(to_hw[0]) / float(from_hw[0]), float(to_hw[1]) / float(from_hw[1])
        return boxes * boxes.new_tensor([rw, rh, rw, rh])


# ------------------------------------------------------------------------------ backbone (a3)
class FrozenBatchNorm2d(nn.Module):
    """Buffers only (weight, bias, running_mean, running_var); folded into the conv epilogue."""

    def __init__(self, n, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def tensors(self):
        return (self.weight, self.bias, self.running_mean, self.running_var)


# Projection-shortcut blocks (first block of each ResNet layer) as one dual-source GEMM (ops.conv2d_dual).  The two
# FrozenBN scales are folded into the weights, so results differ from the two-launch form by ~1e-7 relative (one extra
# rounding per weight); SEAM_FUSE_SHORTCUT=0 keeps the two launches.
import os as _os
FUSE_SHORTCUT = _os.environ.get("SEAM_FUSE_SHORTCUT", "1") != "0"
# The stem on a space-to-depth input: 7x7 / stride 2 / pad 3 over 3 colours == 4x4 / stride 1 / pad (2 before, 1 after) over the 12
# channels (dy, dx, c) -- the same 147 products per output in 192 reduction steps instead of 224 (SEAM_STEM_S2D=0: NHWC4 form).
STEM_S2D = _os.environ.get("SEAM_STEM_S2D", "1") != "0"
# fp16 path: the stem on the streaming kernel over the zero-padded space-to-depth frame (csrc/seam_pwh.hip stem_swh_kernel, round 6);
# SEAM_STEM_SWH=0 keeps the implicit GEMM (seam_conv2d_crop_f16)
STEM_SWH = _os.environ.get("SEAM_STEM_SWH", "1") != "0"
# ResNet body: batch slices on this many HIP streams (bit-identical results, tested).  2 since round 2: the tail of one slice's launch
# (3.05 rounds of tiles cost 4) runs under the other slice's next layer: 119.8 vs 121.5 ms per 8-clip step, 16.6 vs 17.0 ms at one
# clip per step, one box (profiles/r02_body_streams.txt); 3 streams give less, 1 restores the single-stream walk.
BODY_STREAMS = int(_os.environ.get("SEAM_BODY_STREAMS", "2"))


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)     # v1.5: stride on the 3x3
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            FrozenBatchNorm2d(planes * 4))
        self.stride = stride


class ResNet50Body(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        inpl = 64
        for li, (nblk, planes, stride) in enumerate(RESNET50_LAYERS, start=1):
            blocks = []
            for bi in range(nblk):
                blocks.append(Bottleneck(inpl, planes, stride if bi == 0 else 1, bi == 0))
                inpl = planes * 4
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self._pk, self._pk_key = None, None

    def _all(self):
        return list(self.parameters()) + list(self.buffers())

    def packed(self):
        dt = cdt(self)
        key = _key(self._all(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                pk = {"stem": ops.pack_conv(self.conv1.weight, None, self.bn1.tensors(), stride=2, pad=3,
                                            cstore=8 if dt == torch.float16 else 4, bn_eps=self.bn1.eps, dtype=dt)}
                if dt in (torch.float32, torch.float16):
                    # w'[k, (dy*2+dx)*3 + c, r', s'] = w[k, c, 2r'+dy-1, 2s'+dx-1]  (zero where the 7x7 kernel has no tap)
                    w8 = F.pad(self.conv1.weight.detach().to(torch.float32), (1, 0, 1, 0))          # [64,3,8,8], index 0 = tap -1
                    ws = w8.view(64, 3, 4, 2, 4, 2).permute(0, 3, 5, 1, 2, 4).reshape(64, 12, 4, 4).contiguous()
                    pk["stem_s2d"] = ops.pack_conv(ws, None, self.bn1.tensors(), stride=1, pad=2, cstore=12 if dt == torch.float32 else 16,
                                                   bn_eps=self.bn1.eps, wino=False, dtype=dt)
                    if dt == torch.float16 and STEM_SWH:
                        # the streaming form of the fp16 stem (ops.stem_s2d_f16): rows [64, (r, s, 16 channels)] of the same weights
                        pk["stem_rows"] = F.pad(ws, (0, 0, 0, 0, 0, 4)).permute(0, 2, 3, 1).reshape(64, 256).to(torch.float16).contiguous()
                for li in range(1, 5):
                    for bi, b in enumerate(getattr(self, f"layer{li}")):
                        e = {"c1": ops.pack_conv(b.conv1.weight, None, b.bn1.tensors(), bn_eps=b.bn1.eps, dtype=dt),
                             "c2": ops.pack_conv(b.conv2.weight, None, b.bn2.tensors(), stride=b.stride, pad=1,
                                                 bn_eps=b.bn2.eps, dtype=dt),
                             "c3": ops.pack_conv(b.conv3.weight, None, b.bn3.tensors(), bn_eps=b.bn3.eps, dtype=dt)}
                        if b.downsample is not None:
                            e["ds"] = ops.pack_conv(b.downsample[0].weight, None, b.downsample[1].tensors(),
                                                    stride=b.stride, bn_eps=b.downsample[1].eps, dtype=dt)
                            if dt in (torch.float32, torch.float16) and FUSE_SHORTCUT:     # conv3 + projection shortcut as one GEMM
                                e["c3ds"] = ops.pack_conv_dual(b.conv3.weight, b.bn3.tensors(), b.downsample[0].weight,
                                                               b.downsample[1].tensors(), bn_eps=b.bn3.eps, dtype=dt)
                        pk[(li, bi)] = e
            self._pk, self._pk_key = pk, key
        return self._pk

    def forward(self, x: torch.Tensor, s2d_padded: bool = False) -> List[torch.Tensor]:
        """x NHWC4 [N,H,W,4] (or the space-to-depth frame [N,H/2,W/2,12|16]; ``s2d_padded``: the fp16 frame with its 2 + 1 zero
        cells around it, [N,H/2+3,W/2+3,16]) -> [C2, C3, C4, C5] NHWC."""
        pk = self.packed()
        n = x.shape[0]
        if BODY_STREAMS >= 2 and n >= 2 * BODY_STREAMS and x.is_cuda:
            # The batch in BODY_STREAMS slices on as many HIP streams, each writing its slice of the stage outputs: while one
            # slice sits in an HBM-bound layer (the 1x1 expansions with their residual) another one runs an MFMA-bound layer on
            # the same CUs.  Per-image results do not depend on the batch an image rides in (tested), so this is the same math.
            cur = torch.cuda.current_stream()
            if (getattr(self, "_streams", None) is None or len(self._streams) != BODY_STREAMS
                    or self._streams[0].device != x.device):            # rebuilt after model.to(another device)
                self._streams = [torch.cuda.Stream(device=x.device) for _ in range(BODY_STREAMS)]
            spad = 3 if s2d_padded else 0      # padded s2d frame (fp16 path)
            h1, w1 = ((x.shape[1] - spad, x.shape[2] - spad) if x.shape[-1] in (12, 16) else
                      ((x.shape[1] + 6 - 7) // 2 + 1, (x.shape[2] + 6 - 7) // 2 + 1))
            hh, ww = (h1 + 2 - 3) // 2 + 1, (w1 + 2 - 3) // 2 + 1
            outs = []
            for li, (nblk, planes, stride) in enumerate(RESNET50_LAYERS, start=1):
                if stride == 2:
                    hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
                outs.append(torch.empty((n, hh, ww, planes * 4), dtype=x.dtype, device=x.device))
            bounds = [n * i // BODY_STREAMS for i in range(BODY_STREAMS + 1)]
            for i, st in enumerate(self._streams):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    lo, hi = bounds[i], bounds[i + 1]
                    self._run(x[lo:hi], pk, [o[lo:hi] for o in outs], s2d_padded)
            for st in self._streams:
                cur.wait_stream(st)
            return outs
        return self._run(x, pk, None, s2d_padded)

    def _run(self, x, pk, outs, s2d_padded=False):
        if x.shape[-1] in (12, 16):    # space-to-depth input [N,H/2,W/2,12] (fp16: 16): the stem as a 4x4 / stride-1 conv, output grid = input grid
            streaming = ("stem_rows" in pk and x.dtype == torch.float16 and x.shape[-1] == 16
                         and (x.shape[1] - (3 if s2d_padded else 0)) * (x.shape[2] - (3 if s2d_padded else 0)) >= 128)
            if s2d_padded and not streaming:      # (cannot happen through the transform: it pads only when the kernel will run)
                x = x[:, 2:-1, 2:-1].contiguous()
            if streaming:
                x = ops.stem_s2d_f16(x, pk["stem_rows"], pk["stem_s2d"].scale, pk["stem_s2d"].shift, relu=True, padded=s2d_padded)
            else:
                x = ops.conv2d(x, pk["stem_s2d"], relu=True, out_hw=(x.shape[1], x.shape[2]))
        else:
            x = ops.conv2d(x, pk["stem"], relu=True)           # 7x7/s2 + FrozenBN + ReLU
        x = ops.maxpool2d(x, 3, 2, 1)
        feats = []
        for li in range(1, 5):
            nblk = len(getattr(self, f"layer{li}"))
            for bi in range(nblk):
                e = pk[(li, bi)]
                o = ops.conv2d(x, e["c1"], relu=True)
                o = ops.conv2d(o, e["c2"], relu=True)
                if "c3ds" in e:       # projection-shortcut block: bn3(conv3(o)) + bn_d(conv_d(x)) + ReLU in one launch
                    x = ops.conv2d_dual(o, x, e["c3ds"], getattr(self, f"layer{li}")[bi].stride, relu=True)
                    continue
                idt = ops.conv2d(x, e["ds"]) if "ds" in e else x
                x = ops.conv2d(o, e["c3"], relu=True, residual=idt,   # bn3 + add + ReLU fused
                               out=outs[li - 1] if (outs is not None and bi == nblk - 1) else None)
            feats.append(x)
        return feats


# The small pyramid levels (100^2 and below) cannot fill 256 CUs on their own; their FPN output convs and RPN-head launches run on
# a side stream next to the 200^2 level's launches, whose blocks take the CUs they leave idle (SEAM_LEVEL_STREAMS=0: one stream).
# Outputs are allocated on the calling stream and written through `out=`; tensors produced on one stream and read on the other are
# kept referenced until the join (fork / join lifetimes, see FPN.forward).  Same kernels on the same data: results are bit-identical.
LEVEL_STREAMS = _os.environ.get("SEAM_LEVEL_STREAMS", "1") != "0"
_SIDE_STREAMS = {}


def _side_stream(dev):
    st = _SIDE_STREAMS.get(dev)
    if st is None:
        st = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return st


class FeaturePyramidNetwork(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256):
        super().__init__()
        self.inner_blocks = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for c in in_channels])
        self.layer_blocks = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels])
        self._pk, self._pk_key = None, None

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        # torchvision >= 0.13 wraps the convs in Conv2dNormActivation: ``inner_blocks.{i}.0.weight``
        for key in [k_ for k_ in state_dict if k_.startswith(prefix)]:
            parts = key[len(prefix):].split(".")
            if len(parts) == 4 and parts[0] in ("inner_blocks", "layer_blocks") and parts[2] == "0":
                state_dict[prefix + ".".join((parts[0], parts[1], parts[3]))] = state_dict.pop(key)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def packed(self):
        dt = cdt(self)
        key = _key(self.parameters(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                self._pk = ([ops.pack_conv(m.weight, m.bias, dtype=dt) for m in self.inner_blocks],
                            [ops.pack_conv(m.weight, m.bias, pad=1, dtype=dt) for m in self.layer_blocks])
            self._pk_key = key
        return self._pk

    def forward(self, feats: List[torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
        inner, layer = self.packed()
        x3 = feats[3]
        if not (LEVEL_STREAMS and x3.is_cuda):
            last = ops.conv2d(x3, inner[3])
            outs = [None, None, None, ops.conv2d(last, layer[3])]
            for i in (2, 1, 0):
                last = ops.conv2d_topdown(feats[i], inner[i], last)      # lateral 1x1 + nearest top-down merge, one launch
                outs[i] = ops.conv2d(last, layer[i])
            od = OrderedDict((str(i), o) for i, o in enumerate(outs))
            od["pool"] = ops.maxpool2d(outs[3], 1, 2, 0)          # LastLevelMaxPool
            return od
        # the lateral / top-down chain stays on the calling stream; the output convs of levels 3, 2, 1 (and the pool level) go to the
        # side stream as soon as their input exists, the 200^2 output conv follows the chain on the calling stream
        cur, side = torch.cuda.current_stream(), _side_stream(x3.device)
        ydt = adt(self)
        outs = [torch.empty(tuple(f.shape[:3]) + (layer[i].K,), dtype=ydt, device=x3.device) for i, f in enumerate(feats)]
        pool = None
        last = ops.conv2d(x3, inner[3])
        # Cross-stream lifetimes by fork / join, not by `record_stream`: a tensor of the calling stream that the side stream reads is
        # kept referenced until `cur.wait_stream(side)` below (its block returns to the calling stream's pool, whose later work is
        # ordered behind the join); a tensor the side stream allocates and the calling stream reads returns to the side stream's pool,
        # and every side-stream phase of the model starts with `side.wait_stream(cur)`.  (`record_stream`-ed blocks are reusable only
        # once the recorded work has RUN: a host that enqueues steps ahead of the device then needs a fresh set per step in flight --
        # +2.4 GB and one hipMalloc per step at BASELINE config 2)
        keep = []
        for i in (3, 2, 1, 0):
            if i < 3:
                keep.append(last)
                last = ops.conv2d_topdown(feats[i], inner[i], last)
            if i == 0:
                ops.conv2d(last, layer[0], out=outs[0])
                break
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ops.conv2d(last, layer[i], out=outs[i])
                if i == 3:
                    pool = ops.maxpool2d(outs[3], 1, 2, 0)      # LastLevelMaxPool
        cur.wait_stream(side)
        del keep
        od = OrderedDict((str(i), o) for i, o in enumerate(outs))
        od["pool"] = pool
        return od


class BackboneWithFPN(nn.Module):
    def __init__(self):
        super().__init__()
        self.body = ResNet50Body()
        self.fpn = FeaturePyramidNetwork()
        self.out_channels = 256

    def forward(self, x, s2d_padded: bool = False):
        return self.fpn(self.body(x, s2d_padded))


def resnet_fpn_backbone(backbone_name="resnet50", pretrained=False, **_):
    if backbone_name != "resnet50":
        raise NotImplementedError("only resnet50 is on the SEAM path (ref models/video_matchrcnn.py:337)")
    if pretrained:
        import warnings
        warnings.warn("pretrained_backbone=True ignored: no network in this environment; load weights with "
                      "load_state_dict() (keys are torchvision-compatible)")
    return BackboneWithFPN()


# ------------------------------------------------------------------------------ RPN (a5)
class RPNHead(nn.Module):
    def __init__(self, in_channels=256, num_anchors=3):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, 3, padding=1)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors, 1)
        self.bbox_pred = nn.Conv2d(in_channels, num_anchors * 4, 1)
        self.num_anchors = num_anchors
        self._pk, self._pk_key = None, None

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        old = prefix + "conv.0.0."           # torchvision >= 0.13 layout
        for s in ("weight", "bias"):
            if old + s in state_dict:
                state_dict[prefix + "conv." + s] = state_dict.pop(old + s)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def packed(self):
        dt = cdt(self)
        key = _key(self.parameters(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                w = torch.cat([self.cls_logits.weight, self.bbox_pred.weight], 0)       # A + 4A rows, one launch
                b = torch.cat([self.cls_logits.bias, self.bbox_pred.bias], 0)
                self._pk = (ops.pack_conv(self.conv.weight, self.conv.bias, pad=1, dtype=dt), ops.pack_conv(w, b, dtype=dt))
            self._pk_key = key
        return self._pk

    def forward(self, feats: Sequence[torch.Tensor]):
        """-> per level ([N,H,W,A] objectness, [N,H,W,4A] deltas), NHWC == torchvision's
        permute(0,2,3,1) order (h, w, anchor[, coord])."""
        a = self.num_anchors
        return [(o[..., :a], o[..., a:]) for o in self.fused(feats)]

    def fused(self, feats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """-> per level the fused head output [N,H,W,A+4A] fp32 (objectness logits | deltas of each pixel)."""
        conv, heads = self.packed()
        feats = list(feats)
        if not (LEVEL_STREAMS and len(feats) > 1 and feats[0].is_cuda):
            return [ops.conv2d(ops.conv2d(f, conv, relu=True), heads, out_f32=True) for f in feats]     # logits/deltas leave in fp32
        # level 0 on the calling stream, the smaller levels on the side stream (see LEVEL_STREAMS)
        cur, side = torch.cuda.current_stream(), _side_stream(feats[0].device)
        outs = [torch.empty(tuple(f.shape[:3]) + (heads.K,), dtype=torch.float32, device=f.device) for f in feats]
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for f, o in zip(feats[1:], outs[1:]):       # (the caller's `feats` outlive the join below: no record_stream, see FPN.forward)
                ops.conv2d(ops.conv2d(f, conv, relu=True), heads, out=o, out_f32=True)
        ops.conv2d(ops.conv2d(feats[0], conv, relu=True), heads, out=outs[0], out_f32=True)
        cur.wait_stream(side)
        return outs


def _base_anchors(size: float, ratios=(0.5, 1.0, 2.0)) -> np.ndarray:
    r = np.asarray(ratios, dtype=np.float32)
    h_r = np.sqrt(r)
    w_r = (np.float32(1.0) / h_r).astype(np.float32)
    ws = (w_r * np.float32(size)).astype(np.float32)
    hs = (h_r * np.float32(size)).astype(np.float32)
    return np.round(np.stack([-ws, -hs, ws, hs], 1) / np.float32(2.0)).astype(np.float32)   # half-to-even


def grid_anchors(padded_hw, feat_hws, sizes=(32, 64, 128, 256, 512)) -> List[np.ndarray]:
    """AnchorGenerator: stride = padded_size // feature_size (integer division: 800//13 = 61);
    order (y, x, anchor)."""
    out = []
    for (fh, fw), s in zip(feat_hws, sizes):
        sh, sw = padded_hw[0] // fh, padded_hw[1] // fw
        xs = (np.arange(fw, dtype=np.float32) * np.float32(sw))
        ys = (np.arange(fh, dtype=np.float32) * np.float32(sh))
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        sh4 = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], 1)
        out.append((sh4[:, None, :] + _base_anchors(s)[None]).reshape(-1, 4).astype(np.float32))
    return out


def batched_nms(boxes, scores, idxs, thr, top_n: int = 0):
    """torchvision ``batched_nms`` (coordinate-offset form) for ONE image -> kept indices by descending
    score (the first ``top_n`` of them when top_n > 0).  The IoU bit-matrix and the greedy scan are HIP kernels;
    the sort is a device sort."""
    if boxes.numel() == 0:
        return torch.zeros((0,), dtype=torch.int64, device=boxes.device)
    off = idxs.to(boxes) * (boxes.max() + 1.0)
    order = torch.argsort(scores, descending=True, stable=True)
    keep = ops.nms_sorted((boxes + off[:, None])[order].contiguous(), thr, max_keep=int(top_n))
    return order[keep.bool()]


def batched_nms_images(boxes, scores, idxs, valid, thr, top_n, prefix: int = 0):
    """``batched_nms`` for a whole batch of images in one set of launches.

    boxes [B,n,4], scores [B,n], idxs [B,n] (class / level id), valid [B,n] bool (entries removed by
    the score / small-box filters).  Per image: offset boxes by idx*(max_coord+1) (max over the
    image's valid boxes, as torchvision computes it on the filtered set), sort by score (stable,
    descending), greedy NMS, keep the first ``top_n`` survivors (the scan kernel stops there).

    prefix > 0: only the ``prefix`` best-scored candidates of each image enter the IoU matrix (its workspace and work grow
    with the square of that number).  Greedy NMS decides a box from higher-scored boxes only, so the result is EXACT for an
    image whenever the prefix already holds ``top_n`` survivors or every valid candidate; ``exact`` says so per image and the
    caller repeats the call without a prefix in the (rare) other case.

    -> (order [B,n] int64: candidate index at each sorted position, sel [B,n] bool: sorted positions
    that survive, exact [B] bool).  No host synchronisation."""
    n = boxes.shape[1]
    neg = torch.finfo(boxes.dtype).min
    mx = torch.where(valid[..., None], boxes, boxes.new_full((), neg)).amax(dim=(1, 2))            # [B]
    off = idxs.to(boxes) * (mx[:, None] + 1.0)
    sc = torch.where(valid, scores, scores.new_full((), -1.0))                                      # invalid sort last
    order = torch.argsort(sc, dim=1, descending=True, stable=True)
    m = n if prefix <= 0 else min(n, int(prefix))
    head = order[:, :m]
    sb = torch.gather(boxes + off[..., None], 1, head[..., None].expand(-1, -1, 4))
    sv = torch.gather(valid, 1, head)
    # invalid entries become far-away degenerate boxes: they suppress nothing, sort behind every valid entry (so the scan's
    # survivor count reaches them only after all valid survivors were counted) and are masked out below
    sb = torch.where(sv[..., None], sb, sb.new_full((), -1.0e8)).contiguous()
    keep = ops.nms_sorted(sb, thr, max_keep=int(top_n)).bool() & sv
    if m == n:
        return order, keep, torch.ones((boxes.shape[0],), dtype=torch.bool, device=boxes.device)
    exact = (keep.sum(1) >= int(top_n)) | (valid.sum(1) <= m)
    full = torch.zeros((boxes.shape[0], n), dtype=torch.bool, device=boxes.device)
    full[:, :m] = keep
    return order, full, exact


def compact_rows(sel: torch.Tensor, vals, cap: int):
    """Per row of ``sel`` [B,m] (at most ``cap`` True entries each): the selected entries of every tensor in ``vals``
    ([B,m] or [B,m,c]) moved to the front, in order -> tensors [B,cap(,c)] (zeros behind the selected ones).  Device-side
    compaction by prefix sum + scatter: no boolean-mask indexing, hence no host synchronisation."""
    b = sel.shape[0]
    pos = sel.cumsum(1) - 1
    idx = torch.where(sel, pos, pos.new_full((), cap))          # unselected entries go to a dump slot behind the row
    out = []
    for v in vals:
        if v.dim() == 3:
            o = v.new_zeros((b, cap + 1, v.shape[2])).scatter_(1, idx[..., None].expand(-1, -1, v.shape[2]), v)
        else:
            o = v.new_zeros((b, cap + 1)).scatter_(1, idx, v)
        out.append(o[:, :cap])
    return out


class RegionProposalNetwork(nn.Module):
    def __init__(self, pre_nms_top_n_test=1000, post_nms_top_n_test=1000, nms_thresh=0.7, min_size=1e-3,
                 pre_nms_top_n_train=2000, post_nms_top_n_train=2000):
        super().__init__()
        self.head = RPNHead()
        if pre_nms_top_n_test < 1 or post_nms_top_n_test < 1:
            raise ValueError("rpn_pre_nms_top_n_test / rpn_post_nms_top_n_test must be positive")
        if 5 * pre_nms_top_n_test > NMS_MAX_BOXES:
            # the batched NMS kernel scans at most NMS_MAX_BOXES candidates per image (5 pyramid levels x pre_nms_top_n)
            raise ValueError(f"rpn_pre_nms_top_n_test = {pre_nms_top_n_test}: 5 levels x top-n exceeds the {NMS_MAX_BOXES} "
                             f"candidates per image seam_nms_sorted_f32 handles (max {NMS_MAX_BOXES // 5})")
        self.pre_nms_top_n, self.post_nms_top_n = pre_nms_top_n_test, post_nms_top_n_test
        self.nms_thresh, self.min_size = nms_thresh, min_size
        self._anchor_cache = {}

    def anchors(self, padded_hw, feat_hws, device):
        key = (tuple(padded_hw), tuple(map(tuple, feat_hws)), str(device))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = [torch.from_numpy(a).to(device) for a in grid_anchors(padded_hw, feat_hws)]
        return self._anchor_cache[key]

    def forward(self, feats: "OrderedDict[str, torch.Tensor]", image_sizes, padded_hw, padded_out: bool = False):
        """RegionProposalNetwork.filter_proposals for the whole batch: per-level top-k + decode + clip + sigmoid in ONE
        launch per level (``seam_rpn_topk_decode_f32``: a radix select over the logits in place, no device sort), small-box
        filter, per-level NMS, first post_nms_top_n -- one host synchronisation (the variable-length split of the
        result) instead of ~20 per image.

        ``padded_out``: return ``(boxes [N, post_nms_top_n, 4], counts [N] int64 on the device)`` instead of the list -- the
        proposals of an image at the front of its row, zeros behind -- with NO host synchronisation (``VideoMatchRCNN.forward``
        hands this form to its own RoI heads: the whole drop-in forward then waits for the device once, at the detections)."""
        fl = list(feats.values())
        n = fl[0].shape[0]
        dev = fl[0].device
        anchors = self.anchors(padded_hw, [f.shape[1:3] for f in fl], dev)
        a = self.head.num_anchors
        ks = [min(self.pre_nms_top_n, f.shape[1] * f.shape[2] * a) for f in fl]
        if max(ks) <= ops.rpn_topk_max():
            heads = self.head.fused(fl)
            ktot = sum(ks)
            bx = torch.empty((n, ktot, 4), dtype=torch.float32, device=dev)
            sc = torch.empty((n, ktot), dtype=torch.float32, device=dev)
            off = 0
            clip = self._clip_hw(image_sizes, dev)
            for o, anc, k in zip(heads, anchors, ks):
                ops.rpn_topk_decode(o, a, anc, clip, k, bx, sc, off)
                off += k
            lv = self._levels(tuple(ks), dev)[None].expand(n, -1)
        else:
            bx, sc, lv = self._topk_by_sort(self.head(fl), anchors, image_sizes, ks, n, dev)
        valid = ((bx[..., 2] - bx[..., 0]) >= self.min_size) & ((bx[..., 3] - bx[..., 1]) >= self.min_size)
        order, sel, _ = batched_nms_images(bx, sc, lv, valid, self.nms_thresh, self.post_nms_top_n)
        kept = torch.gather(bx, 1, order[..., None].expand(-1, -1, 4))
        if padded_out:
            return compact_rows(sel, [kept], self.post_nms_top_n)[0].contiguous(), sel.sum(1)
        counts = sel.sum(1).tolist()                                                    # the one sync
        return list(kept[sel].split(counts, 0))

    def _clip_hw(self, image_sizes, dev):
        key = ("clip", tuple(map(tuple, image_sizes)), str(dev))
        if key not in self._anchor_cache:
            if len(self._anchor_cache) > 64:
                self._anchor_cache.clear()
            self._anchor_cache[key] = torch.tensor([[float(h), float(w)] for h, w in image_sizes], dtype=torch.float32).to(dev)
        return self._anchor_cache[key]

    def _levels(self, ks, dev):
        key = ("lv", ks, str(dev))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = torch.cat([torch.full((k,), l, dtype=torch.int64) for l, k in enumerate(ks)]).to(dev)
        return self._anchor_cache[key]

    def _topk_by_sort(self, head, anchors, image_sizes, ks, n, dev):
        """pre_nms_top_n beyond the select kernel's capacity (1024): per-level stable device sort + decode kernel."""
        bx, sc, lv = [], [], []
        same_size = all(tuple(s) == tuple(image_sizes[0]) for s in image_sizes)
        for l, ((obj, dlt), anc, k) in enumerate(zip(head, anchors, ks)):
            o = obj.reshape(n, -1)
            top = torch.argsort(o, dim=1, descending=True, stable=True)[:, :k]          # ties -> lower index
            d = torch.gather(dlt.reshape(n, -1, 4), 1, top[..., None].expand(-1, -1, 4)).reshape(n * k, 4)
            a = anc[top.reshape(-1)]
            if same_size:
                b = ops.decode_boxes(d.contiguous(), a.contiguous(), (1.0, 1.0, 1.0, 1.0), image_sizes[0])
            else:
                b = torch.cat([ops.decode_boxes(d[i * k:(i + 1) * k].contiguous(), a[i * k:(i + 1) * k].contiguous(),
                                                (1.0, 1.0, 1.0, 1.0), image_sizes[i]) for i in range(n)])
            bx.append(b.view(n, k, 4))
            sc.append(torch.sigmoid(torch.gather(o, 1, top)))
            lv.append(torch.full((n, k), l, dtype=torch.int64, device=dev))
        return torch.cat(bx, 1), torch.cat(sc, 1), torch.cat(lv, 1)


# ------------------------------------------------------------------------------ RoIAlign (a7)
class MultiScaleRoIAlign(nn.Module):
    def __init__(self, featmap_names=("0", "1", "2", "3"), output_size=7, sampling_ratio=2):
        super().__init__()
        self.featmap_names = list(featmap_names)
        self.output_size = output_size if isinstance(output_size, int) else output_size[0]
        self.sampling_ratio = sampling_ratio

    @staticmethod
    def infer_scales(feat_hws, image_sizes):
        hm = max(s[0] for s in image_sizes)
        return [2.0 ** round(math.log2(float(fh) / float(hm))) for fh, _ in feat_hws]

    def forward(self, feats, boxes: Sequence[torch.Tensor], image_sizes) -> torch.Tensor:
        """-> NHWC [sum k_i, P, P, C] (the reference's NCHW view is produced at the model boundary)."""
        fl = [feats[k] for k in self.featmap_names]
        dev = fl[0].device
        # [image index | box] rows: the index column is cached per (counts, device) -- one H2D copy the first time a
        # batch shape is seen, then two cat launches per call instead of two tiny kernels per image
        counts = tuple(int(b.shape[0]) for b in boxes)
        key = (counts, str(dev))
        if getattr(self, "_idx_key", None) != key:
            self._idx = torch.repeat_interleave(torch.arange(len(counts), dtype=torch.float32),
                                                torch.tensor(counts, dtype=torch.int64)).view(-1, 1).to(dev)
            self._idx_key = key
        allb = torch.cat([b.to(torch.float32) for b in boxes], 0) if len(boxes) > 1 else boxes[0].to(torch.float32)
        rois = torch.cat([self._idx, allb.view(-1, 4)], 1).contiguous()
        scales = self.infer_scales([f.shape[1:3] for f in fl], image_sizes)
        k_min = int(round(-math.log2(scales[0])))
        return ops.roi_align(fl, rois, scales, self.output_size, self.sampling_ratio, k_min)


# ------------------------------------------------------------------------------ box branch (a6)
class TwoMLPHead(nn.Module):
    def __init__(self, in_channels=256 * 7 * 7, representation_size=1024):
        super().__init__()
        self.fc6 = nn.Linear(in_channels, representation_size)
        self.fc7 = nn.Linear(representation_size, representation_size)
        self._pk, self._pk_key = None, None

    def packed(self):
        dt = cdt(self)
        key = _key(self.parameters(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                # fc6 over flatten(C,7,7) == a 7x7 valid conv over the NHWC ROI tile
                w6 = self.fc6.weight.view(self.fc6.out_features, 256, 7, 7)
                self._pk = (ops.pack_conv(w6, self.fc6.bias, dtype=dt), ops.pack_conv(self.fc7.weight, self.fc7.bias, dtype=dt))
            self._pk_key = key
        return self._pk

    def forward(self, x):                        # NHWC [K,7,7,256] -> [K,1024]
        fc6, fc7 = self.packed()
        x = ops.conv2d(x, fc6, relu=True).view(x.shape[0], -1)
        return ops.linear(x, fc7, relu=True)


class FastRCNNPredictor(nn.Module):
    def __init__(self, in_channels=1024, num_classes=91):
        super().__init__()
        self.cls_score = nn.Linear(in_channels, num_classes)
        self.bbox_pred = nn.Linear(in_channels, num_classes * 4)
        self.num_classes = num_classes
        self._pk, self._pk_key = None, None

    def forward(self, x):
        dt = cdt(self)
        key = _key(self.parameters(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                self._pk = ops.pack_conv(torch.cat([self.cls_score.weight, self.bbox_pred.weight], 0),
                                         torch.cat([self.cls_score.bias, self.bbox_pred.bias], 0), dtype=dt)
            self._pk_key = key
        o = ops.linear(x, self._pk, out_f32=True)
        return o[:, :self.num_classes], o[:, self.num_classes:]


# ------------------------------------------------------------------------------ mask branch (a8)
class MaskRCNNHeads(nn.Module):
    def __init__(self, in_channels=256, layers=(256, 256, 256, 256)):
        super().__init__()
        c = in_channels
        for i, l in enumerate(layers, 1):
            setattr(self, f"mask_fcn{i}", nn.Conv2d(c, l, 3, padding=1))
            setattr(self, f"relu{i}", nn.ReLU(inplace=True))
            c = l
        self.n = len(layers)
        self._pk, self._pk_key = None, None

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        for i in range(self.n):              # torchvision >= 0.13: ``mask_head.{i}.0.weight``
            for s in ("weight", "bias"):
                old = f"{prefix}{i}.0.{s}"
                if old in state_dict:
                    state_dict[f"{prefix}mask_fcn{i + 1}.{s}"] = state_dict.pop(old)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def forward(self, x):                        # NHWC [K,14,14,256]
        dt = cdt(self)
        key = _key(self.parameters(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                self._pk = [ops.pack_conv(getattr(self, f"mask_fcn{i}").weight, getattr(self, f"mask_fcn{i}").bias, pad=1,
                                          dtype=dt) for i in range(1, self.n + 1)]
            self._pk_key = key
        for pc in self._pk:
            x = ops.conv2d(x, pc, relu=True)
        return x


class MaskRCNNPredictor(nn.Module):
    def __init__(self, in_channels=256, dim_reduced=256, num_classes=91):
        super().__init__()
        self.conv5_mask = nn.ConvTranspose2d(in_channels, dim_reduced, 2, 2, 0)
        self.relu = nn.ReLU(inplace=True)
        self.mask_fcn_logits = nn.Conv2d(dim_reduced, num_classes, 1, 1, 0)
        self.num_classes = num_classes
        self._pk, self._pk_key = None, None

    def forward(self, x):
        """NHWC [K,14,14,256] -> logits [K,14,14,4*ncls]: the 2x2/s2 transposed conv is a 1x1 conv to
        4 sub-pixel channel groups (a,b); the 1x1 logits conv commutes with the depth-to-space, so
        it runs per group and ``seam_mask_select_f32`` reads the 28x28 map straight out of it."""
        dt = cdt(self)
        key = _key(self.parameters(), dt)
        if self._pk is None or key != self._pk_key:
            with torch.no_grad():
                self._pk = (ops.pack_conv(self.conv5_mask.weight, self.conv5_mask.bias, transposed2x2=True, dtype=dt),
                            ops.pack_conv(self.mask_fcn_logits.weight, self.mask_fcn_logits.bias, dtype=dt))
            self._pk_key = key
        up, logits = self._pk
        k = x.shape[0]
        y = ops.conv2d(x, up, relu=True)                                  # [K,14,14,4*256]
        o = ops.linear(y.view(k * 14 * 14 * 4, -1), logits)              # [K*196*4, ncls]
        return o.view(k, 14, 14, 4 * self.num_classes)


def maskrcnn_inference(mask_logits_sub, labels: Sequence[torch.Tensor], num_classes: int):
    """sigmoid + per-ROI label channel -> list of [k_i,1,28,28]."""
    counts = [int(l.numel()) for l in labels]
    prob = ops.mask_select(mask_logits_sub, torch.cat(list(labels)).to(torch.int64).contiguous(), num_classes)
    return list(prob.split(counts, 0))


def paste_masks_in_image(masks: torch.Tensor, boxes: torch.Tensor, img_hw, padding: int = 1) -> torch.Tensor:
    """transform.postprocess mask paste [TV] -> [K,1,H,W] (``seam_paste_masks_f32``).  Read by no caller of
    the reference (stuffs/engine.py:15,118; evaluate_movingfashion.py never touches ``masks``); produced
    for output-dict parity."""
    assert padding == 1 and masks.shape[-1] == 28
    return ops.paste_masks(masks.contiguous(), boxes.to(torch.float32).contiguous(), img_hw)
