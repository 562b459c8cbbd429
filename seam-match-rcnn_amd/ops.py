"""Launch wrappers over the C ABI (``include/seam_hip.h``): torch tensors in, torch tensors out.

torch is plumbing only here -- device memory (``torch.empty``), the current HIP stream
and raw ``data_ptr()`` values handed to ``libseam_hip.so``.  Every op requires CUDA(HIP)
fp32 contiguous tensors and raises otherwise: there is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _native

F32 = torch.float32
F16 = torch.float16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype=F32, name: str = "tensor") -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _native.SeamNativeError(f"{name}: expected a tensor on the HIP device (no CPU path exists)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _req_fp(t: torch.Tensor, name: str = "tensor") -> torch.Tensor:
    """fp32 or fp16 activation (the two precisions the library computes in)."""
    t = _req(t, None, name)
    if t.dtype not in (F32, F16):
        raise TypeError(f"{name}: expected float32 or float16, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ------------------------------------------------------------------------------ conv
@dataclass
class PackedConv:
    """Weights of one conv/linear in the kernel's layout + its fused epilogue vectors."""
    w: torch.Tensor                 # [rows_padded, kred]
    scale: Optional[torch.Tensor]   # [K] or None
    shift: Optional[torch.Tensor]   # [K] or None
    K: int
    Cstore: int
    R: int
    S: int
    stride: int = 1
    pad: int = 0
    Cin: int = 0                    # real (unpadded) input channels: algorithmic FLOP accounting
    dtype: object = F32             # operand precision of the packed weights: torch.float32 | torch.float16 | BX3
    u: Optional[torch.Tensor] = None  # fp32 stride-1 3x3 layers: Winograd F(2x2,3x3) weights (seam_pack_conv_weight_wino_f32)
    u24: Optional[torch.Tensor] = None  # ... and F(2x4,3x3) weights (seam_pack_conv_weight_wino24_f32)
    wn: Optional[torch.Tensor] = None   # fp32 1x1 layers with <= 16 outputs: register-resident weights of seam_linear_narrow_f32
    ws: Optional[torch.Tensor] = None   # fp32 1x1 / stride-1 layers with C <= 256: row-major [K, C] weights (scale folded) of seam_conv1x1_sw_f32
    shift_sw: Optional[torch.Tensor] = None   # ... and its shift vector (zeros when the layer has none)
    wq: Optional[torch.Tensor] = None   # fp32 1x1 / stride-1 layers with C >= 256 (a multiple of 128), K % 128 == 0: fragment-order weights of seam_conv1x1_pc_f32
    wh: Optional[torch.Tensor] = None   # fp16 stride-1 3x3 layers (C, K multiples of 128, or C = K = 64): fragment-order weights of seam_conv3x3_f16pc
    wsh: Optional[torch.Tensor] = None  # fp16 1x1 / stride-1 layers (C multiple of 64, <= 512): row-major fp16 [K, C] weights of seam_conv1x1_swh_f16
    wph: Optional[torch.Tensor] = None  # fp16 1x1 / stride-1 layers with C >= 512 (a multiple of 256), K a multiple of 128: fragment-order weights of seam_conv1x1_f16pc


BX3 = "bf16x3"      # fp32 activations, split-bf16 operands (3 bf16 MFMAs per product, fp32 accumulate)


# Exact-fp32 path: run the stride-1 3x3 layers through the Winograd F(2x2,3x3) kernel (csrc/seam_wino.hip).  All fp32
# arithmetic; SEAM_WINOGRAD=0 keeps every layer on the implicit-GEMM kernel.
import os as _os
WINOGRAD = _os.environ.get("SEAM_WINOGRAD", "1") != "0"
# fp16 path: the stride-1 3x3 layers with C, K multiples of 128 through the producer / consumer kernel (csrc/seam_f16pc.hip);
# SEAM_F16PC=0 keeps them on the implicit GEMM
F16PC = _os.environ.get("SEAM_F16PC", "1") != "0"
F16PC64 = _os.environ.get("SEAM_F16PC64", "1") != "0"      # the C = K = 64 form (layer1's 3x3 layers; A/B switch)
# exact-fp32 path: the long-reduction 1x1 layers (C >= 256, a multiple of 128; K a multiple of 128) through the producer / consumer
# pointwise kernel (csrc/seam_pwpc.hip); SEAM_PWPC=0 keeps them on the implicit GEMM.  Layers the weights-stationary kernel takes
# (C <= 256) stay there.
PWPC = _os.environ.get("SEAM_PWPC", "1") != "0"
# fp16 path: the 1x1 / stride-1 layers through the streaming weights-stationary kernel (csrc/seam_pwh.hip, round 6); SEAM_PWH=0 keeps
# them on the implicit GEMM.  Maps of >= SW_MIN_HW pixels only -- a function of the map, like the fp32 rule.
SWH = _os.environ.get("SEAM_PWH", "1") != "0"
PWHPC = _os.environ.get("SEAM_PWHPC", "1") != "0"       # conv1x1_f16pc: the long-reduction fp16 1x1 layers (csrc/seam_pwhpc.hip)
PWHPC_MIN_C = int(_os.environ.get("SEAM_PWHPC_MIN_C", "1024"))  # reductions from this length on take it (C = 512 stays on conv1x1_swh: 1.04-1.2x alone, nothing in the step)
F16PC_RULE = True      # dispatch by seam_conv3x3_f16pc_pays (False: every supported shape -- tests, tools/f16pc_ab.py)


def _pack_wino(lib, weight: torch.Tensor, K: int, cin: int, cs: int, mode: int) -> Optional[torch.Tensor]:
    if not lib.seam_wino_supported(cs, K, 3, 3, 1):
        return None
    u = torch.empty((int(lib.seam_wino_weight_floats(K, cs)),), dtype=F32, device=weight.device)
    _native.check(lib.seam_pack_conv_weight_wino_f32(_ptr(weight), _ptr(u), K, cin, cs, mode, _stream()),
                  "seam_pack_conv_weight_wino_f32")
    return u


def _pack_wino24(lib, weight: torch.Tensor, K: int, cin: int, cs: int, mode: int) -> Optional[torch.Tensor]:
    if not lib.seam_wino_supported(cs, K, 3, 3, 1):
        return None
    u = torch.empty((int(lib.seam_wino24_weight_floats(K, cs)),), dtype=F32, device=weight.device)
    _native.check(lib.seam_pack_conv_weight_wino24_f32(_ptr(weight), _ptr(u), K, cin, cs, mode, _stream()),
                  "seam_pack_conv_weight_wino24_f32")
    return u


# F(2x4,3x3) (csrc/seam_wino24.hip): 0 = never, 1 = where it issues fewer MFMAs than F(2x2,3x3) (x WINO24_MARGIN), 2 = always
WINOGRAD24 = int(_os.environ.get("SEAM_WINOGRAD24", "1"))
WINO24_MARGIN = float(_os.environ.get("SEAM_WINO24_MARGIN", "0.95"))
_WINO24 = {}


def _wino24_pays(lib, n, h, w, c, k, pad) -> bool:
    if WINOGRAD24 >= 2:
        return True
    # decided on the map geometry alone (a canonical batch of 256 images): an image's result must not depend on the batch it
    # rides in (the two forms round differently), and the stacked layouts of small maps would make the ratio vary with n
    key = (h, w, c, k, pad)
    f = _WINO24.get(key)
    if f is None:
        s24, s22 = int(lib.seam_wino24_issue_slots(256, h, w, c, k, pad)), int(lib.seam_wino_issue_slots(256, h, w, c, k, pad))
        f = _WINO24[key] = s24 > 0 and s24 <= WINO24_MARGIN * s22
    return f


# <= 16-output 1x1 layers (RPN logits / deltas, mask logits) on the row-stream kernel of csrc/seam_narrow.hip (SEAM_NARROW=0: implicit GEMM)
NARROW = _os.environ.get("SEAM_NARROW", "1") != "0"
WINO_MIN_FILL = int(_os.environ.get("SEAM_WINO_MIN_FILL", "55"))    # % of tile slots in use below which the implicit GEMM wins
_WINO_FILL = {}


def _wino_pays(lib, n, h, w, c, k, pad) -> bool:
    # decided on the map geometry alone (a canonical batch of 256 images), like _wino24_pays: batch-independent results
    key = (h, w, c, k, pad)
    f = _WINO_FILL.get(key)
    if f is None:
        f = _WINO_FILL[key] = int(lib.seam_wino_slot_fill_pct(256, h, w, c, k, pad))
    return f >= WINO_MIN_FILL


# Short-reduction pointwise layers (1x1, stride 1, C <= 256) on the weights-stationary kernel of csrc/seam_pw.hip (SEAM_PW=0: implicit
# GEMM).  Decided on the map geometry alone -- maps of >= SW_MIN_HW pixels, any batch -- so that an image's (or a ROI's) result never
# depends on the batch it rides in (the two kernels round differently: folded scale, different k order).
SW = _os.environ.get("SEAM_PW", "1") != "0"
SW_MIN_HW = int(_os.environ.get("SEAM_PW_MIN_HW", "196"))


# maps of >= 50 x 50 pixels (a function of the map alone, like SW_MIN_HW: an image's result never depends on the batch it rides in): on
# the 25 x 25 maps a step's 80 frames are 3-6 tiles per CU and the kernel is level with the implicit GEMM (profiles/r05_pwpc_ab.txt)
PWPC_MIN_HW = int(_os.environ.get("SEAM_PWPC_MIN_HW", "2500"))


def _sw_ok(pc: "PackedConv", h: int, w: int) -> bool:
    return SW and pc.ws is not None and h * w >= SW_MIN_HW


def _sw_launch(lib, x, x2, pc, residual, y, m, c1, c2, relu, res_mode=0, ho=0, wo=0, rh=0, rw=0):
    _native.check(lib.seam_conv1x1_sw_f32(_ptr(x), _ptr(x2), _ptr(pc.ws), _ptr(pc.shift_sw), _ptr(residual), _ptr(y), m, c1, c2, pc.K,
                                          int(relu), res_mode, ho, wo, rh, rw, _stream()), "seam_conv1x1_sw_f32")


def _swh_variant(lib, m, c, k) -> str:
    cfg = int(lib.seam_conv1x1_swh_config(m, c, 0, k))
    return f"conv1x1_swh<{cfg // 100},{cfg % 100}>"


def _sw_variant(lib, m, c, k) -> str:
    cfg = int(lib.seam_conv1x1_sw_config(m, c, 0, k))
    return f"conv1x1_sw<{cfg // 100},{cfg % 100}>"


# When set to a list, every conv launch is bracketed by HIP events on the launch stream and
# (variant, algorithmic_flops, start_event, end_event) is appended (bench.py's roofline leg).
CONV_TRACE = None


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn=None, *, stride: int = 1,
              pad: int = 0, cstore: Optional[int] = None, transposed2x2: bool = False,
              bn_eps: float = 1e-5, dtype: torch.dtype = F32, wino: bool = True) -> PackedConv:
    """Pack a PyTorch-layout weight for ``conv2d``.

    weight: Conv2d [K,Cin,R,S] | Linear [K,Cin] | (transposed2x2) ConvTranspose2d [Cin,Cout,2,2]
    bn:     optional (weight, bias, running_mean, running_var) folded into the epilogue:
            FrozenBatchNorm2d [TV] / BatchNorm1d eval (ref models/match_head.py:62).
    wino:   also pack Winograd F(2x2,3x3) weights for fp32 stride-1 3x3 layers (the inference path); the grad-enabled
            pass of the heads (autograd.py) packs per step and keeps the bit-exact fp32 fma chain of the implicit GEMM."""
    lib = _native.lib()
    weight = _req(weight.detach(), name="weight")
    is_linear = False
    if transposed2x2:
        cin, cout = weight.shape[0], weight.shape[1]
        K, R, S, mode = 4 * cout, 1, 1, 1
    else:
        is_linear = weight.dim() == 2      # a Linear runs on 1x1 maps: no weight form of the map-sized pointwise kernels
        if weight.dim() == 2:
            weight = weight[:, :, None, None]
        if weight.dim() == 3:
            weight = weight[:, :, :, None]
        weight = weight.contiguous()
        K, cin, R, S = weight.shape
        mode = 0
    epv = 8 if dtype == F16 else 4
    cs = cstore if cstore is not None else ((cin + epv - 1) // epv) * epv
    rows = lib.seam_conv_rows_padded(K)
    u = u24 = wh = None
    if dtype == F32:
        kred = lib.seam_conv_kred(cs, R, S)
        wp = torch.empty((rows, kred), dtype=F32, device=weight.device)
        _native.check(lib.seam_pack_conv_weight_f32(_ptr(weight), _ptr(wp), K, cin, R, S, cs, mode, _stream()),
                      "seam_pack_conv_weight_f32")
        if wino and mode == 0 and R == 3 and S == 3 and stride == 1:
            u = _pack_wino(lib, weight, K, cin, cs, 0)
            u24 = _pack_wino24(lib, weight, K, cin, cs, 0) if u is not None else None
    elif dtype == BX3:
        kred = lib.seam_conv_kred(cs, R, S)
        wp = torch.empty((rows, kred), dtype=F32, device=weight.device)      # opaque: [32 hi | 32 lo] bf16 per 128-byte row-chunk
        tmp = torch.empty((rows, kred), dtype=F32, device=weight.device)
        _native.check(lib.seam_pack_conv_weight_bx3(_ptr(weight), _ptr(wp), _ptr(tmp), K, cin, R, S, cs, mode, _stream()),
                      "seam_pack_conv_weight_bx3")
    else:
        kred = lib.seam_conv_kred_f16(cs, R, S)
        wp = torch.empty((rows, kred), dtype=F16, device=weight.device)
        _native.check(lib.seam_pack_conv_weight_f16(_ptr(weight), _ptr(wp), K, cin, R, S, cs, mode, _stream()),
                      "seam_pack_conv_weight_f16")
        if F16PC and mode == 0 and R == 3 and S == 3 and stride == 1 and pad in (0, 1) and ((cs % 128 == 0 and K % 128 == 0) or (cs == 64 and K == 64 and F16PC64)):
            wh = torch.empty((int(lib.seam_f16pc_weight_halves(K, cs)),), dtype=F16, device=weight.device)
            _native.check(lib.seam_pack_conv_weight_f16pc(_ptr(weight), _ptr(wh), K, cin, cs, _stream()), "seam_pack_conv_weight_f16pc")
    scale = shift = None
    if bias is not None:
        bias = bias.detach().to(F32)
        if transposed2x2:
            bias = bias.repeat(4)
    if bn is not None:
        bw, bb, rm, rv = (t.detach().to(F32) for t in bn)
        scale = (bw * torch.rsqrt(rv + bn_eps)).contiguous()
        shift = bb - rm * scale
        if bias is not None:
            shift = shift + bias * scale
        shift = shift.contiguous()
    elif bias is not None:
        shift = bias.contiguous()
    wn = None
    # (fp32 only: the same kernel on fp16 rows -- 8-byte loads, widened -- measured 1.7 % SLOWER end to end than the 128x64 fp16 tile)
    if (dtype == F32 and mode == 0 and R == 1 and S == 1 and stride == 1 and pad == 0 and scale is None and cs == cin
            and lib.seam_linear_narrow_supported(cs, K)):
        wn = torch.empty((64 * cs,), dtype=F32, device=weight.device)
        _native.check(lib.seam_pack_linear_narrow_f32(_ptr(weight), _ptr(wn), K, cs, _stream()), "seam_pack_linear_narrow_f32")
    ws = shift_sw = None
    if (dtype == F32 and mode in (0, 1) and R == 1 and S == 1 and stride == 1 and pad == 0 and cs == cin
            and lib.seam_conv1x1_sw_config(1 << 20, cs, 0, K)):
        # weights-stationary pointwise kernel (csrc/seam_pw.hip): plain row-major [K, C], the per-channel scale folded in
        wm = (weight.permute(2, 3, 1, 0).reshape(K, cin) if transposed2x2 else weight.reshape(K, cin)).to(F32)
        ws = (wm * scale[:, None] if scale is not None else wm).contiguous()
        shift_sw = shift if shift is not None else torch.zeros((K,), dtype=F32, device=weight.device)
    wq = None
    if (PWPC and dtype == F32 and mode == 0 and not is_linear and R == 1 and S == 1 and stride == 1 and pad == 0 and cs == cin and ws is None
            and lib.seam_conv1x1_pc_supported(1 << 20, cs, K)):
        wq = torch.empty((int(lib.seam_conv1x1_pc_weight_floats(K, cs)),), dtype=F32, device=weight.device)
        _native.check(lib.seam_pack_conv1x1_pc_f32(_ptr(weight.reshape(K, cin).to(F32).contiguous()), _ptr(wq), K, cs, _stream()),
                      "seam_pack_conv1x1_pc_f32")
    wsh = None
    if (SWH and dtype == F16 and mode in (0, 1) and not is_linear and R == 1 and S == 1 and stride == 1 and pad == 0 and cs == cin
            and lib.seam_conv1x1_swh_config(1 << 20, cs, 0, K)):
        wm = weight.permute(2, 3, 1, 0).reshape(K, cin) if transposed2x2 else weight.reshape(K, cin)
        wsh = wm.to(F16).contiguous()
    wph = None
    if (PWHPC and dtype == F16 and mode == 0 and not is_linear and R == 1 and S == 1 and stride == 1 and pad == 0 and cs == cin
            and lib.seam_conv1x1_f16pc_supported(1 << 20, cs, K)):
        wph = torch.empty((int(lib.seam_conv1x1_f16pc_weight_halves(K, cs)),), dtype=F16, device=weight.device)
        _native.check(lib.seam_pack_conv1x1_weight_f16pc(_ptr(weight.reshape(K, cin).to(F32).contiguous()), _ptr(wph), K, cs, _stream()),
                      "seam_pack_conv1x1_weight_f16pc")
    return PackedConv(wp, scale, shift, K, cs, R, S, stride, pad, cin, dtype, u, u24, wn, ws, shift_sw, wq, wh, wsh, wph)


def pack_conv_dgrad(weight: torch.Tensor, pad_fwd: int = 0, wino: bool = True) -> PackedConv:
    """Weights of the INPUT-GRADIENT conv of a stride-1 Conv2d / Linear with OIHW weight [Cout,Cin,R,S]:
    dX = conv2d(dY, pack_conv_dgrad(W))  (taps rotated 180 degrees, channels swapped, pad R-1-pad_fwd)."""
    lib = _native.lib()
    weight = _req(weight.detach(), name="weight")
    if weight.dim() == 2:
        weight = weight[:, :, None, None]
    weight = weight.contiguous()
    cout, cin, R, S = weight.shape
    if cout % 32 or R != S:
        raise ValueError("pack_conv_dgrad: Cout must be a multiple of 32 and the kernel square")
    rows = lib.seam_conv_rows_padded(cin)
    kred = lib.seam_conv_kred(cout, R, S)
    wp = torch.empty((rows, kred), dtype=F32, device=weight.device)
    _native.check(lib.seam_pack_conv_weight_f32(_ptr(weight), _ptr(wp), cin, cout, R, S, cout, 2, _stream()),
                  "seam_pack_conv_weight_f32")
    u = _pack_wino(lib, weight, cin, cout, cout, 2) if (wino and R == 3) else None
    u24 = _pack_wino24(lib, weight, cin, cout, cout, 2) if u is not None else None
    return PackedConv(wp, None, None, cin, cout, R, S, 1, R - 1 - pad_fwd, cout, F32, u, u24)


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, R: int, S: int, stride: int = 1, pad: int = 0) -> torch.Tensor:
    """x NHWC [N,H,W,C], dy NHWC [N,Ho,Wo,K] -> dW in PyTorch OIHW layout [K,C,R,S] (fp32 MFMA, split over pixels)."""
    lib = _native.lib()
    x, dy = _req(x, name="x"), _req(dy, name="dy")
    n, h, w, c = x.shape
    k = dy.shape[-1]
    m = dy.numel() // k
    dw = torch.empty((k, c, R, S), dtype=F32, device=x.device)
    ws = torch.empty((int(lib.seam_conv_wgrad_workspace_floats(m, c, k, R, S)),), dtype=F32, device=x.device)
    _native.check(lib.seam_conv_wgrad_f32(_ptr(x), _ptr(dy), _ptr(dw), n, h, w, c, k, R, S, stride, pad, _ptr(ws), _stream()),
                  "seam_conv_wgrad_f32")
    return dw


def colsum(x: torch.Tensor) -> torch.Tensor:
    """[..., K] -> [K] sum over all leading dims (bias gradients)."""
    x = _req(x)
    k = x.shape[-1]
    m = x.numel() // k
    lib = _native.lib()
    out = torch.empty((k,), dtype=F32, device=x.device)
    ws = torch.empty((int(lib.seam_colsum_workspace_floats(m, k)),), dtype=F32, device=x.device)
    _native.check(lib.seam_colsum_f32(_ptr(x), _ptr(out), m, k, _ptr(ws), _stream()), "seam_colsum_f32")
    return out


def avgpool_relu_bwd(dpool: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dpool [N,C], y NHWC [N,H,W,C] (post-ReLU conv output) -> dy [N,H,W,C]."""
    dpool, y = _req(dpool), _req(y)
    n, h, w, c = y.shape
    dy = torch.empty_like(y)
    _native.check(_native.lib().seam_avgpool_relu_bwd_f32(_ptr(dpool), _ptr(y), _ptr(dy), n, h * w, c, _stream()),
                  "seam_avgpool_relu_bwd_f32")
    return dy


def bn1d_train_fwd(x, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """BatchNorm1d with batch statistics; running buffers (or None) updated in place -> (y, save_mean, save_invstd)."""
    x = _req(x)
    m, f = x.shape
    if m < 2:
        raise ValueError("Expected more than 1 value per channel when training, got input size " + str(tuple(x.shape)))
    y = torch.empty_like(x)
    mean = torch.empty((f,), dtype=F32, device=x.device)
    inv = torch.empty((f,), dtype=F32, device=x.device)
    _native.check(_native.lib().seam_bn1d_train_fwd_f32(_ptr(x), _ptr(_req(gamma.detach())), _ptr(_req(beta.detach())), _ptr(y),
                                                        _ptr(mean), _ptr(inv), _ptr(running_mean), _ptr(running_var), m, f,
                                                        float(momentum), float(eps), _stream()), "seam_bn1d_train_fwd_f32")
    return y, mean, inv


def ce2_fwd_bwd(logits: torch.Tensor, target: torch.Tensor, weight: torch.Tensor):
    """Weighted 2-class cross entropy (mean): logits [n,2], target int64 [n], weight [2] -> (loss [], dlogits [n,2])."""
    logits = _req(logits)
    target = _req(target.to(logits.device), torch.int64, "target")
    weight = _req(weight.to(logits.device))
    loss = torch.empty((), dtype=F32, device=logits.device)
    dl = torch.empty_like(logits)
    _native.check(_native.lib().seam_ce2_fwd_bwd_f32(_ptr(logits), _ptr(target), _ptr(weight), _ptr(loss), _ptr(dl),
                                                     logits.shape[0], _stream()), "seam_ce2_fwd_bwd_f32")
    return loss, dl


def bn1d_bwd(dy, x, save_mean, save_invstd, gamma, frozen: bool = False):
    dy, x = _req(dy), _req(x)
    m, f = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty((f,), dtype=F32, device=x.device)
    db = torch.empty((f,), dtype=F32, device=x.device)
    _native.check(_native.lib().seam_bn1d_bwd_f32(_ptr(dy), _ptr(x), _ptr(save_mean), _ptr(save_invstd), _ptr(_req(gamma.detach())),
                                                  _ptr(dx), _ptr(dg), _ptr(db), m, f, int(frozen), _stream()), "seam_bn1d_bwd_f32")
    return dx, dg, db


def pair_logits_bwd(a, b, w, g):
    """Gradients of ``pair_logits``: g [Q,G,2] -> (da [Q,256], db [G,256], dw [2,256], dbias [2])."""
    a, b, w, g = _req(a), _req(b), _req(w.detach()), _req(g)
    q, gg, d = a.shape[0], b.shape[0], a.shape[1]
    da, db = torch.zeros_like(a), torch.zeros_like(b)
    dw = torch.zeros((2, d), dtype=F32, device=a.device)
    dbias = torch.zeros((2,), dtype=F32, device=a.device)
    if q and gg:
        _native.check(_native.lib().seam_pair_logits_bwd_f32(_ptr(a), _ptr(b), _ptr(w), _ptr(g), _ptr(da), _ptr(db), _ptr(dw),
                                                             _ptr(dbias), q, gg, d, _stream()), "seam_pair_logits_bwd_f32")
    return da, db, dw, dbias


def nlb_attnpool_bwd(seq, t_stride, s_stride, lens, n_seq, t_max, pk: "PackedNLB", dout, use_nlb=1):
    """Gradients of ``nlb_attnpool``: dout [S,256] -> (dseq like seq, [11 parameter gradients in the reference's layouts:
    theta.w, theta.b, phi.w, phi.b, g.w, g.b, concat_project.w, W.w, W.b, attention_scorer.w, attention_scorer.b])."""
    lib = _native.lib()
    seq, dout = _req(seq, name="seq"), _req(dout, name="dout")
    dev = seq.device
    dseq = torch.zeros_like(seq)
    shapes = [(128, 256, 1), (128,), (128, 256, 1), (128,), (128, 256, 1), (128,), (1, 256, 1, 1), (256, 128, 1), (256,), (1, 256), (1,)]
    grads = [torch.zeros(sh, dtype=F32, device=dev) for sh in shapes]
    if n_seq:
        ws = torch.empty((int(lib.seam_nlb_bwd_workspace_floats(n_seq, t_max)),), dtype=F32, device=dev)
        arr = (C.c_void_p * 11)(*[g.data_ptr() for g in grads])
        _native.check(lib.seam_nlb_attnpool_bwd_f32(_ptr(seq), t_stride, s_stride, _ptr(_req(lens, torch.int32, "lens")), n_seq, t_max,
                                                    _ptr(pk.w_proj_t), _ptr(pk.b_proj), _ptr(pk.w_cat), _ptr(pk.w_out_t),
                                                    _ptr(pk.b_out), _ptr(pk.w_att), _ptr(pk.b_att), _ptr(dout), _ptr(dseq), arr,
                                                    _ptr(ws), int(use_nlb), _stream()), "seam_nlb_attnpool_bwd_f32")
    return dseq, grads


def nlb_block_bwd(seq, t_stride, s_stride, lens, n_seq, t_max, pk: "PackedNLB", dz, dz_t_stride, dz_s_stride, use_nlb=2):
    """Gradients of the non-local block alone (the ``z`` output of ``nlb_attnpool``): dz rows -> (dseq like seq, [9 parameter
    gradients: theta.w, theta.b, phi.w, phi.b, g.w, g.b, concat_project.w, W.w, W.b])."""
    lib = _native.lib()
    seq, dz = _req(seq, name="seq"), _req(dz, name="dz")
    dev = seq.device
    dseq = torch.zeros_like(seq)
    shapes = [(128, 256, 1), (128,), (128, 256, 1), (128,), (128, 256, 1), (128,), (1, 256, 1, 1), (256, 128, 1), (256,)]
    grads = [torch.zeros(sh, dtype=F32, device=dev) for sh in shapes]
    if n_seq:
        ws = torch.empty((int(lib.seam_nlb_bwd_workspace_floats(n_seq, t_max)),), dtype=F32, device=dev)
        arr = (C.c_void_p * 9)(*[g.data_ptr() for g in grads])
        _native.check(lib.seam_nlb_block_bwd_f32(_ptr(seq), t_stride, s_stride, _ptr(_req(lens, torch.int32, "lens")), n_seq, t_max,
                                                 _ptr(pk.w_proj_t), _ptr(pk.b_proj), _ptr(pk.w_cat), _ptr(pk.w_out_t), _ptr(pk.b_out),
                                                 _ptr(dz), dz_t_stride, dz_s_stride, _ptr(dseq), arr, _ptr(ws), int(use_nlb), _stream()),
                      "seam_nlb_block_bwd_f32")
    return dseq, grads


def conv2d(x: torch.Tensor, pc: PackedConv, relu: bool = False, residual: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, out_f32: bool = False, out_hw: Optional[tuple] = None) -> torch.Tensor:
    """NHWC implicit-GEMM conv (+scale/shift, +residual, +ReLU) -> NHWC [N,Ho,Wo,K].
    The precision (fp32 exact / fp16 MFMA with fp32 accumulate) is that of the packed weights;
    ``out_f32`` makes the fp16 kernel write fp32 (hand-off to the fp32 descriptor heads)."""
    x = _req(x, None, "x")             # device check first: CPU tensors fail loudly
    x = _req(x, F16 if pc.dtype == F16 else F32, "x")
    n, h, w, c = x.shape
    if c != pc.Cstore:
        raise ValueError(f"conv2d: input has {c} channels, weights packed for {pc.Cstore}")
    ho = (h + 2 * pc.pad - pc.R) // pc.stride + 1
    wo = (w + 2 * pc.pad - pc.S) // pc.stride + 1
    if out_hw is not None:      # top-left crop of the output grid (``seam_conv2d_crop_f32``: the space-to-depth stem)
        if pc.dtype not in (F32, F16) or out_f32 or residual is not None or out_hw[0] > ho or out_hw[1] > wo:
            raise ValueError("conv2d: out_hw needs fp32 or fp16 weights, no residual and a size inside the full output")
        ho, wo = int(out_hw[0]), int(out_hw[1])
    narrow = pc.wn is not None and NARROW and residual is None and out_hw is None
    ydt = F32 if (pc.dtype != F16 or out_f32) else F16
    if out is not None:
        if (not isinstance(out, torch.Tensor) or not out.is_cuda or out.device != x.device or tuple(out.shape) != (n, ho, wo, pc.K)
                or out.dtype != ydt or not out.is_contiguous()):
            raise ValueError(f"conv2d: out must be a contiguous {ydt} tensor of shape {(n, ho, wo, pc.K)} on {x.device}")
    y = out if out is not None else torch.empty((n, ho, wo, pc.K), dtype=ydt, device=x.device)
    if residual is not None:
        residual = _req(residual, F16 if pc.dtype == F16 else F32, "residual")
        if residual.shape != y.shape:
            raise ValueError("conv2d: residual shape mismatch")
    trace = CONV_TRACE
    if trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib = _native.lib()
    wino = pc.dtype == F32 and pc.u is not None and WINOGRAD and out_hw is None and _wino_pays(lib, n, h, w, c, pc.K, pc.pad)
    wino24 = wino and pc.u24 is not None and WINOGRAD24 and _wino24_pays(lib, n, h, w, c, pc.K, pc.pad)
    sw = not narrow and pc.dtype == F32 and out_hw is None and _sw_ok(pc, h, w)
    pwpc = (not narrow and not sw and pc.dtype == F32 and pc.wq is not None and PWPC and out_hw is None and h * w >= PWPC_MIN_HW
            and lib.seam_conv1x1_pc_supported(n * h * w, c, pc.K) == 1)
    f16pc = (pc.dtype == F16 and pc.wh is not None and F16PC and residual is None and out_hw is None and not out_f32 and relu in (0, 1, False, True)
             and (lib.seam_conv3x3_f16pc_pays if F16PC_RULE else lib.seam_conv3x3_f16pc_supported)(n, h, w, c, pc.K, pc.pad) == 1)
    pwhpc = (pc.dtype == F16 and pc.wph is not None and PWHPC and residual is None and out_hw is None and not out_f32
             and relu in (0, 1, False, True) and c >= PWHPC_MIN_C and h * w >= SW_MIN_HW
             and lib.seam_conv1x1_f16pc_supported(n * h * w, c, pc.K) == 1)
    swh = (not pwhpc and pc.dtype == F16 and pc.wsh is not None and SWH and out_hw is None and not out_f32 and h * w >= SW_MIN_HW
           and relu in (0, 1, False, True))
    if pwhpc:
        _native.check(lib.seam_conv1x1_f16pc(_ptr(x), _ptr(pc.wph), _ptr(pc.scale), _ptr(pc.shift), None, _ptr(y), n * h * w, c, pc.K,
                                             1 if relu else 0, _stream()), "seam_conv1x1_f16pc")
    elif swh:
        _native.check(lib.seam_conv1x1_swh_f16(_ptr(x), None, _ptr(pc.wsh), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                               n * h * w, c, 0, pc.K, 1 if relu else 0, 1 if residual is not None else 0, 0, 0, 0, 0,
                                               _stream()), "seam_conv1x1_swh_f16")
    elif narrow:
        _native.check(lib.seam_linear_narrow_f32(_ptr(x), _ptr(pc.wn), _ptr(pc.shift), _ptr(y), n * h * w, c, pc.K, int(relu), _stream()),
                      "seam_linear_narrow_f32")
    elif sw:
        _sw_launch(lib, x, None, pc, residual, y, n * h * w, c, 0, relu, 1 if residual is not None else 0)
    elif pwpc:
        _native.check(lib.seam_conv1x1_pc_f32(_ptr(x), _ptr(pc.wq), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                              n * h * w, c, pc.K, int(relu), _stream()), "seam_conv1x1_pc_f32")
    elif wino24:
        _native.check(lib.seam_conv3x3_wino24_f32(_ptr(x), _ptr(pc.u24), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                                  n, h, w, c, pc.K, pc.pad, int(relu), _stream()), "seam_conv3x3_wino24_f32")
    elif wino:
        _native.check(lib.seam_conv3x3_wino_f32(_ptr(x), _ptr(pc.u), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                                n, h, w, c, pc.K, pc.pad, int(relu), _stream()), "seam_conv3x3_wino_f32")
    elif out_hw is not None:            # (fp32 or fp16: checked above)
        crop = lib.seam_conv2d_crop_f16 if pc.dtype == F16 else lib.seam_conv2d_crop_f32
        _native.check(crop(_ptr(x), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.shift), _ptr(y), n, h, w, c, pc.K,
                           pc.R, pc.S, pc.stride, pc.pad, ho, wo, int(relu), _stream()), "seam_conv2d_crop")
    elif pc.dtype == F32:
        _native.check(lib.seam_conv2d_f32(_ptr(x), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                          n, h, w, c, pc.K, pc.R, pc.S, pc.stride, pc.pad, int(relu), _stream()),
                      "seam_conv2d_f32")
    elif pc.dtype == BX3:
        _native.check(lib.seam_conv2d_bx3(_ptr(x), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                          n, h, w, c, pc.K, pc.R, pc.S, pc.stride, pc.pad, int(relu), _stream()),
                      "seam_conv2d_bx3")
    elif f16pc:
        _native.check(lib.seam_conv3x3_f16pc(_ptr(x), _ptr(pc.wh), _ptr(pc.scale), _ptr(pc.shift), None, _ptr(y),
                                             n, h, w, c, pc.K, pc.pad, 1 if relu else 0, _stream()), "seam_conv3x3_f16pc")
    else:
        _native.check(lib.seam_conv2d_f16(_ptr(x), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.shift), _ptr(residual), _ptr(y),
                                          n, h, w, c, pc.K, pc.R, pc.S, pc.stride, pc.pad, 1 if relu else 0,
                                          1 if out_f32 else 0, _stream()), "seam_conv2d_f16")
    if trace is not None:
        e1.record()
        tile = lib.seam_conv_tile_taps(2 if pc.dtype == BX3 else 1 if pc.dtype == F16 else 0, n * ho * wo, pc.K, pc.R * pc.S)
        if pwhpc:
            variant = "conv1x1_f16pc"
        elif swh:
            variant = _swh_variant(lib, n * h * w, c, pc.K)
        elif narrow:
            variant = "linear_narrow"
        elif sw:
            variant = _sw_variant(lib, n * h * w, c, pc.K)
        elif pwpc:
            variant = "conv1x1_pc"
        elif wino24:
            variant = ("conv3x3_wino24pc" if lib.seam_wino24_form(n, h, w, c, pc.K, pc.pad) == 1
                       else f"conv3x3_wino24<{lib.seam_wino24_variant(n, h, w, c, pc.K, pc.pad)}>")
        elif wino:
            variant = f"conv3x3_wino<{lib.seam_wino_tile_variant(n, h, w, c, pc.K, pc.pad)}>"
        elif f16pc:
            variant = "conv3x3_f16pc"
        elif pc.dtype == BX3:
            variant = f"conv_igemm_bx3<{tile // 1000},{tile % 1000}>"
        else:
            variant = f"conv_igemm<{'float' if pc.dtype == F32 else '_Float16'},{tile // 1000},{tile % 1000}>"
        es = x.element_size()
        trace.append((variant, 2.0 * n * ho * wo * pc.K * pc.R * pc.S * (pc.Cin or pc.Cstore), e0, e1,
                      (n, h, w, c, pc.K, pc.R, pc.stride),
                      float(es * (x.numel() + pc.w.numel()) + y.element_size() * y.numel()
                            + (es * y.numel() if residual is not None else 0))))
    return y


def stem_s2d_f16(x: torch.Tensor, w_rows: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
                 relu: bool = True, padded: bool = False) -> torch.Tensor:
    """ResNet stem of the fp16 path on the space-to-depth frame ``x`` [N,H2,W2,16] (``preprocess(..., s2d=True)``) through the streaming
    kernel (``seam_stem_s2d_swh_f16``): the frame is zero-padded (2 cells before, 1 after) so that a tap is a constant shift of the
    flattened cell index -- ``padded=True``: ``x`` is that frame already, [N,H2+3,W2+3,16] (``preprocess(..., s2d_pad=(2, 1))``), else
    one padding copy is made.  ``w_rows`` fp16 [64,256], k = (4 r + s) * 16 + channel.  -> NHWC [N,H2,W2,64] fp16."""
    x = _req(x, F16, "x")
    n, h2, w2, c = x.shape
    if padded:
        h2, w2 = h2 - 3, w2 - 3
    if c != 16 or h2 < 1 or w2 < 1 or tuple(w_rows.shape) != (64, 256) or w_rows.dtype != F16:
        raise ValueError("stem_s2d_f16: x must be [N,H2,W2,16] fp16 (padded: +3 cells per direction) and w_rows [64,256] fp16")
    xp = x if padded else torch.nn.functional.pad(x, (0, 0, 2, 1, 2, 1))
    y = torch.empty((n, h2, w2, 64), dtype=F16, device=x.device)
    trace = CONV_TRACE
    if trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _native.check(_native.lib().seam_stem_s2d_swh_f16(_ptr(xp), _ptr(w_rows), _ptr(scale), _ptr(shift), _ptr(y), n, h2, w2,
                                                      1 if relu else 0, _stream()), "seam_stem_s2d_swh_f16")
    if trace is not None:
        e1.record()
        trace.append(("stem_swh<4,2>", 2.0 * n * h2 * w2 * 64 * 16 * 12, e0, e1, (n, h2, w2, 16, 64, 4, 1),
                      float(2 * (xp.numel() + w_rows.numel() + y.numel()))))
    return y


def conv2d_topdown(x: torch.Tensor, pc: PackedConv, top: torch.Tensor) -> torch.Tensor:
    """FPN top-down merge [TV]: conv(x) + nearest-upsample(top) -> NHWC.  Exact-fp32 weights: ONE launch (the coarse map is
    added in the conv epilogue, ``seam_conv2d_upres_f32``); other precisions: conv + ``upsample_add_``.  Both forms round
    identically."""
    if (pc.dtype == F16 and pc.wsh is not None and SWH and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4
            and x.shape[1] * x.shape[2] >= max(SW_MIN_HW, 128) and top.dim() == 4 and top.shape[0] == x.shape[0] and top.shape[3] == pc.K):
        # fp16 path (round 6): the merge in the streaming pointwise kernel's epilogue -- the lateral is never written and re-read
        x, top = _req(x, F16, "x"), _req(top, F16, "top")
        n, h, w, c = x.shape
        if c != pc.Cstore:
            raise ValueError(f"conv2d_topdown: input has {c} channels, weights packed for {pc.Cstore}")
        y = torch.empty((n, h, w, pc.K), dtype=F16, device=x.device)
        trace = CONV_TRACE
        if trace is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        lib = _native.lib()
        _native.check(lib.seam_conv1x1_swh_f16(_ptr(x), None, _ptr(pc.wsh), _ptr(pc.scale), _ptr(pc.shift), _ptr(top), _ptr(y),
                                               n * h * w, c, 0, pc.K, 0, 2, h, w, top.shape[1], top.shape[2], _stream()),
                      "seam_conv1x1_swh_f16")
        if trace is not None:
            e1.record()
            trace.append((_swh_variant(lib, n * h * w, c, pc.K), 2.0 * n * h * w * pc.K * (pc.Cin or pc.Cstore), e0, e1,
                          (n, h, w, c, pc.K, 1, 1), float(2 * (x.numel() + pc.w.numel() + y.numel() + top.numel()))))
        return y
    if pc.dtype != F32 or pc.K % 4:
        return upsample_add_(conv2d(x, pc), top)
    x = _req(x, None, "x")
    x, top = _req(x, F32, "x"), _req(top, F32, "top")
    n, h, w, c = x.shape
    if c != pc.Cstore:
        raise ValueError(f"conv2d_topdown: input has {c} channels, weights packed for {pc.Cstore}")
    ho = (h + 2 * pc.pad - pc.R) // pc.stride + 1
    wo = (w + 2 * pc.pad - pc.S) // pc.stride + 1
    if top.dim() != 4 or top.shape[0] != n or top.shape[3] != pc.K:
        raise ValueError("conv2d_topdown: top must be NHWC [N,Ht,Wt,K]")
    y = torch.empty((n, ho, wo, pc.K), dtype=F32, device=x.device)
    trace = CONV_TRACE
    if trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib = _native.lib()
    sw = _sw_ok(pc, h, w) and h * w >= 64
    if sw:
        _sw_launch(lib, x, None, pc, top, y, n * h * w, c, 0, False, 2, ho, wo, top.shape[1], top.shape[2])
    else:
        _native.check(lib.seam_conv2d_upres_f32(_ptr(x), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.shift), _ptr(top), _ptr(y),
                                                n, h, w, c, pc.K, pc.R, pc.S, pc.stride, pc.pad, top.shape[1], top.shape[2], 0,
                                                _stream()), "seam_conv2d_upres_f32")
    if trace is not None:
        e1.record()
        tile = lib.seam_conv_tile_prec(0, n * ho * wo, pc.K)
        trace.append((_sw_variant(lib, n * h * w, c, pc.K) if sw else f"conv_igemm<float,{tile // 1000},{tile % 1000}>",
                      2.0 * n * ho * wo * pc.K * pc.R * pc.S * (pc.Cin or pc.Cstore), e0, e1,
                      (n, h, w, c, pc.K, pc.R, pc.stride),
                      float(4 * (x.numel() + pc.w.numel() + y.numel() + top.numel()))))
    return y


def pack_conv_dual(w1: torch.Tensor, bn1, w2: torch.Tensor, bn2, bn_eps: float = 1e-5, dtype=F32) -> PackedConv:
    """Weights of ``conv2d_dual``: two 1x1 convs, each followed by its own (Frozen)BatchNorm, summed.
    The scales are folded into the weights (one fp32 rounding per weight) and the shifts added:
    bn_a(W_a . h) + bn_b(W_b . x) = [s_a W_a | s_b W_b] . [h ; x] + (t_a + t_b)."""
    def fold(w, bn):
        bw, bb, rm, rv = (t.detach().to(F32) for t in bn)
        sc = bw * torch.rsqrt(rv + bn_eps)
        return w.detach().to(F32).reshape(w.shape[0], -1) * sc[:, None], bb - rm * sc
    wa, ta = fold(w1, bn1)
    wb, tb = fold(w2, bn2)
    w = torch.cat([wa, wb], 1).contiguous()
    return pack_conv(w[:, :, None, None], (ta + tb).contiguous(), wino=False, dtype=dtype)


def conv2d_dual(x1: torch.Tensor, x2: torch.Tensor, pc: PackedConv, stride2: int = 1, relu: bool = False) -> torch.Tensor:
    """y = act(W[:, :C1] . x1 + W[:, C1:] . x2[:, ::stride2, ::stride2] + shift) in ONE launch (``seam_conv2d_dual_f32`` / ``_f16``):
    the residual block with a projection shortcut [TV Bottleneck.forward] without writing the shortcut branch to memory.
    x1 NHWC [N,Ho,Wo,C1], x2 NHWC [N,H2,W2,C2] -> NHWC [N,Ho,Wo,K]; exact fp32 or fp16 operands (the packed weights' precision)."""
    x1 = _req(x1, None, "x1")
    if pc.dtype not in (F32, F16) or pc.R != 1 or pc.S != 1:
        raise ValueError("conv2d_dual: needs fp32 / fp16 1x1 weights from pack_conv_dual")
    dt = pc.dtype
    x1, x2 = _req(x1, dt, "x1"), _req(x2, dt, "x2")
    n, ho, wo, c1 = x1.shape
    n2, h2, w2, c2 = x2.shape
    cm = 32 if dt == F32 else 64
    if n2 != n or c1 + c2 != pc.Cstore or c1 % cm or c2 % cm:
        raise ValueError(f"conv2d_dual: channel / batch mismatch (C1, C2 must be multiples of {cm} and sum to the packed width)")
    if (ho - 1) * stride2 >= h2 or (wo - 1) * stride2 >= w2:
        raise ValueError("conv2d_dual: x2 too small for the output grid")
    y = torch.empty((n, ho, wo, pc.K), dtype=dt, device=x1.device)
    trace = CONV_TRACE
    if trace is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib = _native.lib()
    sw = dt == F32 and stride2 == 1 and (h2, w2) == (ho, wo) and _sw_ok(pc, ho, wo)
    swh = (dt == F16 and stride2 == 1 and (h2, w2) == (ho, wo) and SWH and ho * wo >= SW_MIN_HW
           and lib.seam_conv1x1_swh_config(n * ho * wo, c1, c2, pc.K) != 0)
    if swh and pc.wsh is None:            # (the dual weights are packed as ONE [K, C1 + C2] matrix: take the fp16 rows from it once)
        swh = False
    if sw:
        _sw_launch(lib, x1, x2, pc, None, y, n * ho * wo, c1, c2, relu)
    elif swh:
        _native.check(lib.seam_conv1x1_swh_f16(_ptr(x1), _ptr(x2), _ptr(pc.wsh), _ptr(pc.scale), _ptr(pc.shift), None, _ptr(y),
                                               n * ho * wo, c1, c2, pc.K, 1 if relu else 0, 0, 0, 0, 0, 0, _stream()), "seam_conv1x1_swh_f16")
    else:
        fn = lib.seam_conv2d_dual_f32 if dt == F32 else lib.seam_conv2d_dual_f16
        _native.check(fn(_ptr(x1), _ptr(x2), _ptr(pc.w), _ptr(pc.scale), _ptr(pc.shift), _ptr(y), n, ho, wo, c1,
                         h2, w2, c2, stride2, pc.K, int(relu), _stream()), "seam_conv2d_dual")
    if trace is not None:
        e1.record()
        tile = lib.seam_conv_tile_prec(0 if dt == F32 else 1, n * ho * wo, pc.K)
        if tile // 1000 == 256:
            tile = 128128
        es = x1.element_size()
        trace.append((_sw_variant(lib, n * ho * wo, c1 + c2, pc.K) if sw else _swh_variant(lib, n * ho * wo, c1 + c2, pc.K) if swh else f"conv_igemm<{'float' if dt == F32 else '_Float16'},{tile // 1000},{tile % 1000}>", 2.0 * n * ho * wo * pc.K * (c1 + c2), e0, e1,
                      (n, ho, wo, c1 + c2, pc.K, 1, 1),
                      float(es * (x1.numel() + n * ho * wo * c2 + pc.w.numel() + y.numel()))))
    return y


def linear(x: torch.Tensor, pc: PackedConv, relu: bool = False, out_f32: bool = False) -> torch.Tensor:
    """[M,C] x packed [K,C] -> [M,K] through the conv kernel (1x1 on a 1x1 map)."""
    m, c = x.shape
    return conv2d(x.view(m, 1, 1, c), pc, relu, out_f32=out_f32).view(m, pc.K)


def match_trunk(x: torch.Tensor, convs: Sequence[PackedConv], lin: PackedConv) -> torch.Tensor:
    """The match trunk in ONE ABI call (seam_match_trunk_f32): x NHWC fp32 [K,14,14,256], the four packed 3x3 convs and the packed
    Linear + BatchNorm1d -> x3 [K,256].  The form of every conv is decided here, by the rule ``conv2d`` applies (so the Python
    knobs SEAM_WINOGRAD / SEAM_WINOGRAD24 / SEAM_WINO_MIN_FILL act on both paths alike) and handed down."""
    lib = _native.lib()
    x = _req(x, F32, "x")
    k = x.shape[0]
    if tuple(x.shape[1:]) != (14, 14, 256) or len(convs) != 4 or any(pc.dtype != F32 for pc in list(convs) + [lin]):
        raise ValueError("match_trunk: fp32 NHWC [K,14,14,256] input and fp32-packed layers expected")
    out = torch.empty((k, 256), dtype=F32, device=x.device)
    if k == 0:
        return out

    def layer(pc):
        return _native.TrunkLayer(*(None if t is None else t.data_ptr() for t in (pc.w, pc.u, pc.u24, pc.scale, pc.shift)))
    layers = (_native.TrunkLayer * 4)(*[layer(pc) for pc in convs])
    form = (C.c_int * 4)()
    for i, (pc, h) in enumerate(zip(convs, (14, 12, 10, 8))):
        wino = pc.u is not None and WINOGRAD and _wino_pays(lib, k, h, h, 256, pc.K, 0)
        form[i] = 2 if (wino and pc.u24 is not None and WINOGRAD24 and _wino24_pays(lib, k, h, h, 256, pc.K, 0)) else (1 if wino else 0)
    ws = torch.empty((int(lib.seam_match_trunk_workspace_floats(k)),), dtype=F32, device=x.device)
    _native.check(lib.seam_match_trunk_f32(_ptr(x), layers, C.byref(layer(lin)), _ptr(out), k, _ptr(ws), form, _stream()),
                  "seam_match_trunk_f32")
    return out


# ------------------------------------------------------------------------------ elementwise
def _sfx(dtype) -> str:
    return "f32" if dtype == F32 else "f16"


def preprocess(images: Sequence[torch.Tensor], sizes: Sequence[tuple], hp: int, wp: int, dtype=F32, s2d: bool = False,
               s2d_pad: tuple = (0, 0)) -> torch.Tensor:
    """normalise + resize + pad + CHW->NHWC for a list of fp32 images -> [N,hp,wp,4] fp32 / [N,hp,wp,8] fp16.
    ``s2d`` (fp32 [3,H,W] images only): the space-to-depth layout [N,hp/2,wp/2,12] of ``seam_preprocess_s2d_batch_f32``
    (fp16: [N,hp/2,wp/2,16], four zero channels, ``_f16``); ``s2d_pad`` = (lo, hi) (fp16 only): zero cells before / after the frame
    in both directions -- [N, hp/2 + lo + hi, wp/2 + lo + hi, 16], the input of ``stem_s2d_f16(..., padded=True)``."""
    lib = _native.lib()
    # device check FIRST, for every image and every branch below: a CPU tensor must raise SeamNativeError, never reach a
    # kernel as a raw host pointer (the one-launch batch branches take data_ptr() of the views directly)
    images = [_req(i, None, "image") for i in images]
    if not images:
        raise ValueError("preprocess: empty image list")
    if s2d:
        if dtype not in (F32, F16) or hp % 2 or wp % 2 or any(i.dtype != F32 or i.dim() != 3 or i.shape[0] != 3 for i in images):
            raise ValueError("preprocess(s2d=True): fp32 [3,H,W] images and an even padded size")
        n = len(images)
        plo, phi = (int(s2d_pad[0]), int(s2d_pad[1])) if dtype == F16 else (0, 0)
        out = torch.empty((n, hp // 2 + plo + phi, wp // 2 + plo + phi, 12 if dtype == F32 else 16), dtype=dtype, device=images[0].device)
        if dtype == F32:
            s2d_fn = lib.seam_preprocess_s2d_batch_f32
        else:
            def s2d_fn(src, stride, dst, cnt, ih, iw, oh, ow, hpp, wpp, st):
                return lib.seam_preprocess_s2d_pad_batch_f16(src, stride, dst, cnt, ih, iw, oh, ow, hpp, wpp, plo, phi, st)
        same = n <= 65535 and all(i.shape == images[0].shape and i.is_contiguous() for i in images) \
            and all(tuple(z) == tuple(sizes[0]) for z in sizes)
        p0 = images[0].data_ptr()
        step = (images[1].data_ptr() - p0) if n > 1 else 0
        if same and n > 1 and step > 0 and step % 4 == 0 and all(im.data_ptr() == p0 + k * step for k, im in enumerate(images)):
            _native.check(s2d_fn(C.c_void_p(p0), step // 4, _ptr(out), n, images[0].shape[1],
                                                            images[0].shape[2], sizes[0][0], sizes[0][1], hp, wp, _stream()),
                          "seam_preprocess_s2d_batch")
            return out
        for i, (img, (oh, ow)) in enumerate(zip(images, sizes)):
            img = _req(img, name="image")
            _native.check(s2d_fn(_ptr(img), 0, C.c_void_p(out[i].data_ptr()), 1, img.shape[1], img.shape[2],
                                                            oh, ow, hp, wp, _stream()), "seam_preprocess_s2d_batch")
        return out
    cs = 4 if dtype == F32 else 8
    out = torch.empty((len(images), hp, wp, cs), dtype=dtype, device=images[0].device)
    fn = lib.seam_preprocess_f32 if dtype == F32 else lib.seam_preprocess_f16
    # the frames of one clip tensor (a list of same-shape fp32 views at a constant stride): one launch for the batch
    n = len(images)
    if n > 1 and n <= 65535 and all(i.dtype == F32 and i.dim() == 3 and i.shape == images[0].shape and i.is_contiguous()
                                     and i.is_cuda for i in images) \
            and images[0].shape[0] == 3 and all(tuple(s) == tuple(sizes[0]) for s in sizes):
        p0 = images[0].data_ptr()
        step = images[1].data_ptr() - p0
        if step > 0 and step % 4 == 0 and all(im.data_ptr() == p0 + k * step for k, im in enumerate(images)):
            fb = lib.seam_preprocess_batch_f32 if dtype == F32 else lib.seam_preprocess_batch_f16
            _native.check(fb(C.c_void_p(p0), step // 4, _ptr(out), n, images[0].shape[1], images[0].shape[2], sizes[0][0],
                             sizes[0][1], hp, wp, _stream()), "seam_preprocess_batch")
            return out
    for i, (img, (oh, ow)) in enumerate(zip(images, sizes)):
        if img.dtype == torch.uint8:        # extension: raw HWC RGB frame, ToTensor fused (row f4)
            img = _req(img, torch.uint8, "image")
            if img.dim() != 3 or img.shape[2] != 3:
                raise ValueError("uint8 images must be [H,W,3]")
            _native.check(lib.seam_preprocess_u8(_ptr(img), C.c_void_p(out[i].data_ptr()), img.shape[0], img.shape[1], oh, ow,
                                                 hp, wp, 0 if dtype == F32 else 1, _stream()), "seam_preprocess_u8")
            continue
        img = _req(img, name="image")
        if img.dim() != 3 or img.shape[0] != 3:
            raise ValueError("images must be [3,H,W]")
        _native.check(fn(_ptr(img), C.c_void_p(out[i].data_ptr()), img.shape[1], img.shape[2], oh, ow, hp, wp, _stream()),
                      "seam_preprocess")
    return out


def maxpool2d(x: torch.Tensor, k: int, stride: int, pad: int) -> torch.Tensor:
    x = _req_fp(x)
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    fn = getattr(_native.lib(), "seam_maxpool2d_" + _sfx(x.dtype))
    _native.check(fn(_ptr(x), _ptr(y), n, h, w, c, k, stride, pad, _stream()), "seam_maxpool2d")
    return y


def upsample_add_(lat: torch.Tensor, top: torch.Tensor) -> torch.Tensor:
    lat, top = _req_fp(lat), _req(top, lat.dtype)
    n, h, w, c = lat.shape
    fn = getattr(_native.lib(), "seam_upsample_add_" + _sfx(lat.dtype))
    _native.check(fn(_ptr(lat), _ptr(top), n, h, w, top.shape[1], top.shape[2], c, _stream()), "seam_upsample_add")
    return lat


def nchw_to_nhwc(x: torch.Tensor, dtype=F32) -> torch.Tensor:
    """fp32 [B,C,*sp] (the reference's layout) -> [B,*sp,C] in the library's compute dtype."""
    x = _req(x)
    b, c = x.shape[0], x.shape[1]
    l = x[0, 0].numel() if b else 0
    y = torch.empty((b,) + tuple(x.shape[2:]) + (c,), dtype=dtype, device=x.device)
    if b:
        fn = _native.lib().seam_nchw_to_nhwc_f32 if dtype == F32 else _native.lib().seam_nchw_f32_to_nhwc_f16
        _native.check(fn(_ptr(x), _ptr(y), b, c, l, _stream()), "seam_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    """[B,*sp,C] fp32 / fp16 -> fp32 [B,C,*sp] (the reference's layout and dtype)."""
    x = _req_fp(x)
    b, c = x.shape[0], x.shape[-1]
    sp = tuple(x.shape[1:-1])
    l = 1
    for s in sp:
        l *= s
    y = torch.empty((b, c) + sp, dtype=F32, device=x.device)
    if b:
        fn = _native.lib().seam_nhwc_to_nchw_f32 if x.dtype == F32 else _native.lib().seam_nhwc_f16_to_nchw_f32
        _native.check(fn(_ptr(x), _ptr(y), b, l, c, _stream()), "seam_nhwc_to_nchw")
    return y


def avgpool(x: torch.Tensor) -> torch.Tensor:
    """NHWC [K,h,w,C] -> [K,C] mean over the spatial positions (fp32 accumulate)."""
    x = _req_fp(x)
    k, c = x.shape[0], x.shape[-1]
    l = x[0].numel() // c if k else 0
    y = torch.empty((k, c), dtype=x.dtype, device=x.device)
    if k:
        fn = getattr(_native.lib(), "seam_avgpool_" + _sfx(x.dtype))
        _native.check(fn(_ptr(x), _ptr(y), k, l, c, _stream()), "seam_avgpool")
    return y


def cat_rows(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """``torch.cat(parts, 0)`` without the copy when the parts are consecutive dim-0 slices of one allocation with equal
    strides (the per-image ``roi_features`` / ``match_features`` views the model hands out are): returns the covering view."""
    parts = list(parts)
    if len(parts) == 1:
        return parts[0]
    p0 = parts[0]
    if p0.dim() >= 1 and p0.shape[0] > 0:
        st, es, ptr, rows, ok = p0.stride(), p0.element_size(), p0.data_ptr(), 0, True
        for p in parts:
            if (p.dtype != p0.dtype or p.device != p0.device or p.shape[1:] != p0.shape[1:] or p.shape[0] == 0 or p.stride() != st
                    or p.untyped_storage().data_ptr() != p0.untyped_storage().data_ptr() or p.data_ptr() != ptr + rows * st[0] * es):
                ok = False
                break
            rows += p.shape[0]
        if ok:
            return p0.as_strided((rows,) + tuple(p0.shape[1:]), st, p0.storage_offset())
    return torch.cat(parts, 0)


# ------------------------------------------------------------------------------ RoIAlign
def roi_align(feats: Sequence[torch.Tensor], rois: torch.Tensor, scales: Sequence[float], pooled: int,
              sampling_ratio: int = 2, k_min: int = 2, levels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """feats: 4 NHWC maps (fp32 or fp16); rois fp32 [K,5] (batch_idx,x1,y1,x2,y2) -> NHWC [K,P,P,C]."""
    feats = [_req_fp(f) for f in feats]
    dt = feats[0].dtype
    rois = _req(rois, name="rois")
    k = rois.shape[0]
    c = feats[0].shape[-1]
    out = torch.empty((k, pooled, pooled, c), dtype=dt, device=rois.device)
    if k == 0:
        return out
    hw = (C.c_int * 8)(*[d for f in feats for d in (f.shape[1], f.shape[2])])
    if levels is not None:
        levels = _req(levels, torch.int32, "levels")
    fn = getattr(_native.lib(), "seam_roi_align_" + _sfx(dt))
    _native.check(fn(_ptr(feats[0]), _ptr(feats[1]), _ptr(feats[2]), _ptr(feats[3]), hw, c, scales[0], scales[1],
                     scales[2], scales[3], k_min, _ptr(rois), _ptr(levels), _ptr(out), k, pooled, sampling_ratio,
                     _stream()), "seam_roi_align")
    return out


# ------------------------------------------------------------------------------ SEAM heads
@dataclass
class PackedNLB:
    w_proj_t: torch.Tensor   # [256,384]
    b_proj: torch.Tensor     # [384]
    w_cat: torch.Tensor      # [256]
    w_out_t: torch.Tensor    # [128,256]
    b_out: torch.Tensor      # [256]
    w_att: torch.Tensor      # [256]
    b_att: torch.Tensor      # [1]


NLB_MFMA = _os.environ.get("SEAM_NLB_MFMA", "1") != "0"      # 0: the VALU form of the block for every sequence length


def _nlb_mfma_pack(pk: PackedNLB):
    """Weights of ``seam_nlb_attnpool_mfma_f32`` derived (once per PackedNLB) from the plain pack:
    MFMA B-fragment order of the g and W projections -- element e of lane (h = lane >> 5, n = lane & 31) of fragment
    [n_tile][k / 8] is W[k = 8 j + 4 h + e][32 n_tile + n] -- and the folded theta / phi projections: theta and phi enter the
    block only through a_i = theta_i . wc[:128], b_j = phi_j . wc[128:] (ref models/nlb.py:80-90), i.e. a_i = X_i . u + c with
    u = W_theta^T wc[:128] (folded in fp64, rounded once)."""
    m = getattr(pk, "_mfma", None)
    if m is None:
        wp, bp, wc = pk.w_proj_t.double(), pk.b_proj.double(), pk.w_cat.double()       # [256,384], [384], [256]
        u = (wp[:, :128] @ wc[:128]).float().contiguous()
        v = (wp[:, 128:256] @ wc[128:]).float().contiguous()
        cd = torch.stack([bp[:128] @ wc[:128], bp[128:256] @ wc[128:]]).float().contiguous()
        wg = pk.w_proj_t[:, 256:]                                                        # [256 k, 128 n]
        wg_frag = wg.reshape(32, 2, 4, 4, 32).permute(3, 0, 1, 4, 2).contiguous()       # [nt, j, h, n, e]
        wo_frag = pk.w_out_t.reshape(16, 2, 4, 8, 32).permute(3, 0, 1, 4, 2).contiguous()   # [128 k, 256 n] -> [nt, j, h, n, e]
        m = pk._mfma = (wg_frag, pk.b_proj[256:].contiguous(), u, v, cd, wo_frag)
    return m


def nlb_attnpool(seq: torch.Tensor, t_stride: int, s_stride: int, lens: torch.Tensor, n_seq: int, t_max: int,
                 pk: PackedNLB, use_nlb: bool = True, want_att: bool = False, want_z: bool = False):
    """Batched non-local block + attention pooling.
    Returns (out[S,256], att[S,Tmax] | None) and additionally z[S,Tmax,256] when want_z.
    Sequences of up to ``seam_nlb_mfma_max_len()`` (96) rows run the block's GEMMs on the matrix cores
    (``seam_nlb_attnpool_mfma_f32``); longer ones take the VALU kernel with its global scratch."""
    lib = _native.lib()
    seq = _req(seq, name="seq")
    lens = _req(lens, torch.int32, "lens")
    out = torch.empty((n_seq, 256), dtype=F32, device=seq.device)
    att = torch.zeros((n_seq, t_max), dtype=F32, device=seq.device) if want_att else None
    z = torch.zeros((n_seq, t_max, 256), dtype=F32, device=seq.device) if want_z else None
    if n_seq == 0:
        return (out, att, z) if want_z else (out, att)
    if (NLB_MFMA and t_max <= int(lib.seam_nlb_mfma_max_len()) and t_stride % 4 == 0 and s_stride % 4 == 0
            and seq.data_ptr() % 16 == 0):
        wg_frag, b_g, u, v, cd, wo_frag = _nlb_mfma_pack(pk)
        _native.check(lib.seam_nlb_attnpool_mfma_f32(_ptr(seq), t_stride, s_stride, _ptr(lens), n_seq, t_max, _ptr(wg_frag), _ptr(b_g),
                                                     _ptr(u), _ptr(v), _ptr(cd), _ptr(wo_frag), _ptr(pk.b_out), _ptr(pk.w_att),
                                                     _ptr(pk.b_att), _ptr(out), _ptr(att), _ptr(z), int(use_nlb), _stream()),
                      "seam_nlb_attnpool_mfma_f32")
        return (out, att, z) if want_z else (out, att)
    ws = torch.empty((int(lib.seam_nlb_workspace_floats(n_seq, t_max)),), dtype=F32, device=seq.device)
    _native.check(lib.seam_nlb_attnpool_f32(_ptr(seq), t_stride, s_stride, _ptr(lens), n_seq, t_max, _ptr(pk.w_proj_t),
                                            _ptr(pk.b_proj), _ptr(pk.w_cat), _ptr(pk.w_out_t), _ptr(pk.b_out),
                                            _ptr(pk.w_att), _ptr(pk.b_att), _ptr(out), _ptr(att), _ptr(z), _ptr(ws),
                                            int(use_nlb), _stream()), "seam_nlb_attnpool_f32")
    return (out, att, z) if want_z else (out, att)


def pair_logits(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """a[Q,D], b[G,D], w[2,D], bias[2] -> [Q,G,2]."""
    a, b, w, bias = _req(a), _req(b), _req(w.detach()), _req(bias.detach())
    q, g, d = a.shape[0], b.shape[0], a.shape[1]
    out = torch.empty((q, g, 2), dtype=F32, device=a.device)
    _native.check(_native.lib().seam_pair_logits_f32(_ptr(a), _ptr(b), _ptr(w), _ptr(bias), _ptr(out), q, g, d,
                                                     _stream()), "seam_pair_logits_f32")
    return out


def rank_topk(logits: torch.Tensor, k: int):
    """logits[Q,G,2] -> (idx int64 [Q,k], score [Q,k]) by descending softmax(x)[...,1]."""
    logits = _req(logits)
    q, g = logits.shape[0], logits.shape[1]
    k = min(k, g)
    idx = torch.empty((q, k), dtype=torch.int64, device=logits.device)
    sc = torch.empty((q, k), dtype=F32, device=logits.device)
    _native.check(_native.lib().seam_rank_topk_f32(_ptr(logits), _ptr(idx), _ptr(sc), q, g, k, _stream()),
                  "seam_rank_topk_f32")
    return idx, sc


def match_scores(logits: torch.Tensor) -> torch.Tensor:
    """logits [...,2] -> softmax(logits)[...,1]."""
    logits = _req(logits)
    out = torch.empty(logits.shape[:-1], dtype=F32, device=logits.device)
    _native.check(_native.lib().seam_match_scores_f32(_ptr(logits), _ptr(out), out.numel(), _stream()),
                  "seam_match_scores_f32")
    return out


def pair_scores_blockdiag(x: torch.Tensor, seg, w: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """Self-similarity of every group of rows of ``x`` [rows, D] (groups = consecutive row ranges, boundaries ``seg`` [n_seg + 1]):
    the flat concatenation of the n_s x n_s score blocks, softmax(W (x_i - x_j)^2 + b)[1].  One launch for all groups
    (``seam_pair_scores_blockdiag_f32``); bit-identical to ``match_scores(pair_logits(x_s, x_s))`` per group
    (ref evaluate_movingfashion.py:102-121 ``compute_selfdist``)."""
    import numpy as np
    x = _req(x, name="x")
    seg = np.asarray(seg, dtype=np.int64)
    n = np.diff(seg)
    if len(n) == 0 or int(n.max()) == 0:
        return torch.empty((0,), dtype=F32, device=x.device)
    if seg[0] != 0 or int(seg[-1]) > x.shape[0] or (n < 0).any():
        raise ValueError("pair_scores_blockdiag: seg must be non-decreasing row offsets inside x, starting at 0")
    off = np.concatenate([[0], np.cumsum(n * n)])
    out = torch.empty((int(off[-1]),), dtype=F32, device=x.device)
    seg_d = torch.as_tensor(seg, dtype=torch.int32, device=x.device)
    off_d = torch.as_tensor(off, dtype=torch.int64, device=x.device)
    _native.check(_native.lib().seam_pair_scores_blockdiag_f32(_ptr(x), _ptr(seg_d), _ptr(off_d), _ptr(_req(w, name="w")), _ptr(_req(bias, name="bias")),
                                                               _ptr(out), len(n), int(n.max()), x.shape[1], _stream()),
                  "seam_pair_scores_blockdiag_f32")
    return out


def rank_of(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """logits [Q,G,2], target int64 [Q] -> rank int64 [Q] of the target product in each query's ranking."""
    logits, target = _req(logits), _req(target, torch.int64, "target")
    q, g = logits.shape[0], logits.shape[1]
    out = torch.empty((q,), dtype=torch.int64, device=logits.device)
    _native.check(_native.lib().seam_rank_of_f32(_ptr(logits), _ptr(target), _ptr(out), q, g, _stream()), "seam_rank_of_f32")
    return out


def score_reduce(score: torch.Tensor, mode: str) -> torch.Tensor:
    """score [n,G] -> [G]: column mean (``"mean"``) or max (``"max"``)."""
    score = _req(score)
    n, g = score.shape
    out = torch.empty((g,), dtype=F32, device=score.device)
    _native.check(_native.lib().seam_score_reduce_f32(_ptr(score), _ptr(out), n, g, {"mean": 0, "max": 1}[mode], _stream()),
                  "seam_score_reduce_f32")
    return out


def score_reduce_segments(score: torch.Tensor, seg: torch.Tensor, mode: str) -> torch.Tensor:
    """score [N,G], seg [P+1] int32 row offsets (device) -> [P,G]: ``score_reduce`` of the rows seg[p]:seg[p+1] of every segment in
    ONE launch (the same sequential row order per column: bit-identical to P separate calls)."""
    score = _req(score)
    seg = _req(seg, torch.int32, "seg")
    n, g = score.shape
    p = seg.numel() - 1
    out = torch.empty((p, g), dtype=F32, device=score.device)
    _native.check(_native.lib().seam_score_reduce_seg_f32(_ptr(score), _ptr(seg), _ptr(out), p, g, {"mean": 0, "max": 1}[mode], _stream()),
                  "seam_score_reduce_seg_f32")
    return out


def rank_of_scores(score: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """score [Q,G], target int64 [Q] -> rank int64 [Q] of the target in each row's descending order."""
    score, target = _req(score), _req(target, torch.int64, "target")
    q, g = score.shape
    out = torch.empty((q,), dtype=torch.int64, device=score.device)
    _native.check(_native.lib().seam_rank_of_scores_f32(_ptr(score), _ptr(target), _ptr(out), q, g, _stream()),
                  "seam_rank_of_scores_f32")
    return out


def box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """xyxy boxes a [Na,4], b [Nb,4] -> IoU [Na,Nb]."""
    a, b = _req(a), _req(b)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=F32, device=a.device)
    _native.check(_native.lib().seam_box_iou_f32(_ptr(a), _ptr(b), _ptr(out), a.shape[0], b.shape[0], _stream()),
                  "seam_box_iou_f32")
    return out


PAIR_MFMA = True       # large banks: candidates on the fp32 matrix cores (seam_pair_topk_mfma_f32); False: the chunked VALU path


def pair_topk(a: torch.Tensor, b: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, k: int, fused: bool = False,
              q_chunk: int = 64, mfma: Optional[bool] = None, stats: Optional[torch.Tensor] = None, force_exact: bool = False):
    """a13+a14 without materialising the full [Q,G,2] logits -> (idx int64 [Q,k], score [Q,k]).  Three implementations with
    bit-identical results:

    * ``mfma`` (default when the bank has >= seam_pair_topk_mfma_min_gallery() rows, D == 256, k <= 64): seam_pair_topk_mfma_f32 --
      the logit-difference GEMM on the fp32 matrix cores finds candidates, the winners are re-scored with the direct form and a
      per-query error bound proves the top k (``stats``: optional int32[4] device tensor that receives [queries redone with the
      direct form, largest candidate list, overflowed lists, 0]; ``force_exact``: test hook, every query takes the direct form);
    * query chunks of ``q_chunk`` through seam_pair_logits_f32 + seam_rank_topk_f32 with one reused logits buffer (small banks);
    * ``fused=True``: seam_pair_topk_f32, the single-pass VALU kernel."""
    lib = _native.lib()
    a, b, w, bias = _req(a), _req(b), _req(w.detach()), _req(bias.detach())
    q, g, d = a.shape[0], b.shape[0], a.shape[1]
    k = min(k, g)
    idx = torch.empty((q, k), dtype=torch.int64, device=a.device)
    sc = torch.empty((q, k), dtype=F32, device=a.device)
    can_mfma = d == 256 and g >= lib.seam_pair_topk_mfma_min_gallery() and 0 < k <= lib.seam_pair_topk_mfma_max_k() and q > 0
    if mfma is None:
        mfma = PAIR_MFMA and can_mfma and not fused
    if mfma:
        if not can_mfma:
            raise ValueError(f"pair_topk(mfma=True) needs D == 256, G >= {lib.seam_pair_topk_mfma_min_gallery()} and "
                             f"k <= {lib.seam_pair_topk_mfma_max_k()} (got D={d}, G={g}, k={k})")
        ws = torch.empty((int(lib.seam_pair_topk_mfma_workspace_floats(q, g, k)),), dtype=F32, device=a.device)
        if stats is not None:
            stats = _req(stats, torch.int32, "stats")
        _native.check(lib.seam_pair_topk_mfma_f32(_ptr(a), _ptr(b), _ptr(w), _ptr(bias), _ptr(idx), _ptr(sc), q, g, d, k, _ptr(ws),
                                                  1 if force_exact else 0, _ptr(stats) if stats is not None else None, _stream()),
                      "seam_pair_topk_mfma_f32")
        return idx, sc
    if fused:
        ws = torch.empty((int(lib.seam_pair_topk_workspace_floats(q, g, k)),), dtype=F32, device=a.device)
        _native.check(lib.seam_pair_topk_f32(_ptr(a), _ptr(b), _ptr(w), _ptr(bias), _ptr(idx), _ptr(sc), q, g, d, k,
                                             _ptr(ws), _stream()), "seam_pair_topk_f32")
        return idx, sc
    buf = torch.empty((min(q, q_chunk), g, 2), dtype=F32, device=a.device)
    for s0 in range(0, q, q_chunk):
        n = min(q_chunk, q - s0)
        _native.check(lib.seam_pair_logits_f32(C.c_void_p(a[s0:].data_ptr()), _ptr(b), _ptr(w), _ptr(bias), _ptr(buf), n, g, d,
                                               _stream()), "seam_pair_logits_f32")
        _native.check(lib.seam_rank_topk_f32(_ptr(buf), C.c_void_p(idx[s0:].data_ptr()), C.c_void_p(sc[s0:].data_ptr()), n, g,
                                             k, _stream()), "seam_rank_topk_f32")
    return idx, sc


# ------------------------------------------------------------------------------ detection
def decode_boxes(deltas: torch.Tensor, boxes: torch.Tensor, weights, clip_hw=None) -> torch.Tensor:
    """deltas [N,ncls*4], boxes [N,4] -> [N,ncls*4]; optional clip to (h,w)."""
    deltas, boxes = _req(deltas), _req(boxes)
    n = boxes.shape[0]
    ncls = deltas.shape[1] // 4
    out = torch.empty_like(deltas)
    ch, cw = (float(clip_hw[0]), float(clip_hw[1])) if clip_hw is not None else (0.0, 0.0)
    _native.check(_native.lib().seam_decode_boxes_f32(_ptr(deltas), _ptr(boxes), _ptr(out), n, ncls, *map(float, weights),
                                                      ch, cw, _stream()), "seam_decode_boxes_f32")
    return out


def nms_sorted(boxes: torch.Tensor, thr: float, max_keep: int = 0) -> torch.Tensor:
    """boxes [N,4] or [B,N,4], already sorted by descending score (per image) -> keep mask int32 [N] / [B,N].
    max_keep > 0: only the first ``max_keep`` survivors of each image are marked (== the full result truncated; the scan stops
    there)."""
    boxes = _req(boxes)
    single = boxes.dim() == 2
    b3 = boxes[None] if single else boxes
    bsz, n = b3.shape[0], b3.shape[1]
    keep = torch.empty((bsz, n), dtype=torch.int32, device=boxes.device)
    if n and bsz:
        nb = (n + 63) // 64
        ws = torch.empty((bsz * n * nb,), dtype=torch.int64, device=boxes.device)
        _native.check(_native.lib().seam_nms_sorted_topn_f32(_ptr(b3), _ptr(keep), bsz, n, float(thr), int(max_keep), _ptr(ws),
                                                             _stream()), "seam_nms_sorted_topn_f32")
    return keep[0] if single else keep


def rpn_topk_max() -> int:
    return int(_native.lib().seam_rpn_topk_max())


def rpn_topk_decode(head: torch.Tensor, num_anchors: int, anchors: torch.Tensor, clip_hw: torch.Tensor, k: int,
                    boxes: torch.Tensor, scores: torch.Tensor, offset: int, index: Optional[torch.Tensor] = None) -> None:
    """One pyramid level of RPN ``filter_proposals`` [TV] in one launch (``seam_rpn_topk_decode_f32``).

    head    [N,H,W,A+4A] fp32: the fused RPN head output (objectness logits | deltas per pixel), read in place
    anchors [H*W*A,4]; clip_hw [N,2] (height, width of each resized image), both fp32 on the device
    Writes, for every image, the top-``k`` anchors by logit (descending; ties lowest anchor index first) as decoded + clipped
    boxes into ``boxes[:, offset:offset+k]`` ([N,Ktot,4]) and sigmoid scores into ``scores[:, offset:offset+k]`` ([N,Ktot]);
    ``index`` (int64 [N,Ktot], optional) receives the anchor indices."""
    lib = _native.lib()
    head, anchors, clip_hw = _req(head, name="head"), _req(anchors, name="anchors"), _req(clip_hw, name="clip_hw")
    n_img, h, w, c = head.shape
    a = int(num_anchors)
    if c != 5 * a or anchors.shape != (h * w * a, 4) or clip_hw.shape != (n_img, 2):
        raise ValueError("rpn_topk_decode: head must be [N,H,W,5A], anchors [H*W*A,4], clip_hw [N,2]")
    if (boxes.dtype != F32 or scores.dtype != F32 or not boxes.is_cuda or not boxes.is_contiguous() or not scores.is_contiguous()
            or boxes.dim() != 3 or boxes.shape[0] != n_img or boxes.shape[2] != 4 or tuple(scores.shape) != tuple(boxes.shape[:2])
            or offset < 0 or offset + k > boxes.shape[1]):
        raise ValueError("rpn_topk_decode: boxes [N,Ktot,4] / scores [N,Ktot] fp32 contiguous with room for k rows at offset")
    if index is not None and (index.dtype != torch.int64 or tuple(index.shape) != tuple(scores.shape) or not index.is_contiguous()):
        raise ValueError("rpn_topk_decode: index must be int64 [N,Ktot] contiguous")
    if k < 1 or k > h * w * a or k > int(lib.seam_rpn_topk_max()):
        raise ValueError(f"rpn_topk_decode: k = {k} outside 1..min(n, {int(lib.seam_rpn_topk_max())})")
    obj = C.c_void_p(head.data_ptr())
    dlt = C.c_void_p(head.data_ptr() + 4 * a)
    _native.check(lib.seam_rpn_topk_decode_f32(obj, dlt, _ptr(anchors), _ptr(clip_hw), _ptr(boxes), _ptr(scores), _ptr(index),
                                               n_img, h * w * a, a, k, h * w * c, c, h * w * c, c, boxes.shape[1], offset,
                                               _stream()), "seam_rpn_topk_decode_f32")


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, hw) -> torch.Tensor:
    """masks [K,1,28,28] prob, boxes [K,4] original-image px -> [K,1,H,W]."""
    masks, boxes = _req(masks), _req(boxes)
    k = masks.shape[0]
    out = torch.empty((k, 1, int(hw[0]), int(hw[1])), dtype=F32, device=masks.device)
    if k:
        _native.check(_native.lib().seam_paste_masks_f32(_ptr(masks), _ptr(boxes), _ptr(out), k, int(hw[0]), int(hw[1]),
                                                         _stream()), "seam_paste_masks_f32")
    return out


def mask_select(logits: torch.Tensor, labels: torch.Tensor, ncls: int) -> torch.Tensor:
    """logits [K,14,14,4*ncls] (sub-pixel groups) -> sigmoid prob of channel labels[k]: [K,1,28,28]."""
    logits = _req_fp(logits)
    labels = _req(labels, torch.int64, "labels")
    k = logits.shape[0]
    out = torch.empty((k, 1, 28, 28), dtype=F32, device=logits.device)
    fn = getattr(_native.lib(), "seam_mask_select_" + _sfx(logits.dtype))
    _native.check(fn(_ptr(logits), _ptr(labels), _ptr(out), k, ncls, _stream()), "seam_mask_select")
    return out


# ------------------------------------------------------------------------------ input pipeline (row f4)
def frame_noise(bgr: torch.Tensor, sigma: float, noise: Optional[torch.Tensor] = None, seed: int = 0) -> torch.Tensor:
    """uint8 [H,W,3] BGR -> uint8 [H,W,3] RGB with additive noise (float64 maths, ref datasets/MFDataset.py:81-88).
    noise: float64 [H,W,3] standard-normal draws, or None for on-device counter-based draws keyed by ``seed``;
    sigma == 0 and noise None: channel flip only (the dataset's noise=False branch)."""
    bgr = _req(bgr, torch.uint8, "bgr")
    if bgr.dim() != 3 or bgr.shape[2] != 3:
        raise ValueError("frame must be uint8 [H,W,3]")
    if noise is not None:
        noise = _req(noise, torch.float64, "noise")
        if noise.shape != bgr.shape:
            raise ValueError("noise must have the frame's shape")
    out = torch.empty_like(bgr)
    _native.check(_native.lib().seam_frame_noise_u8(_ptr(bgr), _ptr(noise), _ptr(out), bgr.shape[0], bgr.shape[1], float(sigma),
                                                    int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()), "seam_frame_noise_u8")
    return out


def resize_bicubic_u8(img: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """``PIL.Image.resize((out_w, out_h))`` (BICUBIC, antialiased, 8-bit fixed point) on a uint8 [H,W,3] device image."""
    lib = _native.lib()
    img = _req(img, torch.uint8, "img")
    h, w = img.shape[0], img.shape[1]
    out = torch.empty((out_h, out_w, 3), dtype=torch.uint8, device=img.device)
    ws = torch.empty((int(lib.seam_resize_workspace_bytes(h, w, out_h, out_w)),), dtype=torch.uint8, device=img.device)
    _native.check(lib.seam_resize_bicubic_u8(_ptr(img), _ptr(out), h, w, out_h, out_w, _ptr(ws), _stream()), "seam_resize_bicubic_u8")
    return out
