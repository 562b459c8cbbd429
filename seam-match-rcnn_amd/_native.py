"""ctypes binding of ``lib/libseam_hip.so`` (the C ABI declared in ``include/seam_hip.h``).

There is NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using the product path raises.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C seam-match-rcnn_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP runtime (soname libamdhip64.so.7).  It MUST be in the process before
# libseam_hip.so is dlopen'ed, so that the library binds to the same runtime instance torch uses
# (loading /opt/rocm's copy first gives two runtimes and "no ROCm-capable device" at launch).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SEAM_LIB_PATH") or os.path.join(_HERE, "lib", "libseam_hip.so")   # env: kernel experiments only

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_i64 = C.c_int64

class TrunkLayer(C.Structure):
    """seam_trunk_layer_t of include/seam_hip.h."""
    _fields_ = [("w", C.c_void_p), ("u", C.c_void_p), ("u24", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/seam_hip.h one to one
SIGNATURES = {
    "seam_version": (_i, []),
    "seam_error_string": (C.c_char_p, [_i]),
    "seam_option_count": (_i, []),
    "seam_option_name": (C.c_char_p, [_i]),
    "seam_set_option": (_i, [C.c_char_p, _i]),
    "seam_get_option": (_i, [C.c_char_p, C.POINTER(_i)]),
    "seam_conv_kred": (_i, [_i, _i, _i]),
    "seam_conv_rows_padded": (_i, [_i]),
    "seam_conv_tile": (_i, [_i, _i]),
    "seam_conv_tile_prec": (_i, [_i, _i, _i]),
    "seam_conv_tile_taps": (_i, [_i, _i, _i, _i]),
    "seam_pack_conv_weight_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv2d_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_linear_narrow_supported": (_i, [_i, _i]),
    "seam_pack_linear_narrow_f32": (_i, [_p, _p, _i, _i, _p]),
    "seam_linear_narrow_f32": (_i, [_p, _p, _p, _p, C.c_longlong, _i, _i, _i, _p]),
    "seam_conv2d_crop_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv2d_crop_f16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_s2d_batch_f32": (_i, [_p, C.c_size_t, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_s2d_batch_f16": (_i, [_p, C.c_size_t, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_s2d_pad_batch_f16": (_i, [_p, C.c_size_t, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv2d_dual_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv1x1_pc_supported": (_i, [C.c_longlong, _i, _i]),
    "seam_conv1x1_pc_weight_floats": (C.c_longlong, [_i, _i]),
    "seam_pack_conv1x1_pc_f32": (_i, [_p, _p, _i, _i, _p]),
    "seam_conv1x1_pc_f32": (_i, [_p, _p, _p, _p, _p, _p, C.c_longlong, _i, _i, _i, _p]),
    "seam_conv1x1_sw_config": (_i, [_i, _i, _i, _i]),
    "seam_conv1x1_sw_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv1x1_swh_config": (_i, [C.c_longlong, _i, _i, _i]),
    "seam_conv1x1_f16pc_supported": (_i, [C.c_longlong, _i, _i]),
    "seam_conv1x1_f16pc_weight_halves": (C.c_longlong, [_i, _i]),
    "seam_pack_conv1x1_weight_f16pc": (_i, [_p, _p, _i, _i, _p]),
    "seam_conv1x1_f16pc": (_i, [_p, _p, _p, _p, _p, _p, C.c_longlong, _i, _i, _i, _p]),
    "seam_conv1x1_swh_f16": (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_longlong, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_stem_s2d_swh_f16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "seam_conv2d_dual_f16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv2d_upres_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_pack_conv_weight_bx3": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv2d_bx3": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_wino_supported": (_i, [_i, _i, _i, _i, _i]),
    "seam_wino_weight_floats": (C.c_longlong, [_i, _i]),
    "seam_wino_slot_fill_pct": (_i, [_i, _i, _i, _i, _i, _i]),
    "seam_wino_tile_variant": (_i, [_i, _i, _i, _i, _i, _i]),
    "seam_pack_conv_weight_wino_f32": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "seam_wino_issue_slots": (C.c_longlong, [_i, _i, _i, _i, _i, _i]),
    "seam_wino24_issue_slots": (C.c_longlong, [_i, _i, _i, _i, _i, _i]),
    "seam_wino24_variant": (_i, [_i, _i, _i, _i, _i, _i]),
    "seam_wino24_form": (_i, [_i, _i, _i, _i, _i, _i]),
    "seam_wino24_weight_floats": (C.c_longlong, [_i, _i]),
    "seam_pack_conv_weight_wino24_f32": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "seam_conv3x3_wino24_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv3x3_wino_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv3x3_f16pc_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "seam_conv3x3_f16pc_pays": (_i, [_i, _i, _i, _i, _i, _i]),
    "seam_f16pc_weight_halves": (C.c_longlong, [_i, _i]),
    "seam_pack_conv_weight_f16pc": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_conv3x3_f16pc": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv_kred_f16": (_i, [_i, _i, _i]),
    "seam_pack_conv_weight_f16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_conv2d_f16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_batch_f32": (_i, [_p, C.c_size_t, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_batch_f16": (_i, [_p, C.c_size_t, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_u8": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_preprocess_f16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_maxpool2d_f16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_upsample_add_f16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_roi_align_f16": (_i, [_p, _p, _p, _p, C.POINTER(_i), _i, _f, _f, _f, _f, _i, _p, _p, _p, _i, _i, _i, _p]),
    "seam_roi_align_set_lds": (None, [_i]),
    "seam_nchw_f32_to_nhwc_f16": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_nhwc_f16_to_nchw_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_avgpool_f16": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_mask_select_f16": (_i, [_p, _p, _p, _i, _i, _p]),
    "seam_maxpool2d_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "seam_upsample_add_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "seam_roi_align_f32": (_i, [_p, _p, _p, _p, C.POINTER(_i), _i, _f, _f, _f, _f, _i, _p, _p, _p, _i, _i, _i, _p]),
    "seam_nchw_to_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_nhwc_to_nchw_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_avgpool_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_nlb_workspace_floats": (_i64, [_i, _i]),
    "seam_nlb_attnpool_f32": (_i, [_p, _i64, _i64, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "seam_nlb_mfma_max_len": (_i, []),
    "seam_nlb_attnpool_mfma_f32": (_i, [_p, _i64, _i64, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "seam_pair_logits_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "seam_rank_topk_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "seam_match_scores_f32": (_i, [_p, _p, _i64, _p]),
    "seam_host_build_tracklets": (_i, [_p, _p, _p, _p, _i, C.c_double, _p, _p, _p]),
    "seam_pair_scores_blockdiag_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "seam_rank_of_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "seam_score_reduce_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "seam_score_reduce_seg_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "seam_rank_of_scores_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "seam_box_iou_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "seam_pair_topk_workspace_floats": (_i64, [_i, _i, _i]),
    "seam_pair_topk_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "seam_match_trunk_workspace_floats": (_i64, [_i]),
    "seam_match_trunk_f32": (_i, [_p, C.POINTER(TrunkLayer), C.POINTER(TrunkLayer), _p, _i, _p, C.POINTER(_i), _p]),
    "seam_pair_topk_mfma_min_gallery": (_i, []),
    "seam_pair_topk_mfma_max_k": (_i, []),
    "seam_pair_topk_mfma_workspace_floats": (_i64, [_i, _i, _i]),
    "seam_pair_topk_mfma_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "seam_decode_boxes_f32": (_i, [_p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _p]),
    "seam_nms_sorted_f32": (_i, [_p, _p, _i, _i, _f, _p, _p]),
    "seam_nms_sorted_topn_f32": (_i, [_p, _p, _i, _i, _f, _i, _p, _p]),
    "seam_rpn_topk_max": (_i, []),
    "seam_rpn_topk_decode_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i64, _i, _i64, _i, _i64, _i, _p]),
    "seam_paste_masks_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "seam_mask_select_f32": (_i, [_p, _p, _p, _i, _i, _p]),
    "seam_conv_wgrad_workspace_floats": (_i64, [_i, _i, _i, _i, _i]),
    "seam_conv_wgrad_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "seam_colsum_workspace_floats": (_i64, [_i, _i]),
    "seam_colsum_f32": (_i, [_p, _p, _i, _i, _p, _p]),
    "seam_avgpool_relu_bwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "seam_bn1d_train_fwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _p]),
    "seam_bn1d_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "seam_frame_noise_u8": (_i, [_p, _p, _p, _i, _i, C.c_double, C.c_uint64, _p]),
    "seam_resize_workspace_bytes": (_i64, [_i, _i, _i, _i]),
    "seam_resize_bicubic_u8": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "seam_ce2_fwd_bwd_f32": (_i, [_p, _p, _p, _p, _p, _i64, _p]),
    "seam_pair_logits_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "seam_nlb_bwd_workspace_floats": (_i64, [_i, _i]),
    "seam_nlb_attnpool_bwd_f32": (_i, [_p, _i64, _i64, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, C.POINTER(_p), _p, _i, _p]),
    "seam_nlb_block_bwd_f32": (_i, [_p, _i64, _i64, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, C.POINTER(_p), _p, _i, _p]),
}

_lib = None


class SeamNativeError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the bound library; raise loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SeamNativeError(
            f"{LIB_PATH} not found: the HIP extension is required (there is no CPU/PyTorch fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:   # e.g. libamdhip64 missing
        raise SeamNativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise SeamNativeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _apply_env_options(handle)
    _lib = handle
    return _lib


def _apply_env_options(handle) -> None:
    """The launchers' variant selectors (csrc/seam_opts.h) are plain ints behind ``seam_set_option``: the library itself never
    reads the environment.  For the parity tests and the A/B tools, SEAM_* environment variables of the same names are applied
    here, once, when the library is loaded (``SEAM_CONV_TILE=256x128`` is passed as 256128)."""
    for i in range(handle.seam_option_count()):
        name = handle.seam_option_name(i)
        raw = os.environ.get(name.decode())
        if raw is None or raw == "":
            continue
        if "x" in raw:
            bm, bn = raw.split("x")
            val = int(bm) * 1000 + int(bn)
        else:
            val = int(raw)
        check_rc = handle.seam_set_option(name, val)
        if check_rc != 0:
            raise SeamNativeError(f"seam_set_option({name!r}, {val}) failed: {check_rc}")


def set_option(name: str, value: int) -> None:
    check(lib().seam_set_option(name.encode(), int(value)), f"seam_set_option({name})")


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib().seam_get_option(name.encode(), C.byref(v)), f"seam_get_option({name})")
    return v.value


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().seam_error_string(code)
        raise SeamNativeError(f"{what} failed: hipError {code} ({msg.decode() if msg else '?'})")
