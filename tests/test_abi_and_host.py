"""CPU: the C-ABI library loads and exports every symbol include/seam_hip.h declares (no compute
calls), and the host-side logic (sizes, anchors, sequence packing plan, sharding, state-dict
keys) agrees with the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import ROOT, to_torch
from oracle import detection as OD
from oracle import heads as OH


@pytest.fixture(scope="module")
def native():
    from seam_match_rcnn_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "seam-match-rcnn_amd", "csrc"), "-j4"], check=True)
    return _native


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "seam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(seam_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(native):
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = native.lib()
    for s in syms:
        assert hasattr(lib, s), f"libseam_hip.so does not export {s}"
        assert s in native.SIGNATURES, f"{s} declared in seam_hip.h but not bound in _native.SIGNATURES"
    assert sorted(native.SIGNATURES) == syms
    out = subprocess.run(["nm", "-D", "--defined-only", native.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (seam_[a-z0-9_]+)", out))
    assert exported == set(syms)


def test_host_helpers_no_gpu(native):
    lib = native.lib()
    assert lib.seam_version() >= 1000
    assert lib.seam_conv_kred(256, 3, 3) == 2304 and lib.seam_conv_kred(4, 7, 7) == 224
    assert lib.seam_conv_rows_padded(15) == 64 and lib.seam_conv_rows_padded(256) == 256
    assert lib.seam_nlb_workspace_floats(2, 10) >= 2 * 10 * 130
    # the weights-stationary pointwise kernel only takes output-channel counts whose slab count divides an XCD's 32 blocks (ADVICE r4:
    # K = 768 would have redone tiles, K = 8448 left channels unwritten / hung): other K stay on the implicit GEMM
    cfg = lambda c, k: lib.seam_conv1x1_sw_config(1 << 20, c, 0, k)
    assert cfg(64, 256) == 108 and cfg(128, 512) == 108 and cfg(256, 1024) == 204 and cfg(256, 64) == 202 and cfg(64, 2048) == 108
    assert cfg(64, 768) == 0 and cfg(64, 8448) == 0 and cfg(256, 768) == 0 and cfg(512, 256) == 0 and cfg(64, 96) == 0
    # layer shapes the launcher gives to the producer / consumer Winograd kernel (round 5) and those it must not
    assert lib.seam_wino24_form(80, 200, 200, 256, 256, 1) == 1 and lib.seam_wino24_form(80, 100, 100, 128, 128, 1) == 1
    assert lib.seam_wino24_form(80, 200, 200, 64, 64, 1) == 1 and lib.seam_wino24_form(80, 13, 13, 256, 256, 1) == 0
    assert lib.seam_wino24_form(80, 200, 200, 32, 64, 1) == 0 and lib.seam_wino24_form(80, 200, 200, 64, 32, 1) == 0
    assert lib.seam_wino24_form(2, 20, 20, 256, 256, 1) == 0 and lib.seam_wino24_form(1, 8, 8, 7, 32, 1) == -1
    # the producer / consumer pointwise kernel (round 5): long reductions (C >= 256 in pairs of 64-channel chunks), 128-channel output tiles
    pc = lib.seam_conv1x1_pc_supported
    assert pc(800000, 512, 256) == 1 and pc(1, 1024, 128) == 1 and pc(200000, 2048, 512) == 1 and pc(50000, 384, 2048) == 1
    assert pc(800000, 128, 256) == 0 and pc(800000, 320, 256) == 0 and pc(800000, 512, 192) == 0 and pc(0, 512, 256) == 0
    assert lib.seam_conv1x1_pc_weight_floats(256, 512) == 256 * 512
    # the fp16 producer / consumer 3x3 kernel: shapes it takes, and the dispatch rule (tiles >= 3/4 full, a function of the map only)
    sup, pays = lib.seam_conv3x3_f16pc_supported, lib.seam_conv3x3_f16pc_pays
    assert sup(48, 192, 336, 256, 256, 1) == 1 and sup(1536, 14, 14, 256, 256, 1) == 1 and sup(1536, 8, 8, 256, 1024, 0) == 1
    assert sup(48, 192, 336, 256, 192, 1) == 0 and sup(48, 20, 20, 256, 256, 1) == 0
    # round 6: C = K = 64 (layer1's 3x3 layers) on large maps; no other 64-channel combination, no small maps
    assert sup(48, 192, 336, 64, 64, 1) == 1 and sup(48, 192, 336, 64, 128, 1) == 0 and sup(48, 192, 336, 128, 64, 1) == 0 and sup(48, 14, 14, 64, 64, 1) == 0
    assert sup(48, 192, 336, 256, 256, 2) == 0 and sup(4, 2, 2, 256, 256, 0) == 0
    assert pays(48, 192, 336, 256, 256, 1) == 1 and pays(1, 192, 336, 256, 256, 1) == 1 and pays(1536, 14, 14, 256, 256, 0) == 1
    assert pays(48, 24, 42, 512, 512, 1) == 0 and pays(48, 192, 336, 64, 64, 1) == 1 and pays(48, 33, 33, 64, 64, 1) == 0     # 33 x 33 outputs in 3 x 3 tiles of 16 x 16
    assert lib.seam_f16pc_weight_halves(256, 256) == 256 * 256 * 9


def test_product_path_fails_loudly_without_gpu(native):
    from seam_match_rcnn_amd import ops
    from seam_match_rcnn_amd.models.match_head import MatchPredictor
    with pytest.raises(native.SeamNativeError):
        ops.conv2d(torch.zeros(1, 4, 4, 4), None)
    mp = MatchPredictor().eval()
    with torch.no_grad(), pytest.raises(native.SeamNativeError):
        mp(torch.zeros(2, 256, 14, 14), torch.IntTensor([0, 1]))


def test_cpu_images_raise_on_every_preprocess_branch():
    """CPU images -- including constant-stride views of one CPU clip, which satisfy the one-launch batch branch --
    must raise SeamNativeError before any pointer reaches a kernel (runs without a GPU: the check precedes every launch)."""
    import seam_match_rcnn_amd._native as native
    import seam_match_rcnn_amd.ops as ops
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    clip = torch.zeros(3, 3, 64, 96)
    for imgs, kw in ((list(clip.unbind(0)), dict(s2d=True)), (list(clip.unbind(0)), dict()), ([clip[0]], dict(s2d=True)),
                     ([torch.zeros(64, 96, 3, dtype=torch.uint8)], dict()), (list(clip.unbind(0)), dict(dtype=torch.float16))):
        with pytest.raises(native.SeamNativeError):
            ops.preprocess(imgs, [(64, 96)] * len(imgs), 64, 96, **kw)
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14).eval()
    with torch.no_grad(), pytest.raises(native.SeamNativeError):
        m(list(clip.unbind(0)))


def test_ctor_rejects_rpn_top_n_beyond_the_nms_capacity():
    from seam_match_rcnn_amd.models.detection import NMS_MAX_BOXES
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14, rpn_pre_nms_top_n_test=NMS_MAX_BOXES // 5)
    with pytest.raises(ValueError):
        videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14, rpn_pre_nms_top_n_test=NMS_MAX_BOXES // 5 + 1)


def test_cat_rows_is_zero_copy_for_adjacent_views():
    import seam_match_rcnn_amd.ops as ops
    x = torch.arange(10 * 4 * 3 * 3, dtype=torch.float32).view(10, 3, 3, 4).permute(0, 3, 1, 2)    # channels_last [10,4,3,3]
    parts = list(x.split([3, 2, 5]))
    y = ops.cat_rows(parts)
    assert y.data_ptr() == x.data_ptr() and y.stride() == x.stride() and torch.equal(y, x)
    z = ops.cat_rows([parts[0], parts[2]])                                                        # not adjacent: a real cat
    assert torch.equal(z, torch.cat([parts[0], parts[2]])) and z.data_ptr() != x.data_ptr()
    assert torch.equal(ops.cat_rows([parts[1], parts[0]]), torch.cat([parts[1], parts[0]]))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "seam-match-rcnn_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


@pytest.mark.parametrize("hw", [(800, 800), (1080, 1920), (480, 640), (333, 500), (1200, 700)])
def test_resized_size_matches_interpolate(hw):
    from seam_match_rcnn_amd.models.detection import resized_size
    h, w = hw
    oh, ow, scale = resized_size(h, w)
    ref = torch.nn.functional.interpolate(torch.zeros(1, 1, h, w), scale_factor=scale, mode="bilinear",
                                          recompute_scale_factor=True, align_corners=False)
    assert (oh, ow) == tuple(ref.shape[-2:]) == OD.resized_size(h, w)[:2]


def test_grid_anchors_match_oracle_and_known_values():
    from seam_match_rcnn_amd.models.detection import grid_anchors
    feat = [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]
    got = grid_anchors((800, 800), feat)
    ref = OD.grid_anchors((800, 800), feat)
    assert sum(a.shape[0] for a in got) == 159882
    for g, r in zip(got, ref):
        assert np.array_equal(g, r.numpy())
    # level 0, cell (0,0): ratios 0.5, 1, 2 of size 32 -> rounded half extents
    assert got[0][:3].tolist() == [[-23., -11., 23., 11.], [-16., -16., 16., 16.], [-11., -23., 11., 23.]]
    # 'pool' level stride is 800 // 13 = 61 (integer division quirk)
    assert got[4][3].tolist() == [61. - 362., -181., 61. + 362., 181.]


def test_sequence_plan_matches_reference_packing(golden):
    from seam_match_rcnn_amd.models.match_head import plan_sequences
    ids = torch.from_numpy(golden["ta_ids"])
    types = torch.from_numpy(golden["ta_types"])
    sel0, order, pos, seq_of_row, counts = plan_sequences(types, ids)
    x3_1 = torch.arange(len(sel0), dtype=torch.float32)[:, None].repeat(1, 4)       # row tags
    seq, mask, lst = OH.pack_sequences(x3_1, ids[sel0])
    mine = torch.zeros_like(seq)
    mine[pos + 1, seq_of_row] = x3_1[order]
    assert torch.equal(mine, seq)
    assert torch.equal((torch.arange(seq.shape[0])[None, :] > counts[:, None]), mask)
    assert np.array_equal(mask.numpy(), golden["taA_x3_1_mask"])
    assert np.array_equal(ids[sel0].numpy(), golden["taA_x3_1_ids"])


def test_shard_helpers():
    from seam_match_rcnn_amd.retrieval import clips_for_rank, shard_range
    for n, w in [(1000, 1), (1000, 8), (50000, 8), (10, 4), (3, 8)]:
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
        assert max(hi - lo for lo, hi in rs) - min(hi - lo for lo, hi in rs) <= 1
    assert sorted(sum((clips_for_rank(64, r, 8) for r in range(8)), [])) == list(range(64))
    assert clips_for_rank(64, 3, 8)[:3] == [3, 11, 19]


def test_state_dict_keys_and_both_torchvision_layouts():
    from seam_match_rcnn_amd.models.matchrcnn import matchrcnn_resnet50_fpn
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd, strict=True)
    new = {}
    for k, v in sd.items():
        k2 = k
        if k.startswith("backbone.fpn."):
            p = k.split(".")
            k2 = ".".join(p[:4] + ["0"] + p[4:])
        elif k.startswith("rpn.head.conv."):
            k2 = k.replace("rpn.head.conv.", "rpn.head.conv.0.0.")
        elif "mask_head.mask_fcn" in k:
            i = int(k.split("mask_fcn")[1][0])
            k2 = k.replace(f"mask_fcn{i}", f"{i - 1}.0")
        new[k2] = v
    m.load_state_dict(new, strict=True)                      # torchvision >= 0.13 key layout
    # phase-1 -> phase-2 hand-off copies match_predictor into temporal_aggregator (ref :325-328)
    m.load_saved_matchrcnn({k: v for k, v in sd.items() if "temporal_aggregator" not in k})
    ta, mp = m.roi_heads.temporal_aggregator.state_dict(), m.roi_heads.match_predictor.state_dict()
    assert all(torch.equal(ta[k], mp[k]) for k in mp)
    assert m.roi_heads.temporal_aggregator.n_frames == -1 and m.roi_heads.temporal_aggregator.nlb is True
    m1 = matchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    assert not any("temporal_aggregator" in k for k in m1.state_dict())
    with pytest.raises(NotImplementedError):
        m.train()([torch.zeros(3, 32, 32)])


def test_grad_enabled_nlb_long_sequences_fail_in_backward_only():
    """A grad-enabled direct call of the non-local block on more than 64 frames returns the forward kernel's result (the reference
    module accepts such calls; an eval-mode call without no_grad is one) and raises, with a clear message, only if backward() is
    asked for (the backward kernel keeps a sequence in LDS).  Without a GPU the forward itself fails loudly -- no CPU fallback."""
    import pytest
    import torch
    from seam_match_rcnn_amd import _native
    from seam_match_rcnn_amd.models.nlb import NONLocalBlock1D, _ForwardOnly
    blk = NONLocalBlock1D(256, sub_sample=False, bn_layer=False)
    with pytest.raises(_native.SeamNativeError):
        blk(torch.zeros(1, 256, 65, requires_grad=True))
    z = torch.zeros(1, 256, 65)
    out = _ForwardOnly.apply(z, 65, torch.zeros(3, requires_grad=True))
    assert out.requires_grad and torch.equal(out, z)
    with pytest.raises(NotImplementedError, match="<= 64 frames"):
        out.sum().backward()
