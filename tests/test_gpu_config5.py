"""BASELINE configs[4] at its FULL per-clip size on the fp16-MFMA / fp32-accumulate path: one clip of 30 frames 1080x1920
(-> 749x1333, padded 768x1344), 64 fixed ROIs per frame = 1920 ROIs, 64 sequences of 30, 1000-product gallery.

The oracle (fp32, CPU) finishes a 1080p frame in seconds, so a SAMPLE of the clip's frames is checked against it with the
norm-wise bounds of tests/test_gpu_fp16.py (fp16 storage of every activation: 2^-11 per rounding over ~50 layers); the whole
clip is then checked through what does not need the oracle at full size: the exact-fp32 GPU path on the same clip (itself
within 1e-3 of the oracle on the sampled frames) must give nearly the same aggregated descriptors, match logits and top-20
product sets as the fp16 path, and per-frame results must not depend on the batch the frame rode in."""
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import to_torch
from oracle import detection as OD
from oracle import heads as OH
from oracle import model as OM
from test_gpu_ops import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, R, G, K = 30, 64, 1000, 20
SAMPLE = (0, 17)                       # frames of the clip the CPU oracle recomputes


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def clip():
    from seam_match_rcnn_amd.models.video_matchrcnn import videomatchrcnn_resnet50_fpn
    from seam_match_rcnn_amd.models.detection import resized_size
    sd = to_torch(synth.video_matchrcnn_state(5))
    m = videomatchrcnn_resnet50_fpn(pretrained_backbone=False, num_classes=14)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    frames = torch.from_numpy(synth.frames(500, T, 1080, 1920))
    rh, rw, _ = resized_size(1080, 1920)
    assert (rh, rw) == (749, 1333)
    rois = torch.from_numpy(synth.fixed_rois(R, rh, rw))
    bank = torch.from_numpy(synth.gallery(7, G))
    return m, sd, frames, rois, bank


def run_clip(m, frames, rois, bank, dtype):
    from seam_match_rcnn_amd import ops, retrieval
    m.set_compute_dtype(dtype)
    ta = m.roi_heads.temporal_aggregator
    fl = list(frames.to(DEV).unbind(0))
    with torch.no_grad():
        res, feats, _ = m.forward_fixed_rois(fl, [rois.to(DEV)] * len(fl), run_rpn_head=True)
        x = ops.cat_rows([r["roi_features"] for r in res])
        ids = torch.arange(R, dtype=torch.int64).repeat(len(fl))
        out = ta(x, torch.zeros(len(ids), dtype=torch.int32), ids)
        x5, idx, score = retrieval.match_sequences(ta, out[0], bank.to(DEV), K)
    torch.cuda.synchronize()
    return res, feats, out, x5, idx, score


def test_config5_full_clip_fp16_vs_oracle_sample_and_fp32_path(clip):
    m, sd, frames, rois, bank = clip
    try:
        res16, feats16, out16, x5_16, idx16, _ = run_clip(m, frames, rois, bank, torch.float16)
        assert feats16["0"].dtype == torch.float16 and feats16["0"].shape == (T, 192, 336, 256)
        assert out16[0].shape == (R, 256) and x5_16.shape == (R, G, 2) and idx16.shape == (R, K)
        rf16 = {f: res16[f]["roi_features"].clone() for f in SAMPLE}
        mf16 = {f: res16[f]["match_features"].clone() for f in SAMPLE}
        fp16_lvls = {f: {k: feats16[k][f].float().cpu() for k in ("0", "3")} for f in SAMPLE}
        x3_1b16, x5_16c = out16[0].clone(), x5_16.clone()
        del res16, feats16, out16, x5_16
        torch.cuda.empty_cache()
        res32, feats32, out32, x5_32, idx32, _ = run_clip(m, frames, rois, bank, torch.float32)
    finally:
        m.set_compute_dtype(torch.float32)
    mp = OM.sub(sd, "roi_heads.match_predictor.")
    for f in SAMPLE:
        batch, sizes = OD.transform([frames[f]])
        assert tuple(batch.shape[-2:]) == (768, 1344) and tuple(sizes[0]) == (749, 1333)
        ofe = OD.fpn(OD.resnet50_body(batch, sd), sd)
        orf = OD.multiscale_roi_align([ofe[k] for k in "0123"], [rois], sizes, 14)
        ox3 = OH.match_trunk(orf, mp)
        # exact-fp32 path: the north_star tolerance, element-wise
        assert_close(feats32["0"][f].permute(2, 0, 1), ofe["0"][0])
        assert_close(res32[f]["roi_features"], orf)
        assert_close(res32[f]["match_features"], ox3)
        # fp16 path: norm-wise bounds (same as tests/test_gpu_fp16.py at 256x320)
        assert rel(fp16_lvls[f]["0"].permute(2, 0, 1), ofe["0"][0]) < 5e-3
        assert rel(fp16_lvls[f]["3"].permute(2, 0, 1), ofe["3"][0]) < 5e-3
        assert rel(rf16[f], orf) < 5e-3
        assert mf16[f].dtype == torch.float32 and rel(mf16[f], ox3) < 1e-2
    # whole clip: fp16 vs the exact path on all 64 sequences x 30 frames
    e_desc, e_logit = rel(x3_1b16, out32[0]), rel(x5_16c, x5_32)
    assert e_desc < 1e-2 and e_logit < 2e-2, (e_desc, e_logit)
    a, b = idx16.cpu(), idx32.cpu()
    overlap = sum(len(set(x.tolist()) & set(y.tolist())) for x, y in zip(a, b)) / float(a.numel())
    top1 = float((a[:, 0] == b[:, 0]).float().mean())
    print(f"config 5 full clip: fp16 vs fp32 descriptors {e_desc:.2e}, logits {e_logit:.2e}, top-20 set overlap {overlap:.4f}, "
          f"top-1 equal {top1:.3f}")
    assert overlap >= 0.9 and top1 >= 0.8, (overlap, top1)


def test_config5_frame_results_do_not_depend_on_the_batch(clip):
    """A frame of the 30-frame fp16 batch alone gives the same bits (the kernels pick tiles from the map geometry only)."""
    m, sd, frames, rois, bank = clip
    try:
        m.set_compute_dtype(torch.float16)
        with torch.no_grad():
            fl = list(frames[:4].to(DEV).unbind(0))
            r4, f4, _ = m.forward_fixed_rois(fl, [rois.to(DEV)] * 4)
            r1, f1, _ = m.forward_fixed_rois(fl[2:3], [rois.to(DEV)])
        torch.cuda.synchronize()
        assert torch.equal(f4["0"][2], f1["0"][0]) and torch.equal(f4["3"][2], f1["3"][0])
        assert torch.equal(r4[2]["roi_features"], r1[0]["roi_features"])
        assert torch.equal(r4[2]["match_features"], r1[0]["match_features"])
    finally:
        m.set_compute_dtype(torch.float32)
