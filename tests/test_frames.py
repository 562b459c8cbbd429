"""Input-pipeline row f4: the oracle's Pillow restatement vs fixtures produced by Pillow (CPU), the device kernels vs
the oracle bit-exactly (GPU)."""
import os
import sys

import numpy as np
import pytest
import torch

import seam_match_rcnn_amd.synth as synth
from conftest import ROOT
from oracle import frames as OF

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_frames_golden as MG          # noqa: E402  (the generator's input builder; PIL is imported lazily)


def test_oracle_resize_matches_pillow_fixture():
    g = np.load(os.path.join(ROOT, "tests", "golden", "frames_golden.npz"))
    for i, (h, w, oh, ow) in enumerate(MG.CASES):
        np.testing.assert_array_equal(OF.pil_resize(MG.image(i, h, w), ow, oh), g[f"case{i}"])


def test_oracle_resize_matches_installed_pillow():
    Image = pytest.importorskip("PIL.Image")
    img = MG.image(9, 120, 214)
    np.testing.assert_array_equal(OF.pil_resize(img, 107, 60), np.asarray(Image.fromarray(img).resize((107, 60))))


def test_oracle_noise_flip_known_answers():
    bgr = np.array([[[10, 20, 30], [250, 0, 128]]], np.uint8)
    np.testing.assert_array_equal(OF.noise_flip(bgr, None, 0.0), [[[30, 20, 10], [128, 0, 250]]])
    n = np.zeros((1, 2, 3)); n[0, 0, 0] = 1.0; n[0, 1, 2] = 10.0; n[0, 1, 1] = -1.0
    out = OF.noise_flip(bgr, n, 0.05)
    assert out[0, 0, 0] == int((30 / 255.0 + 0.05) * 255.0) and out[0, 1, 2] == 255 and out[0, 1, 1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(64, 96), (101, 75), (270, 480), (1080, 1920)])
def test_device_frame_prep_bit_exact(h, w):
    from seam_match_rcnn_amd import frames, ops
    dev = torch.device("cuda:0")
    bgr = MG.image(3, h, w)
    noise = synth.normal(synth.stream_id(80, f"n{h}"), (h, w, 3)).astype(np.float64) * 1.3
    for sigma in (0.05, 0.25):
        want = OF.prepare_frame(bgr, noise, sigma)
        got = frames.prepare_frame(torch.from_numpy(bgr).to(dev), True, sigma, noise_values=torch.from_numpy(noise).to(dev))
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(frames.prepare_frame(torch.from_numpy(bgr).to(dev), noise=False).cpu().numpy(),
                                  OF.prepare_frame(bgr, None, 0.0))
    # upscale + non-2x ratios through the same kernel
    np.testing.assert_array_equal(ops.resize_bicubic_u8(torch.from_numpy(bgr).to(dev), h + 7, w // 3).cpu().numpy(),
                                  OF.pil_resize(bgr, w // 3, h + 7))


@pytest.mark.gpu
def test_device_noise_statistics_and_determinism():
    from seam_match_rcnn_amd import ops
    dev = torch.device("cuda:0")
    bgr = torch.full((512, 512, 3), 128, dtype=torch.uint8, device=dev)
    a = ops.frame_noise(bgr, 0.05, seed=5)
    b = ops.frame_noise(bgr, 0.05, seed=5)
    c = ops.frame_noise(bgr, 0.05, seed=6)
    assert torch.equal(a, b) and not torch.equal(a, c)
    d = a.cpu().numpy().astype(np.float64) / 255.0 - 128 / 255.0
    assert abs(d.mean() + 0.5 / 255) < 1e-3          # truncation to uint8 biases by half a level
    assert abs(d.std() - 0.05) < 1e-3
    # a prepared clip feeds the model's transform as uint8 HWC
    from seam_match_rcnn_amd import frames
    clip = frames.prepare_clip([torch.from_numpy(MG.image(i, 96, 128)).to(dev) for i in range(2)], seed=3)
    assert all(f.dtype == torch.uint8 and tuple(f.shape) == (48, 64, 3) for f in clip)
