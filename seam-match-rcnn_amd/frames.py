"""Device-side frame preparation: the per-frame work of ``MovingFashionDataset.__getitem__``
(ref datasets/MFDataset.py:79-93) after ``cv2.VideoCapture.read()``.

The reference does this per frame on a dataloader worker: BGR->RGB, float64 noise over the full-resolution frame
(``np.random.randn`` of 6.2 M values for 1080p), clip, uint8, PIL bicubic resize to half size, ``ToTensor``.  Here the
decoded uint8 frame is uploaded once (1 byte per sample) and everything else runs on the GPU; the result is the uint8
RGB image the model's transform consumes directly (``seam_preprocess_u8`` fuses ToTensor + normalise + resize + pad).
Video decode itself stays with cv2 (out of scope: no decoder in this image).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def noise_sigma(u: float) -> float:
    """``tmp_noise = 0.25 if random.random() > 0.75 else 0.05`` (ref MFDataset.py:82), with the draw made explicit."""
    return 0.25 if u > 0.75 else 0.05


@torch.no_grad()
def prepare_frame(bgr: torch.Tensor, noise: bool = True, sigma: float = 0.05, seed: int = 0,
                  noise_values: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decoded frame uint8 [H,W,3] BGR (device) -> uint8 RGB: noisy and half resolution when ``noise`` (the training /
    default test setting), otherwise the plain channel flip.  ``noise_values``: optional float64 [H,W,3] standard-normal
    draws (bit-exact with the NumPy pipeline given the same draws); default: on-device draws keyed by ``seed``."""
    if not noise:
        return ops.frame_noise(bgr, 0.0)
    rgb = ops.frame_noise(bgr, sigma, noise_values, seed)
    return ops.resize_bicubic_u8(rgb, bgr.shape[0] // 2, bgr.shape[1] // 2)


@torch.no_grad()
def prepare_clip(frames, noise: bool = True, sigmas=None, seed: int = 0):
    """A list of decoded frames -> list of uint8 RGB images ready for ``model(images)`` (which accepts uint8 HWC)."""
    out = []
    for i, f in enumerate(frames):
        s = 0.05 if sigmas is None else sigmas[i]
        out.append(prepare_frame(f, noise, s, seed + 7919 * i))
    return out
