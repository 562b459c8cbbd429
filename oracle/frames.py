"""ORACLE (test infrastructure, not product code) -- the frame preparation of the input pipeline on the CPU.

NumPy restatement of what ``MovingFashionDataset.__getitem__`` does to a decoded video frame
(ref datasets/MFDataset.py:79-93): BGR->RGB, additive noise in float64, clip, uint8 truncation, then
``PIL.Image.resize`` to half size.  ``Image.resize`` is third-party arithmetic (Pillow, version unpinned by the
reference; Pillow 12.2 is installed here): default filter BICUBIC with an antialiasing support scaled by the reduction
factor, 8-bit fixed-point coefficients, horizontal pass rounded to uint8 before the vertical pass
(Pillow src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
Vertical_8bpc) -- restated below from that published algorithm.

PINNED: tests/test_frames.py checks ``pil_resize`` against fixtures produced by Pillow itself
(tests/golden/make_frames_golden.py) and, where Pillow is importable, against ``Image.resize`` directly.
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def noise_flip(bgr: np.ndarray, noise: np.ndarray | None, sigma: float) -> np.ndarray:
    """ref MFDataset.py:81-88.  bgr uint8 [H,W,3]; noise float64 [H,W,3] standard-normal draws (None: noise=False)."""
    image = bgr[:, :, ::-1]
    if noise is None:
        return np.ascontiguousarray(image)
    image = image / 255.0
    image = image + noise * sigma
    image = image * 255.0
    image = np.clip(image, 0, 255.0)
    return np.asarray(image, dtype=np.uint8)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(in_size: int, out_size: int):
    scale = float(np.float32(in_size) - np.float32(0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        k = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w]
        out.append((xmin, np.asarray(k, dtype=np.int64)))
    return out


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """resample axis 0 (rows) or 1 (columns) of a uint8 [H,W,3] image."""
    co = _coeffs(img.shape[axis], out_size)
    src = img.astype(np.int64)
    shape = list(img.shape)
    shape[axis] = out_size
    dst = np.empty(shape, dtype=np.uint8)
    for o, (lo, k) in enumerate(co):
        sl = src[lo:lo + len(k)] if axis == 0 else src[:, lo:lo + len(k)]
        kk = k[:, None, None] if axis == 0 else k[None, :, None]
        acc = (1 << (PRECISION_BITS - 1)) + (sl * kk).sum(axis)
        v = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        if axis == 0:
            dst[o] = v
        else:
            dst[:, o] = v
    return dst


def pil_resize(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.fromarray(img).resize((out_w, out_h))`` for uint8 RGB [H,W,3] (ref MFDataset.py:89-92)."""
    tmp = _pass(img, out_w, 1) if out_w != img.shape[1] else img       # horizontal pass first
    return _pass(tmp, out_h, 0) if out_h != img.shape[0] else tmp


def prepare_frame(bgr: np.ndarray, noise: np.ndarray | None, sigma: float) -> np.ndarray:
    """decoded frame -> what the dataset hands to ``ToTensor``: noise=True halves the resolution, noise=False does not."""
    rgb = noise_flip(bgr, noise, sigma)
    if noise is None:
        return rgb
    return pil_resize(rgb, rgb.shape[1] // 2, rgb.shape[0] // 2)
