#!/usr/bin/env python3
"""Golden vectors of the half-resolution frame resize, produced by Pillow itself (the reference's dependency:
``img.resize((w // 2, h // 2))`` at datasets/MFDataset.py:92).  Inputs come from synth.py; stored: the resized uint8
images for a few sizes (even / odd, a non-2x ratio, an upscale).  Run: python tests/golden/make_frames_golden.py"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import seam_match_rcnn_amd.synth as synth  # noqa: E402

CASES = [(64, 96, 32, 48), (101, 75, 50, 37), (90, 160, 45, 80), (33, 47, 20, 11), (24, 31, 40, 50)]   # H, W, OH, OW


def image(case_id, h, w):
    u = synth.uniform(synth.stream_id(70 + case_id, "frame"), (h, w, 3))
    # smooth-ish content with hard edges: uniform noise + blocks, so rounding AND clipping are exercised
    img = (u * 255).astype(np.uint8)
    img[h // 4:h // 2, w // 3:2 * w // 3] = 255
    img[h // 2:3 * h // 4, w // 5:w // 2] = 0
    return img


def main():
    import PIL
    g = {"pillow_version": np.asarray(PIL.__version__)}
    for i, (h, w, oh, ow) in enumerate(CASES):
        g[f"case{i}"] = np.asarray(Image.fromarray(image(i, h, w)).resize((ow, oh)))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frames_golden.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, PIL.__version__, {k: v.shape for k, v in g.items()})


if __name__ == "__main__":
    main()
