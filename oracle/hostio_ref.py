"""CPU restatement of the byte-level image I/O either side of `enhance` -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/scripts/inference.py:99-117 (`preprocess_image`: cv2.resize to S x S, `/127.5 - 1`,
HWC -> BCHW) and :120-134 (`postprocess_image`: BCHW -> HWC, `(x+1)*127.5`, `np.clip(0,255).astype(uint8)` i.e.
truncation, cv2.resize back to the original size).

The one third-party piece is `cv2.resize(img, (w, h))` with its default INTER_LINEAR (opencv-python >=4.8,<4.12 per the
reference's requirements.txt:16; not installed here, no reference fixture holds a resized image).  Its published
geometry -- half-pixel centres, source coordinate (o + 0.5) * n_in / n_out - 0.5, the two neighbours clamped to the
image, no antialiasing -- is restated below in float64 with round-half-up; OpenCV's uint8 path evaluates the same
interpolation in 11-bit fixed point, so its results can differ from this by one LSB.  Parity of the resize's last bit
is therefore UNPINNED; everything else on this path (normalise, denormalise, clip, truncate, layout) is exact.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def _axis(n_in: int, n_out: int):
    pos = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
    lo = np.floor(pos)
    frac = pos - lo
    lo = lo.astype(np.int64)
    return np.clip(lo, 0, n_in - 1), np.clip(lo + 1, 0, n_in - 1), frac


def resize_linear_ref(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 HWC -> uint8 HWC, one output pixel at a time per axis (separable), float64, round half up."""
    assert img.dtype == np.uint8 and img.ndim == 3
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()  # cv2.resize returns a copy when the size is unchanged
    y0, y1, fy = _axis(in_h, out_h)
    x0, x1, fx = _axis(in_w, out_w)
    src = img.astype(np.float64)
    out = np.empty((out_h, out_w, img.shape[2]), dtype=np.float64)
    for oy in range(out_h):
        top = src[y0[oy]][x0] * (1.0 - fx)[:, None] + src[y0[oy]][x1] * fx[:, None]
        bot = src[y1[oy]][x0] * (1.0 - fx)[:, None] + src[y1[oy]][x1] * fx[:, None]
        out[oy] = top * (1.0 - fy[oy]) + bot * fy[oy]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


def preprocess_ref(rgb_u8: np.ndarray, target_size: int) -> Tuple[np.ndarray, Tuple[int, int]]:
    """inference.py:99-117 from the decoded RGB array on: -> ([1,3,S,S] float32 in [-1,1], (H, W))."""
    original = rgb_u8.shape[:2]
    image = resize_linear_ref(rgb_u8, target_size, target_size)
    image = image.astype(np.float32) / 127.5 - 1.0
    return image.transpose(2, 0, 1)[np.newaxis, ...], original


def postprocess_ref(output: np.ndarray, original_size: Tuple[int, int]) -> np.ndarray:
    """inference.py:120-134: [1,3,S,S] float -> uint8 HWC at the original size."""
    out = output[0].transpose(1, 2, 0)
    out = (out + 1.0) * 127.5
    out = np.clip(out, 0, 255).astype(np.uint8)
    return resize_linear_ref(out, original_size[0], original_size[1])
