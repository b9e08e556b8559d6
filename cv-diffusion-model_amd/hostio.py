"""Host-side I/O either side of the hot path: checkpoint containers and image pre/post-processing.

Mirrors what the reference's callers do around `enhance` (scripts/inference.py:65-84,99-134,
scripts/benchmark.py:56, src/training/trainer.py:415-456, scripts/export.py:151-155) without cv2
(absent in this image): PIL for file I/O and a NumPy bilinear resize with cv2.INTER_LINEAR's
half-pixel geometry.  cv2's uint8 fixed-point rounding is not reproduced bit for bit (+-1 LSB; cv2 is
not available to pin it against -- "parity unpinned" for the resize alone).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


# ------------------------------------------------------------------ checkpoints
def extract_state_dict(obj) -> Dict[str, torch.Tensor]:
    """Accept both container layouts of the reference: the trainer's dict with "model_state_dict"
    (trainer.py:418-434, read by inference.py:78-79) and a bare state_dict (export.py:151-155, read by
    benchmark.py:56)."""
    if isinstance(obj, dict) and "model_state_dict" in obj:
        return obj["model_state_dict"]
    if isinstance(obj, dict) and all(isinstance(v, torch.Tensor) for v in obj.values()):
        return obj
    raise ValueError("unrecognised checkpoint layout: expected a state_dict or a dict with 'model_state_dict'")


def load_checkpoint(model: torch.nn.Module, path: str, map_location="cpu") -> dict:
    """torch.load with weights_only=True (nothing from the file is executed), then load_state_dict.
    Returns the non-tensor metadata of a trainer checkpoint (epoch, global_step, ...) if present."""
    obj = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(extract_state_dict(obj))
    if isinstance(obj, dict) and "model_state_dict" in obj:
        return {k: v for k, v in obj.items() if k in ("epoch", "global_step", "best_val_loss")}
    return {}


# ------------------------------------------------------------------ images
def resize_bilinear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """HWC uint8/float -> HWC, bilinear with half-pixel centres and edge clamping (the geometry of
    cv2.resize(..., interpolation=INTER_LINEAR), no antialiasing).  uint8 in -> uint8 out (round half up)."""
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img.copy()
    src = img.astype(np.float32)

    def axis(n_in, n_out):
        pos = (np.arange(n_out, dtype=np.float32) + 0.5) * (n_in / n_out) - 0.5
        lo = np.floor(pos).astype(np.int64)
        frac = pos - lo
        lo0 = np.clip(lo, 0, n_in - 1)
        lo1 = np.clip(lo + 1, 0, n_in - 1)
        return lo0, lo1, frac.astype(np.float32)

    y0, y1, fy = axis(in_h, out_h)
    x0, x1, fx = axis(in_w, out_w)
    top = src[y0][:, x0] * (1 - fx)[None, :, None] + src[y0][:, x1] * fx[None, :, None]
    bot = src[y1][:, x0] * (1 - fx)[None, :, None] + src[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    if img.dtype == np.uint8:
        return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    return out.astype(img.dtype)


def preprocess_array(rgb_u8: np.ndarray, target_size: int) -> Tuple[np.ndarray, Tuple[int, int]]:
    """HWC uint8 RGB -> [1,3,S,S] float32 in [-1,1] and the original (H, W) (inference.py:99-117)."""
    original = rgb_u8.shape[:2]
    img = resize_bilinear(rgb_u8, target_size, target_size)
    x = img.astype(np.float32) / 127.5 - 1.0
    return x.transpose(2, 0, 1)[np.newaxis, ...], original


def postprocess_array(output: np.ndarray, original_size: Tuple[int, int]) -> np.ndarray:
    """[1,3,S,S] float in [-1,1] -> HWC uint8 RGB at the original size (inference.py:120-134)."""
    out = output[0].transpose(1, 2, 0)
    out = np.clip((out + 1.0) * 127.5, 0, 255).astype(np.uint8)
    return resize_bilinear(out, original_size[0], original_size[1])


def load_image(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def save_image(path: str, rgb_u8: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(rgb_u8).save(path)
