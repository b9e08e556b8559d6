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


def preprocess_device(rgb_u8: torch.Tensor, target_size: int) -> torch.Tensor:
    """Device twin of preprocess_array: uint8 [B,H,W,3] (or [H,W,3]) on a HIP device -> fp32 [B,3,S,S] in
    [-1,1], resized and normalised by libllie_hip.so; bit-exact with the host path."""
    from . import _native as N
    if rgb_u8.device.type != "cuda" or rgb_u8.dtype != torch.uint8:
        raise RuntimeError("preprocess_device expects a uint8 tensor on a HIP device")
    x = rgb_u8 if rgb_u8.dim() == 4 else rgb_u8.unsqueeze(0)
    x = x.contiguous()
    b, h, w, c = x.shape
    if c != 3:
        raise ValueError("expected RGB images, [.., H, W, 3]")
    out = torch.empty(b, 3, target_size, target_size, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().llie_preprocess_u8(x.data_ptr(), b, h, w, out.data_ptr(), target_size,
                                           torch.cuda.current_stream(x.device).cuda_stream), "preprocess_u8")
    return out


def postprocess_device(output: torch.Tensor, original_size: Tuple[int, int]) -> torch.Tensor:
    """Device twin of postprocess_array: fp32 [B,3,S,S] -> uint8 [B,H0,W0,3]; bit-exact with the host path."""
    from . import _native as N
    if output.device.type != "cuda":
        raise RuntimeError("postprocess_device expects a tensor on a HIP device")
    x = output.detach().float().contiguous()
    b, c, s, s2 = x.shape
    if c != 3 or s != s2:
        raise ValueError("expected [B,3,S,S]")
    h0, w0 = original_size
    img = torch.empty(b, h0, w0, 3, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().llie_postprocess_u8(x.data_ptr(), b, s, img.data_ptr(), h0, w0,
                                            torch.cuda.current_stream(x.device).cuda_stream), "postprocess_u8")
    return img


def load_image(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def save_image(path: str, rgb_u8: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(rgb_u8).save(path)
