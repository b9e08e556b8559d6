"""MI355X-native engine for the LCM denoising hot path of zamazincode/cv-diffusion-model.

Public names mirror `src/models/__init__.py:1-10` of the reference (LowLightDiffusion, EfficientUNet,
EfficientUNetConfig, LCMScheduler) plus the operator classes of efficient_unet.py.  Compute lives in
`libllie_hip.so` (hand-written HIP for gfx950, C ABI in include/llie.h); importing this package does
not need a GPU, running anything does.

The directory name contains '-', so import it as `import cv_diffusion_model_amd` (alias module at the
repository root) or `importlib.import_module("cv-diffusion-model_amd")`.
"""
from .unet import (EfficientUNet, EfficientUNetConfig, create_efficient_unet, InvertedResidualBlock,
                   LinearAttention, Downsample, Upsample, SqueezeExcitation)
from .scheduler import LCMScheduler, LCMSchedulerOutput, LCMDenoisingLoop, get_lcm_timesteps
from .pipeline import LowLightDiffusion, LowLightDiffusionOutput, normalize_image, denormalize_image
from .sharding import shard_range, enhance_sharded, all_gather_batch, all_reduce_gradients
from .training import FusedAdamW, TrainStep
from .build import build_library, library_path
from . import ops  # registers torch.ops.llie.*
from .ops import register_model
from .hostio import (load_checkpoint, extract_state_dict, preprocess_array, postprocess_array, resize_bilinear,
                     preprocess_device, postprocess_device)

__all__ = [
    "EfficientUNet", "EfficientUNetConfig", "create_efficient_unet", "InvertedResidualBlock", "LinearAttention", "SqueezeExcitation",
    "Downsample", "Upsample", "LCMScheduler", "LCMSchedulerOutput", "LCMDenoisingLoop", "get_lcm_timesteps", "LowLightDiffusion",
    "LowLightDiffusionOutput", "normalize_image", "denormalize_image", "shard_range", "enhance_sharded",
    "all_gather_batch", "all_reduce_gradients", "FusedAdamW", "TrainStep", "register_model", "build_library", "library_path", "load_checkpoint", "extract_state_dict",
    "preprocess_array", "postprocess_array", "resize_bilinear", "preprocess_device", "postprocess_device",
]
