"""CPU oracle for the LCM denoising hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU (fp32) functional restatement of the reference's
algorithm for the path `LowLightDiffusion.enhance -> EfficientUNet.forward x n -> LCMScheduler.step`
(/root/reference/src/models/{low_light_diffusion,efficient_unet,lcm_scheduler}.py).

It is the *checker*, never the product:
  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` import it;
  * nothing under `cv-diffusion-model_amd/` imports it, and the product path raises when the HIP
    library is missing instead of falling back to this code.

Pinning: the restatement is checked against golden vectors produced by importing the reference in
the build container (`tools/make_golden.py`, fixtures under `tests/golden/`); see
`tests/test_oracle_golden.py`.  The reference ships no tests or fixtures of its own.
"""
from .spec import UNetSpec, VARIANTS, make_spec, param_shapes, block_plan
from .unet_ref import unet_forward, sinusoidal_embedding
from .scheduler_ref import LCMTables, lcm_timesteps, lcm_step, add_noise, get_velocity
from .pipeline_ref import enhance_ref, draw_noise
from .weightgen import synth_state_dict

__all__ = [
    "UNetSpec", "VARIANTS", "make_spec", "param_shapes", "block_plan",
    "unet_forward", "sinusoidal_embedding",
    "LCMTables", "lcm_timesteps", "lcm_step", "add_noise", "get_velocity",
    "enhance_ref", "draw_noise", "synth_state_dict",
]
