import importlib, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle
M = importlib.import_module("cv-diffusion-model_amd")
dev = torch.device("cuda:0")
for variant, size, b in (("large", 96, 1), ("large", 136, 1), ("base", 72, 2), ("tiny", 88, 2), ("small", 64 + 8, 1), ("small", 328, 1)):
    unp = variant in ("tiny", "base")
    spec = oracle.make_spec(variant, size, allow_unpinned=unp)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant=variant, image_size=size, num_inference_steps=4, allow_unpinned_groupnorm=unp)
    m.load_state_dict(sd); m = m.to(dev).eval()
    low = torch.rand(b, 3, size, size, generator=torch.Generator().manual_seed(1)) * 2 - 1
    noise = oracle.draw_noise(b, size, 4, seed=2)
    ref = oracle.enhance_ref(sd, spec, low, 4, noise)
    out = m.enhance(low.to(dev), 4, noise=torch.stack(noise), return_intermediate=True)
    err = max((a.cpu() - r).abs().max().item() for a, r in zip(out.intermediate, ref["intermediate"]))
    m.compute_dtype = "bf16"
    o = m.enhance(low.to(dev), 4, noise=torch.stack(noise).to(dev))
    mse = (((o.cpu().clamp(-1, 1) - ref["enhanced"].clamp(-1, 1)) / 2) ** 2).mean().item()
    import math
    print(f"{variant}@{size} B={b}: fp32 max-abs {err:.2e}; bf16 PSNR {10 * math.log10(1 / mse):.1f} dB", flush=True)
