"""Stress of the side-stream backward pass: the same training step repeated N times (small@256, B from argv) must give
bit-identical gradients every time, and the same bits as the single-stream order (bwd_async = 0)."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype="bf16").to(dev).train()
low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
normal = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
noise = torch.randn(B, 3, 256, 256, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)


def grads():
    m.zero_grad(set_to_none=True)
    out = m(low, normal, timesteps=t, noise=noise)
    torch.nn.functional.mse_loss(out["noise_pred"], out["noise"]).backward()
    return torch.cat([p.grad.flatten() for p in m.parameters()]).clone()


N.lib().llie_tune(b"bwd_async", 0)
ref = grads()
N.lib().llie_tune(b"bwd_async", 1)
bad = 0
for i in range(reps):
    g = grads()
    if not torch.equal(g, ref):
        bad += 1
        print(f"rep {i}: {(g != ref).sum().item()} differing entries, max |diff| {(g - ref).abs().max().item():.3e}")
print(f"B={B}: {reps} side-stream backward passes, {bad} differ from the single-stream result; finite: {bool(torch.isfinite(ref).all())}")
sys.exit(1 if bad else 0)
