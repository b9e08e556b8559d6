"""Times the weight reload an optimiser step triggers, and the training forward alone (debug aid)."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
M = importlib.import_module("cv-diffusion-model_amd")
dev = torch.device("cuda:0")
m = M.LowLightDiffusion(unet_variant="small", image_size=256).to(dev).train()
m.compute_dtype = "bf16"
low = torch.rand(8, 3, 256, 256, device=dev) * 2 - 1
normal = torch.rand(8, 3, 256, 256, device=dev) * 2 - 1
m.compute_loss(low, normal)
torch.cuda.synchronize()
def bump():
    with torch.no_grad():
        torch._foreach_add_(list(m.parameters()), 0.0)
for name, fn in [("reload", lambda: m.unet._handle(2)), ("compute_loss (no reload)", None)]:
    ts = []
    for _ in range(5):
        if fn is not None:
            bump()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if fn is not None:
            fn()
        else:
            m.compute_loss(low, normal)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, [round(t * 1e3, 2) for t in ts], "ms")
with torch.no_grad():
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.compute_loss(low, normal)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("compute_loss under no_grad (inference forward)", [round(t * 1e3, 2) for t in ts], "ms")
