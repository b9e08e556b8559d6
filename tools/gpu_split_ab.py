"""A/B: llie_enhance captured as one chain vs two concurrent half-batch branches (hipGraph path), small@256 fp16."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
M = importlib.import_module("cv-diffusion-model_amd")
N = importlib.import_module("cv-diffusion-model_amd._native")
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype=dtype).to(dev).eval()
low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
noise = torch.randn(4, B, 3, 256, 256, device=dev)
outs = {}
for rep in range(2):
    for split in (0, 1):
        N.lib().llie_tune(b"enhance_split", split)
        for _ in range(3):
            out = m.enhance(low, 4, noise=noise)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            out = m.enhance(low, 4, noise=noise)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        outs[split] = out
        print(f"split={split}: {dt*1e3:.2f} ms/call = {B/dt:.1f} img/s", flush=True)
print("bitwise equal:", torch.equal(outs[0], outs[1]))
