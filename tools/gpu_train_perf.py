"""Training-step timing (BASELINE config 5 shape: small@256, per-GPU batch from argv): engine forward with kept
activations + reverse pass + AdamW/clip/EMA as the reference trainer does them (trainer.py:281-338)."""
import importlib, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import torch.nn.functional as F
M = importlib.import_module("cv-diffusion-model_amd")

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
size = int(sys.argv[4]) if len(sys.argv) > 4 else 256
dev = torch.device("cuda:0")
if os.environ.get("WGRAD_TARGET"):
    importlib.import_module("cv-diffusion-model_amd._native").lib().llie_tune(b"wgrad_target", int(os.environ["WGRAD_TARGET"]))
if os.environ.get("BWD_ASYNC"):
    importlib.import_module("cv-diffusion-model_amd._native").lib().llie_tune(b"bwd_async", int(os.environ["BWD_ASYNC"]))
m = M.LowLightDiffusion(unet_variant="small", image_size=size).to(dev).train()
m.compute_dtype = None if dtype == "fp32" else dtype
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01, fused=True)
low = torch.rand(batch, 3, size, size, device=dev) * 2 - 1
normal = torch.rand(batch, 3, size, size, device=dev) * 2 - 1


def step(parts=None):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = m.compute_loss(low, normal)
    if parts is not None:
        torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    if parts is not None:
        torch.cuda.synchronize(); t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    if parts is not None:
        torch.cuda.synchronize(); t3 = time.perf_counter()
        parts.append((t1 - t0, t2 - t1, t3 - t2))
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
parts = []
for _ in range(3):
    step(parts)
f, b, o = [sum(p[i] for p in parts) / len(parts) * 1e3 for i in range(3)]
print(f"train step small@{size} B={batch} {dtype}: {dt*1e3:.1f} ms/step = {batch/dt:.1f} img/s  "
      f"(forward+loss {f:.1f} ms, backward {b:.1f} ms, clip+AdamW {o:.1f} ms)  loss {loss.item():.4f}  "
      f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
