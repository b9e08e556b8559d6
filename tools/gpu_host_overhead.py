import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
M = importlib.import_module("cv-diffusion-model_amd")
dev = torch.device("cuda:0")
m = M.LowLightDiffusion(unet_variant="small", image_size=256, compute_dtype="fp16").to(dev).eval()
for B in (1, 4):
    low = torch.rand(B, 3, 256, 256, device=dev) * 2 - 1
    for _ in range(5): m.enhance(low, 4)
    torch.cuda.synchronize()
    # host-only cost: enqueue time of 50 calls without waiting
    t0 = time.perf_counter()
    for _ in range(50): m.enhance(low, 4)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3*(t1-t0)/50:.3f} ms/call, wall incl. drain {1e3*(t2-t0)/50:.3f} ms/call", flush=True)
import cProfile, pstats
low = torch.rand(1, 3, 256, 256, device=dev) * 2 - 1
pr = cProfile.Profile(); pr.enable()
for _ in range(50): m.enhance(low, 4)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
