"""Two ranks sharing one GPU over gloo (RCCL needs one device per rank): a data-parallel training step where each
rank takes half the batch must reproduce the single-process full-batch gradients after all_reduce_gradients, and
TrainStep + FusedAdamW on half batches (one all-reduce of the flat gradient buffer, the 1 / world on the optimiser's gradient
scale) must land on the parameters of a full-batch step."""
import importlib, os, sys, subprocess, socket
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch


def worker(rank, port):
    import torch.distributed as dist
    M = importlib.import_module("cv-diffusion-model_amd")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=2)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = M.LowLightDiffusion(unet_variant="small", image_size=64).to(dev).train()
    g = torch.Generator().manual_seed(5)
    low = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    normal = (torch.rand(4, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(4, 3, 64, 64, generator=g).to(dev)
    t = torch.tensor([5, 300, 650, 999], device=dev)
    # full batch, single process semantics
    out = m(low, normal, timesteps=t, noise=noise)
    torch.nn.functional.mse_loss(out["noise_pred"], out["noise"]).backward()
    full = [p.grad.clone() for p in m.parameters()]
    m.zero_grad(set_to_none=True)
    lo, hi = M.shard_range(4, rank, 2)
    out = m(low[lo:hi], normal[lo:hi], timesteps=t[lo:hi], noise=noise[lo:hi])
    torch.nn.functional.mse_loss(out["noise_pred"], out["noise"]).backward()
    calls = M.all_reduce_gradients(m.parameters())
    worst = max(((p.grad - f).abs().max() / f.abs().max().clamp_min(1e-20)).item() for p, f in zip(m.parameters(), full))
    print(f"rank {rank}: {calls} all_reduce call(s), worst rel-to-max gradient difference vs full batch {worst:.2e}", flush=True)
    assert calls == 1 and worst < 1e-4
    # TrainStep: both ranks on the full batch (sum of two equal gradients, halved) vs each rank on its half
    import copy
    m.zero_grad(set_to_none=True)
    ref = copy.deepcopy(m)
    kw = dict(lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, ema_decay=0.99)
    opt_ref = M.FusedAdamW(ref.parameters(), **kw)
    M.TrainStep(ref, opt_ref)(low, normal, timesteps=t, noise=noise)
    opt = M.FusedAdamW(m.parameters(), **kw)
    M.TrainStep(m, opt)(low[lo:hi], normal[lo:hi], timesteps=t[lo:hi], noise=noise[lo:hi])
    n_ref, n_dp = opt_ref.grad_norm().item(), opt.grad_norm().item()
    worst_p = max((p - q).abs().max().item() for p, q in zip(m.parameters(), ref.parameters()))
    worst_e = max((p - q).abs().max().item() for p, q in zip(opt.ema_tensors(), opt_ref.ema_tensors()))
    digest = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).double().sum().item()
    both = [None, None]
    dist.all_gather_object(both, digest)
    print(f"rank {rank}: train_step gradient norm {n_dp:.6f} vs full batch {n_ref:.6f}, worst parameter difference {worst_p:.2e} "
          f"(lr 1e-3), shadow {worst_e:.2e}, ranks agree: {both[0] == both[1]}", flush=True)
    # Adam's first step moves every entry by ~lr * g / (|g| + eps): entries whose gradient is below eps = 1e-8 turn a 1e-9
    # difference between the summation orders into 1e-4 of a step
    assert abs(n_dp - n_ref) <= 1e-4 * n_ref and worst_p < 1e-4 and worst_e < 1e-5 and both[0] == both[1]
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) == 3:
        worker(int(sys.argv[1]), int(sys.argv[2]))
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [subprocess.Popen([sys.executable, __file__, str(r), str(port)]) for r in range(2)]
        rc = [p.wait(timeout=600) for p in ps]
        print("exit codes", rc)
        sys.exit(max(rc))
