"""Whole-UNet gradient error summary (debug aid): rel-to-max and relative L2 per tensor, worst first."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_gpu_training as T
dev = torch.device("cuda:0")
dtype = sys.argv[1] if len(sys.argv) > 1 else None
m, sd, spec = T._small(64, dev)
m.compute_dtype = dtype
g = torch.Generator().manual_seed(3)
low = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
normal = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
noise = torch.randn(2, 3, 64, 64, generator=g)
t = torch.tensor([500, 37])
loss_ref, pred_ref, gref = T._ref_unet_grads(sd, spec, low, normal, t, noise)
out = m(low.to(dev), normal.to(dev), timesteps=t.to(dev), noise=noise.to(dev))
loss = torch.nn.functional.mse_loss(out["noise_pred"], out["noise"])
loss.backward()
print("loss", loss.item(), loss_ref.item())
rows = []
for k, p in m.named_parameters():
    a, b = p.grad.double().cpu(), gref[k].double()
    rows.append((((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item(), T.cosine(a, b), k))
rows.sort(reverse=True)
for r in rows[:12]:
    print(f"{r[0]:.2e} l2 {r[1]:.2e} cos {r[2]:.6f} {r[3]}")
print("median relmax", sorted(r[0] for r in rows)[len(rows) // 2], "worst l2", max(r[1] for r in rows), "min cos", min(r[2] for r in rows))
