import copy, importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
M = importlib.import_module("cv-diffusion-model_amd")
from conftest import synth_input
dev = torch.device("cuda:0")
cd = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sched = M.LCMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", prediction_type="v_prediction", rescale_betas_zero_snr=True)
torch.manual_seed(3)
a = M.LowLightDiffusion(unet_variant="small", image_size=64, compute_dtype=cd, scheduler=sched).to(dev).train()
b = copy.deepcopy(a)
low = synth_input("r4:tlow", (2, 3, 64, 64), -1.0, -0.2).to(dev)
normal = synth_input("r4:tnormal", (2, 3, 64, 64), -1.0, 1.0).to(dev)
pa, pb = list(a.parameters()), list(b.parameters())
names = [n for n, _ in a.named_parameters()]
kw = dict(lr=1e-3, weight_decay=0.01)
opt_a = torch.optim.AdamW(pa, **kw, foreach=False, fused=False)
opt_b = M.FusedAdamW(pb, **kw, max_grad_norm=1.0, ema_decay=0.999)
step_b = M.TrainStep(b, opt_b, loss_type="mse", use_velocity_target=True)
for it in range(3):
    torch.manual_seed(100 + it)
    opt_a.zero_grad(set_to_none=True)
    la = a.compute_loss(low, normal, loss_type="mse", use_velocity_target=True)
    la.backward()
    ga = [p.grad.detach().clone() for p in pa]
    na = torch.nn.utils.clip_grad_norm_(pa, 1.0)
    opt_a.step()
    torch.manual_seed(100 + it)
    if len(sys.argv) > 2 and sys.argv[2] == "dirty":
        b.unet.mark_weights_dirty()
    lb = step_b(low, normal)
    offs = step_b._offsets
    gb = [step_b._flat[o:o + p.numel()].view_as(p) for o, p in zip(offs, pb)]
    dg = [((x - y).abs().max() / (y.abs().max() + 1e-20)).item() for x, y in zip(gb, ga)]
    dp = [((x - y).abs().max() / (y.abs().max() + 1e-6)).item() for x, y in zip(pb, pa)]
    print(f"step {it}: loss {la.item():.6f} {lb.item():.6f} norm {na.item():.6f} {opt_b.grad_norm().item():.6f}")
    wg = sorted(range(len(dg)), key=lambda i: -dg[i])[:5]
    wp = sorted(range(len(dp)), key=lambda i: -dp[i])[:5]
    print("  worst grads:", [(names[i], f"{dg[i]:.2e}", f"{ga[i].abs().max().item():.2e}") for i in wg])
    print("  worst params:", [(names[i], f"{dp[i]:.2e}", f"{pa[i].abs().max().item():.2e}", tuple(pa[i].shape)) for i in wp])
