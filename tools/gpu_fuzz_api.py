#!/usr/bin/env python3
"""Randomised API-level checks on the GPU box (complements gpu_fuzz.py / gpu_fuzz_unet.py):
  1. hipGraph cache / workspace growth: a random sequence of enhance() calls with changing batch size, step count,
     dtype and output options on ONE model must reproduce what a fresh model gives for each call;
  2. v-prediction and deployment-loop schedulers through the fused-step output head (fp16) and the separate step kernel
     (fp32) against the oracle;
  3. device-side uint8 pre/post-processing vs the host implementation on random image sizes (bit-exact);
  4. training gradients of the whole network at a non-power-of-two size with a ragged batch.
"""
import importlib, os, random, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import oracle
from oracle import scheduler_ref as S
M = importlib.import_module("cv-diffusion-model_amd")
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fails = []

spec = oracle.make_spec("small", 64)
sd = oracle.synth_state_dict(oracle.param_shapes(spec))


def fresh(**kw):
    m = M.LowLightDiffusion(unet_variant="small", image_size=64, **kw)
    m.load_state_dict(sd)
    return m.to(dev).eval()


# ---- 1. graph cache / workspace churn
m = fresh()
for i in range(24):
    b, steps = rng.choice([1, 2, 3, 5, 8]), rng.choice([1, 2, 4, 4, 4, 6, 8])
    cd = rng.choice([None, None, "fp16", "bf16"])
    inter, preds = rng.random() < 0.3, rng.random() < 0.2
    g = torch.Generator().manual_seed(1000 + i)
    low = (torch.rand(b, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(steps, b, 3, 64, 64, generator=g).to(dev)
    m.compute_dtype = cd
    out = m.enhance(low, steps, noise=noise, return_intermediate=inter, return_noise_pred=preds)
    f = fresh(); f.compute_dtype = cd
    ref = f.enhance(low, steps, noise=noise, return_intermediate=True, return_noise_pred=True)
    e = out.enhanced if (inter or preds) else out
    ok = torch.equal(e, ref.enhanced)
    if inter:
        ok = ok and all(torch.equal(a, b_) for a, b_ in zip(out.intermediate, ref.intermediate))
    if preds:
        ok = ok and all(torch.equal(a, b_) for a, b_ in zip(out.noise_pred, ref.noise_pred))
    if not ok:
        fails.append(f"graph-churn call {i}: B={b} steps={steps} dtype={cd} inter={inter} preds={preds}")
    del f
print("1. graph/workspace churn: 24 calls", "OK" if not fails else fails, flush=True)

# ---- 2. alternative schedulers
n0 = len(fails)
low = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(5)) * 2 - 1
noise = oracle.draw_noise(2, 64, 4, seed=6)
mv = M.LowLightDiffusion(unet=m.unet, image_size=64, scheduler=M.LCMScheduler(prediction_type="v_prediction", rescale_betas_zero_snr=True))
tabv = S.LCMTables.build(rescale_betas_zero_snr=True, prediction_type="v_prediction")
refv = oracle.enhance_ref(sd, spec, low, 4, noise, tables=tabv)["enhanced"]
for cd, tol in ((None, 1e-3), ("fp16", 2e-2)):
    mv.compute_dtype = cd
    e = (mv.enhance(low.to(dev), 4, noise=torch.stack(noise)).cpu() - refv).abs().max().item()
    if not e < tol:
        fails.append(f"v_prediction enhance dtype={cd}: {e:.2e}")
mv.compute_dtype = None
print("2. v-prediction enhance (fp32 kernel + fused fp16 head):", "OK" if len(fails) == n0 else fails[n0:], flush=True)

# ---- 3. uint8 pre / post-processing on random sizes
n0 = len(fails)
for i in range(16):
    h0, w0, s = rng.randrange(17, 700), rng.randrange(17, 900), rng.choice([64, 128, 192, 256])
    img = np.random.default_rng(i).integers(0, 256, size=(h0, w0, 3), dtype=np.uint8)
    x_host, orig = M.preprocess_array(img, s)
    x_dev = M.preprocess_device(torch.from_numpy(img).to(dev), s)
    if not np.array_equal(x_dev.cpu().numpy(), x_host):
        fails.append(f"preprocess {h0}x{w0}->{s}")
    y = (np.random.default_rng(100 + i).random((1, 3, s, s), dtype=np.float32) * 2.4 - 1.2).astype(np.float32)
    out_host = M.postprocess_array(y, orig)
    out_dev = M.postprocess_device(torch.from_numpy(y).to(dev), orig)[0].cpu().numpy()
    if not np.array_equal(out_dev, out_host):
        fails.append(f"postprocess {s}->{h0}x{w0}: {np.abs(out_dev.astype(int) - out_host.astype(int)).max()} LSB")
print("3. uint8 pre/post-processing, 16 random sizes:", "OK" if len(fails) == n0 else fails[n0:], flush=True)

# ---- 4. gradients, small@192 ragged batch
n0 = len(fails)
import test_gpu_training as T
mt, sdt, spect = T._model("small", 192, dev)
g = torch.Generator().manual_seed(9)
lowt = torch.rand(3, 3, 192, 192, generator=g) * 2 - 1
normal = torch.rand(3, 3, 192, 192, generator=g) * 2 - 1
nz = torch.randn(3, 3, 192, 192, generator=g)
t = torch.tensor([0, 512, 999])
loss_ref, pred_ref, gref = T._ref_unet_grads(sdt, spect, lowt, normal, t, nz, loss="huber")
out = mt(lowt.to(dev), normal.to(dev), timesteps=t.to(dev), noise=nz.to(dev))
torch.nn.functional.huber_loss(out["noise_pred"], out["noise"]).backward()
worst = (0.0, "")
for k, p in mt.named_parameters():
    a, b_ = p.grad.double().cpu(), gref[k].double()
    l2 = ((a - b_).norm() / b_.norm().clamp_min(1e-30)).item()
    if l2 > worst[0]:
        worst = (l2, k)
if not worst[0] < 1e-2:
    fails.append(f"gradients small@192 B=3: rel L2 {worst[0]:.2e} at {worst[1]}")
print(f"4. whole-network gradients small@192 B=3 (huber): worst relative L2 {worst[0]:.1e} ({worst[1]})", flush=True)

print("failures:", fails)
sys.exit(1 if fails else 0)
