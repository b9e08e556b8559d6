"""Print per-tensor gradient errors of the module-level backward tests (debug aid)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_training as T
from oracle import unet_ref
dev = torch.device("cuda:0")
M = T.M
for cin, cout, hw, split in [(32, 32, 16, 0), (96, 32, 16, 64)]:
    name = f"g_irb_{cin}_{cout}_{split}"
    blk = T.fill(M.InvertedResidualBlock(cin, cout, 128, concat_split=split), name + ".", dev)
    x = T.synth_input(name + ".x", (2, cin, hw, hw), -2, 2)
    te = T.synth_input(name + ".temb", (2, 128), -1, 1)
    w = T.check_module(blk, name, lambda sd, x, te: unet_ref.irb_forward(sd, name, x, te), [x, te], dev, tol=1e9)
    print(name, {k: f"{v:.2e}" for k, v in w.items()})
    print("  grad norms", {k: f"{p.grad.norm().item():.3e}" for k, p in blk.named_parameters()})
name = "g_attn_256_16"
at = T.fill(M.LinearAttention(256, 4), name + ".", dev)
x = T.synth_input(name + ".x", (2, 256, 16, 16), -2, 2)
print(name, {k: f"{v:.2e}" for k, v in T.check_module(at, name, lambda sd, x: unet_ref.linear_attention_forward(sd, name, x, 4), [x], dev, tol=1e9).items()})
for nm, mod, fn, shape in [("g_down_64", M.Downsample(64), unet_ref.downsample, (2, 64, 32, 32)), ("g_up_64", M.Upsample(64), unet_ref.upsample, (2, 64, 16, 16))]:
    m = T.fill(mod, nm + ".", dev)
    x = T.synth_input(nm + ".x", shape, -2, 2)
    print(nm, {k: f"{v:.2e}" for k, v in T.check_module(m, nm, lambda sd, x: fn(sd, nm, x), [x], dev, tol=1e9).items()})
