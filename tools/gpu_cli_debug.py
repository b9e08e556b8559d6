import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle
from oracle import hostio_ref
M = importlib.import_module("cv-diffusion-model_amd")
H = importlib.import_module("cv-diffusion-model_amd.hostio")
dev = torch.device("cuda:0")
for size in (64, 96):
    spec = oracle.make_spec("small", size)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    m = M.LowLightDiffusion(unet_variant="small", image_size=size); m.load_state_dict(sd); m = m.to(dev).eval()
    rng = np.random.default_rng(2)
    for (h, w) in [(48, 80), (100, 70)]:
        img = (rng.random((h, w, 3)) * 60).astype(np.uint8)
        x, orig = hostio_ref.preprocess_ref(img, size)
        g = torch.Generator().manual_seed(77)
        noise = [torch.randn(1, 3, size, size, generator=g) for _ in range(4)]
        ref = oracle.enhance_ref(sd, spec, torch.from_numpy(x), 4, noise)["enhanced"].numpy()
        out = m.enhance(torch.from_numpy(x).to(dev), 4, noise=torch.stack(noise)).cpu().numpy()
        print(size, (h, w), "float max diff", np.abs(out - ref).max(), "range", ref.min(), ref.max())
        a = hostio_ref.postprocess_ref(out, orig); b = hostio_ref.postprocess_ref(ref, orig)
        d = np.abs(a.astype(int) - b.astype(int)); print("   post max", d.max(), "frac", (d > 0).mean())
        # uint8 before resize back
        ua = np.clip((out[0].transpose(1, 2, 0) + 1) * 127.5, 0, 255).astype(np.uint8); ub = np.clip((ref[0].transpose(1, 2, 0) + 1) * 127.5, 0, 255).astype(np.uint8)
        print("   u8 before resize max", np.abs(ua.astype(int) - ub.astype(int)).max())
