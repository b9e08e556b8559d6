"""Host-side I/O and the CLI counterparts of the reference's scripts."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from conftest import ROOT

M = importlib.import_module("cv-diffusion-model_amd")


def test_preprocess_postprocess_semantics():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(40, 56, 3), dtype=np.uint8)
    x, orig = M.preprocess_array(img, 64)
    assert x.shape == (1, 3, 64, 64) and x.dtype == np.float32 and orig == (40, 56)
    assert x.min() >= -1.0 and x.max() <= 1.0
    # normalise / denormalise at equal size: the reference truncates ((x+1)*127.5).astype(uint8)
    # (inference.py:128-129), so the round trip is within one LSB and never above the input
    sq = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    xs, o = M.preprocess_array(sq, 64)
    back = M.postprocess_array(xs, o).astype(np.int64)
    assert np.all(back <= sq) and np.all(sq - back <= 1)
    # out-of-range model outputs are clipped like the reference's np.clip(...).astype(uint8)
    big = np.full((1, 3, 64, 64), 3.0, dtype=np.float32)
    assert M.postprocess_array(big, (64, 64)).max() == 255


@pytest.mark.parametrize("shape,out", [((37, 53), (64, 64)), ((96, 80), (32, 48)), ((64, 64), (64, 64))])
def test_resize_geometry_matches_half_pixel_bilinear(shape, out):
    """cv2.INTER_LINEAR's geometry == interpolate(bilinear, align_corners=False, no antialias)."""
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    r = M.resize_bilinear(a, *out)
    t = F.interpolate(torch.from_numpy(a).permute(2, 0, 1)[None].float(), size=out, mode="bilinear", align_corners=False)
    ref = np.floor(t[0].permute(1, 2, 0).numpy() + 0.5)
    assert np.abs(r.astype(np.float64) - ref).max() <= 1.0


def test_checkpoint_layouts(tmp_path):
    spec = oracle.make_spec("small", 64)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    bare, trainer = tmp_path / "bare.pt", tmp_path / "trainer.pt"
    torch.save(dict(sd), bare)                                                    # export.py:151-155
    torch.save({"epoch": 3, "global_step": 77, "model_state_dict": dict(sd), "best_val_loss": 0.5,
                "optimizer_state_dict": {}, "config": {"lr": 1e-4}}, trainer)      # trainer.py:418-434
    for path in (bare, trainer):
        m = M.LowLightDiffusion(unet_variant="small", image_size=64)
        meta = M.load_checkpoint(m, str(path))
        assert all(torch.equal(v, sd[k]) for k, v in m.state_dict().items())
    assert meta == {"epoch": 3, "global_step": 77, "best_val_loss": 0.5}
    with pytest.raises(ValueError):
        M.extract_state_dict({"weights": 1})


def test_cli_flags_match_reference():
    """Same flag names as the reference's argparse surfaces (inference.py:30-62, benchmark.py:23-44)."""
    inf = open(os.path.join(ROOT, "scripts", "inference.py")).read()
    ben = open(os.path.join(ROOT, "scripts", "benchmark.py")).read()
    for flag in ["--input", "--output", "--checkpoint", "--model", "--format", "--variant", "--image_size", "--num_steps", "--device"]:
        assert f'"{flag}"' in inf, flag
    for flag in ["--model", "--format", "--image_size", "--batch_size", "--num_runs", "--warmup", "--num_steps", "--device", "--threads"]:
        assert f'"{flag}"' in ben, flag


@pytest.mark.gpu
@pytest.mark.parametrize("shape,size", [((37, 53), 64), ((200, 120), 64), ((64, 64), 64), ((300, 400), 128)])
def test_device_pre_post_processing_bit_exact(shape, size):
    """llie_preprocess_u8 / llie_postprocess_u8 vs the host implementation: byte work, so bit-exact."""
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, size=(2,) + shape + (3,), dtype=np.uint8)
    ref = np.concatenate([M.preprocess_array(im, size)[0] for im in imgs])
    got = M.preprocess_device(torch.from_numpy(imgs).to(dev), size).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, ref)
    x = (rng.random((2, 3, size, size), dtype=np.float32) * 2.6 - 1.3)  # includes out-of-range values to clip
    refp = np.stack([M.postprocess_array(x[i:i + 1], shape) for i in range(2)])
    gotp = M.postprocess_device(torch.from_numpy(x).to(dev), shape).cpu().numpy()
    assert gotp.dtype == np.uint8 and np.array_equal(gotp, refp)


@pytest.mark.parametrize("shape,size", [((37, 53), 64), ((200, 120), 64), ((64, 64), 64), ((96, 160), 128)])
def test_host_io_vs_oracle(shape, size):
    """Product host path (hostio.py, fp32 NumPy) against the oracle's restatement of scripts/inference.py:99-134
    (oracle/hostio_ref.py, float64): normalise / denormalise / clip / truncate / layout exact; the bilinear resize
    within one LSB (fp32 vs float64 at round-half-up ties), identical in > 99.9 % of the bytes."""
    from oracle import hostio_ref
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=shape + (3,), dtype=np.uint8)
    x, orig = M.preprocess_array(img, size)
    xr, origr = hostio_ref.preprocess_ref(img, size)
    assert orig == origr and x.shape == xr.shape and x.dtype == xr.dtype
    d = np.abs(x - xr) * 127.5
    assert d.max() <= 1.0 + 1e-4 and (d > 1e-4).mean() < 1e-2
    y = (rng.random((1, 3, size, size), dtype=np.float32) * 2.6 - 1.3)
    p, pr = M.postprocess_array(y, shape), hostio_ref.postprocess_ref(y, shape)
    dp = np.abs(p.astype(np.int64) - pr.astype(np.int64))
    assert p.shape == pr.shape and dp.max() <= 1 and (dp > 0).mean() < 5e-2  # near-.5 ties (rational scale factors) round apart in fp32 vs float64
    if shape == (size, size):  # no resize involved: byte-exact both ways
        assert np.array_equal(x, xr) and np.array_equal(p, pr)


@pytest.mark.gpu
# 96: `--image_size` off the multiples of 64 (the reference's CLI takes any multiple of 8).  There the host resize's uint8 rounding
# (parity-unpinned: cv2 is absent) differs from the oracle's restatement in more input pixels, which the network amplifies
# to two LSB in a handful of output pixels; through the Python API with identical inputs the two agree to one LSB in
# < 1 % of the bytes at either size (second assert of the test).
@pytest.mark.parametrize("size,max_lsb", [(64, 1), (96, 2)])
def test_cli_end_to_end(tmp_path, size, max_lsb):
    from PIL import Image
    from oracle import hostio_ref
    spec = oracle.make_spec("small", size)
    sd = oracle.synth_state_dict(oracle.param_shapes(spec))
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"epoch": 1, "model_state_dict": dict(sd)}, ckpt)
    rng = np.random.default_rng(2)
    src = tmp_path / "in"; src.mkdir()
    imgs = []
    for i, (h, w) in enumerate([(48, 80), (100, 70)]):
        imgs.append((rng.random((h, w, 3)) * 60).astype(np.uint8))
        Image.fromarray(imgs[-1]).save(src / f"dark{i}.png")
    dst = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "inference.py"), "--input", str(src), "--output", str(dst),
                        "--checkpoint", str(ckpt), "--variant", "small", "--image_size", str(size), "--num_steps", "4",
                        "--noise_seed", "77"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for i, (h, w) in enumerate([(48, 80), (100, 70)]):
        out = np.asarray(Image.open(dst / f"dark{i}.png"))
        assert out.shape == (h, w, 3) and out.dtype == np.uint8
        # pixel values: the whole file -> file path against the CPU oracle (preprocess -> 4-step enhance -> postprocess,
        # scripts/inference.py:99-145) on the same seeded CPU noise.  The fp32 engine is within 1e-3 of the oracle, so
        # the bytes agree except where a value sits on a truncation / rounding boundary (the denormalisation truncates, the
        # resize back to the original size rounds): never more than one LSB.
        x, orig = hostio_ref.preprocess_ref(imgs[i], size)
        g = torch.Generator().manual_seed(77)
        noise = [torch.randn(1, 3, size, size, generator=g) for _ in range(4)]
        ref = oracle.enhance_ref(sd, spec, torch.from_numpy(x), 4, noise)["enhanced"].numpy()
        want = hostio_ref.postprocess_ref(ref, orig)
        d = np.abs(out.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= max_lsb and (d > 0).mean() < 0.15, (
            f"file-to-file vs oracle: max {d.max()} LSB, {(d > 0).mean():.3%} bytes differ -- this bound includes the host "
            "resize's uint8 rounding, which is parity-unpinned (cv2 absent, the reference holds no resized fixture)")
        # The network alone, with the resize taken out of the comparison: the oracle is fed the exact tensor the product's
        # preprocessing produced and its output goes through the product's post-processing.  A regression in the ragged
        # edge-tile kernels at 96 x 96 cannot hide inside the looser whole-file bound above.
        xp, origp = M.preprocess_array(imgs[i], size)
        refp = oracle.enhance_ref(sd, spec, torch.from_numpy(xp), 4, noise)["enhanced"].numpy()
        wantp = M.postprocess_array(refp, origp)
        dn = np.abs(out.astype(np.int64) - wantp.astype(np.int64))
        assert dn.max() <= 1 and (dn > 0).mean() < 0.01, f"network only: max {dn.max()} LSB, {(dn > 0).mean():.3%} bytes differ"
    bare = tmp_path / "bare.pt"
    torch.save(dict(sd), bare)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "benchmark.py"), "--model", str(bare), "--format", "pytorch",
                        "--image_size", str(size), "--batch_size", "2", "--num_runs", "3", "--warmup", "1", "--device", "cuda"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert re.search(r"Mean latency:\s+[\d.]+ ms", r.stdout) and "images/s" in r.stdout
