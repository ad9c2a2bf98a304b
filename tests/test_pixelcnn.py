"""PixelCNN / PixelVAE (reference model.py:212-255, :283-300, :331-340).

CPU: the oracle's restatement (pixelcnn_forward + the concat of VAE.forward) reproduces the goldens oracle/make_pixel_golden.py generated
from the reference (bit-exact there).  GPU: the HIP model (mmvae_pixelcnn_fwd / _bwd behind _PixelFn, masked 7x7 convolutions over the 24 /
25 unmasked taps) against the same goldens, f32 mode tight, bf16 mode within the bf16 tolerances of the plain VAE tests.
Weight gradients are compared on the unmasked taps (the fixtures store grad * mask: the reference's gradient at a masked tap belongs to a
weight it zeroes before every use)."""
import ast
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["pixel_only_3", "pixel_vae_cat", "pixel_vae_norm"]


def _load(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return g, ast.literal_eval(str(g["cfg"]))


def _spec(O, cfg):
    pix_in = cfg["in_ch"] if cfg["only"] else cfg["dec_out"] + cfg["in_ch"]
    spec = O.pixelcnn_spec(pix_in, cfg["mid"], cfg["pix_out"], cfg["layers"])
    if not cfg["only"]:
        spec = spec + O.state_spec(cfg["in_ch"], cfg["z"], cfg["dec_out"], cfg["S"], True)
    return spec


@pytest.mark.parametrize("name", CASES)
def test_oracle_pixelcnn_matches_reference_golden(name, oracle):
    O = oracle
    g, cfg = _load(name)
    spec = _spec(O, cfg)
    sd = O.filled_state(spec, seed=0)
    pn = [k for k, _, kind in spec if kind in ("conv", "convT", "bias", "bn_w", "bn_b")]
    assert pn == [str(s) for s in g["grad_names"]]
    for k in pn:
        sd[k].requires_grad_(True)
    N, S, z = cfg["N"], cfg["S"], cfg["z"]
    labels = O.synthetic_labels(N, S, seed=77)
    image = O.normalise(labels, S)
    eps, ts = torch.from_numpy(g["eps"]), torch.from_numpy(g["true_samples"])
    if cfg["only"]:
        mu = lv = enc = None
        rec = O.pixelcnn_forward(sd, image, cfg["layers"])
    else:
        mu, lv, enc, dec = O.vae_forward(sd, image, eps, S, True, True)
        rec = O.pixelcnn_forward(sd, torch.cat([dec, image], dim=1), cfg["layers"])
    loss, px, kl, mmd = O.vae_loss(labels, mu, lv, enc, rec, None if cfg["only"] else ts, nll=1, kl=cfg["kl"], mmd=cfg["mmd"], sigma_decoder=0.0,
                                   categorical=True, class_weight=torch.ones(cfg["pix_out"]))
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["loss"]), rtol=2e-6)
    np.testing.assert_allclose(rec.detach()[:, :, ::4, ::4].numpy(), g["recon_sub"], rtol=2e-5, atol=2e-6)
    for i, k in enumerate(pn):
        gr = sd[k].grad
        if k.startswith("pixelcnn.") and k.endswith(".weight"):
            gr = gr * sd[k[:-6] + "mask"]
        np.testing.assert_allclose(gr.double().norm().item(), g["grad_norm"][i], rtol=2e-5, atol=1e-9)


def test_pixelcnn_mask_is_a_tap_prefix(oracle):
    """The property the kernels rely on: the unmasked taps of a type-A / type-B mask are exactly the first 24 / 25 positions of the 7x7 kernel."""
    for t, n in (("A", 24), ("B", 25)):
        m = oracle.pixelcnn_mask(t, 1, 1).flatten()
        assert m[:n].sum() == n and m[n:].sum() == 0


TOL = {"f32": dict(loss=2e-5, recon=2e-3, gnorm=2e-2, gval=5e-2), "bf16": dict(loss=1e-2, recon=0.35, gnorm=None, gval=None)}     # (cross entropy of logits behind 3-4 bf16 conv + InstanceNorm layers: measured 3.5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_hip_pixelcnn_matches_reference_golden(name, dt, oracle):
    O = oracle
    M = importlib.import_module("moving-mnist-vae_amd.model")
    g, cfg = _load(name)
    dev = torch.device("cuda")
    spec = _spec(O, cfg)
    state = O.filled_state(spec, seed=0)
    m = M.VAE(cfg["in_ch"], cfg["mid"], cfg["dec_out"], cfg["pix_out"], cfg["z"], True, cfg["only"], cfg["layers"], "ReLu", 1, cfg["kl"], cfg["mmd"], True,
              0.0, cfg["S"], compute_dtype=dt)
    assert list(m.state_dict().keys()) == [k for k, _, _ in spec]
    m.load_state_dict(state)
    m.to(dev).train()
    N, S, z = cfg["N"], cfg["S"], cfg["z"]
    labels = O.synthetic_labels(N, S, seed=77)
    image = O.normalise(labels, S).to(dev)
    m.injected_eps = torch.from_numpy(g["eps"]).to(dev)
    m.injected_true_samples = torch.from_numpy(g["true_samples"]).to(dev)
    args = types.SimpleNamespace(data_ratio_of_labels=torch.ones(cfg["pix_out"]))
    mu, lv, enc, rec = m(image)
    if cfg["only"]:
        assert mu is None and lv is None and enc is None
    loss, px, kl, mmd = m.loss(labels.to(dev), mu, lv, enc, rec, dev, args)
    loss.backward()
    torch.cuda.synchronize()
    t = TOL[dt]
    rel = abs(loss.item() - float(g["loss"])) / abs(float(g["loss"]))
    assert rel <= t["loss"], (rel, loss.item(), float(g["loss"]))
    assert abs(px - float(g["px"])) <= t["loss"] * abs(float(g["px"])) and abs(kl - float(g["kl"])) <= 5e-2 * abs(float(g["kl"])) + 1e-6
    rerr = np.abs(rec.detach().cpu()[:, :, ::4, ::4].numpy() - g["recon_sub"]).max() / (np.abs(g["recon_sub"]).max() + 1e-30)
    assert rerr <= t["recon"], rerr
    # the stored weights are masked after a forward, like the reference's (model.py:222)
    for i in range(cfg["layers"]):
        w = m.state_dict()[f"pixelcnn.layers.{i}.weight"].cpu()
        assert torch.equal(w, state[f"pixelcnn.layers.{i}.weight"] * state[f"pixelcnn.layers.{i}.mask"])
    params = dict(m.named_parameters())
    names = [str(s) for s in g["grad_names"]]
    assert names == [k for k, _ in m.named_parameters()]
    gmax = float(g["grad_norm"].max())
    worst_n = worst_v = 0.0
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        gn = float(g["grad_norm"][i])
        if k.startswith("pixelcnn.") and k.endswith(".weight"):
            mask = state[k[:-6] + "mask"].to(dev)
            assert float((gr * (1 - mask)).abs().max()) == 0.0, k            # masked taps: untouched
        if gn < 1e-5 * gmax:
            assert gr.double().norm().item() < 1e-3 * gmax, k
            continue
        if t["gnorm"] is None:
            continue                 # bf16: gated against torch's own bf16 autocast of the oracle below
        worst_n = max(worst_n, abs(gr.double().norm().item() - gn) / gn)
        flat = gr.flatten().cpu()
        for j, idx in enumerate(g["grad_idx"][i]):
            worst_v = max(worst_v, abs(flat[int(idx)].item() - float(g["grad_val"][i][j])) / gn)
    if t["gnorm"] is not None:
        assert worst_n <= t["gnorm"] and worst_v <= t["gval"], (worst_n, worst_v)
        return
    # bf16: per tensor |g_hip - g_fp32| <= 1.5 |g_autocast - g_fp32| + 2 % (the gate of tests/test_model_gpu.py; twice autocast's error for the
    # last up-block's bn1, the known weak spot of the bf16 mode)
    pn = names

    def oracle_grads(autocast):
        sd = O.filled_state(spec, seed=0)
        for k in pn:
            sd[k].requires_grad_(True)
        img = O.normalise(labels, S)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            if cfg["only"]:
                omu = olv = oenc = None
                orec = O.pixelcnn_forward(sd, img, cfg["layers"])
            else:
                omu, olv, oenc, dec = O.vae_forward(sd, img, torch.from_numpy(g["eps"]), S, True, True)
                orec = O.pixelcnn_forward(sd, torch.cat([dec.float(), img], dim=1), cfg["layers"])
        f = lambda v: None if v is None else v.float()
        O.vae_loss(labels, f(omu), f(olv), f(oenc), orec.float(), None if cfg["only"] else torch.from_numpy(g["true_samples"]), nll=1, kl=cfg["kl"],
                   mmd=cfg["mmd"], sigma_decoder=0.0, categorical=True, class_weight=torch.ones(cfg["pix_out"]))[0].backward()
        out = {}
        for k in pn:
            gr = sd[k].grad.detach().clone()
            if k.startswith("pixelcnn.") and k.endswith(".weight"):
                gr = gr * sd[k[:-6] + "mask"]
            out[k] = gr
        return out

    g32, g16 = oracle_grads(False), oracle_grads(True)
    bad = {}
    gmx = max(v.norm().item() for v in g32.values())
    for k in pn:
        ref = g32[k]
        if ref.norm().item() < 1e-6 * gmx:
            continue
        e_hip = (params[k].grad.cpu() - ref).norm().item()
        e_auto = (g16[k] - ref).norm().item()
        last_bn1 = k.startswith(f"decoder.uplayer{5 if S > 32 else 4}.0.bn1.")
        if e_hip > (2.0 if last_bn1 else 1.5) * e_auto + 0.02 * ref.norm().item():
            bad[k] = (e_hip / ref.norm().item(), e_auto / ref.norm().item())
    assert not bad, bad


@pytest.mark.gpu
def test_pixelvae_trains_and_samples(pkg, oracle):
    """A few FusedAdam steps of a PixelVAE through this package's train() (loss decreases, masked taps stay zero), eval-mode
    get_reconstruction with a `sample` (model.py:353-362), and one autoregressive pixel of the sampler (main.py:195-202)."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = M.VAE(1, 16, 2, 2, 32, True, False, 3, "ReLu", 1, 1, 0, True, 0.0, 32, compute_dtype="bf16").to(dev)
    opt = M.FusedAdam(list(m.parameters()))
    args = types.SimpleNamespace(data_ratio_of_labels=torch.ones(2), dataset="MovingMNIST", quiet=True)
    batches = [oracle.synthetic_labels(16, 32, seed=3).view(16, 1024)] * 12
    losses = pkg.train(m, batches, opt, dev, args, data_mean=oracle.DATA_MEAN, data_std=oracle.DATA_STD)[0]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    w = m.state_dict()["pixelcnn.layers.1.weight"]
    assert float((w * (1 - m.state_dict()["pixelcnn.layers.1.mask"])).abs().max()) == 0.0
    m.eval()
    with torch.no_grad():
        z = torch.randn(3, 32, 1, 1, device=dev)
        sample = torch.zeros(3, 1, 32, 32, device=dev)
        out = m.get_reconstruction(z, sample)
        assert tuple(out.shape) == (3, 2, 32, 32) and torch.isfinite(out).all()
        # (no causality assertion: the reference's InstanceNorm2d layers normalise every channel over the whole image, so a PixelCNN
        # built like model.py:227-255 sees later pixels through the statistics -- faithfully reproduced, not a property to test)
        s2 = sample.clone()
        s2[:, :, 20:, :] = 1.0
        assert not torch.equal(m.get_reconstruction(z, s2), out)
        # one step of the autoregressive sampler's inner loop (main.py:195-202)
        import torch.nn.functional as F
        probs = F.softmax(out[:, :, 0, 0], dim=1)
        assert torch.allclose(probs.sum(1), torch.ones(3, device=dev), atol=1e-5)


@pytest.mark.gpu
def test_pixelvae_gradients_land_in_one_flat_buffer(pkg, oracle):
    """Every backward node of a PixelVAE step (PixelCNN -> decoder -> encoder) must pick the SAME flat gradient buffer, step after step:
    FusedAdam then reads it in place (no per-parameter gather) and GradSync's buckets cover the PixelCNN's gradients too."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = M.VAE(1, 16, 2, 2, 32, True, False, 3, "ReLu", 1, 1, 0, True, 0.0, 32, compute_dtype="bf16").to(dev)
    opt = M.FusedAdam(list(m.parameters()))
    args = types.SimpleNamespace(data_ratio_of_labels=torch.ones(2), dataset="MovingMNIST", quiet=True)
    labels = oracle.synthetic_labels(8, 32, seed=5).view(8, 1024)
    for set_to_none in (True, False, False):
        image, target = m.prepare_batch(labels, dev, oracle.DATA_MEAN, oracle.DATA_STD, True)
        out = m(image)
        loss = m.loss(target, *out, dev, args)[0]
        opt.zero_grad(set_to_none=set_to_none)
        loss.backward()
        G = opt._flat_grads(m)
        assert any(G.data_ptr() == g.data_ptr() for g in m._G if g is not None), "gradients were gathered: the backward nodes wrote two buffers"
        for (_, p, off, n, _) in m._ptable:
            assert p.grad is not None and p.grad.data_ptr() == G.data_ptr() + 4 * off
        opt.step()


@pytest.mark.gpu
def test_pixelcnn_only_model_trains_through_train(pkg, oracle):
    """select_model's `pixelcnn_N` (main.py:50-58): no encoder, so prepare_batch has no workspace to stage into -- train() must still run."""
    dev = torch.device("cuda")
    torch.manual_seed(0)
    args = types.SimpleNamespace(data_ratio_of_labels=torch.ones(2), dataset="MovingMNIST", quiet=True, quantization=2, input_channels=1,
                                 z_dimension=32, sigma_decoder=0.0, input_image_size=32, intermediate_channels=16)
    M = importlib.import_module("moving-mnist-vae_amd.model")
    m = M.VAE(1, 16, 1, 2, 32, True, True, 3, "ReLu", 1, 1, 0, True, 0.0, 32, compute_dtype="f32").to(dev)
    opt = M.FusedAdam(list(m.parameters()))
    batches = [oracle.synthetic_labels(8, 32, seed=3).view(8, 1024)] * 8
    losses = pkg.train(m, batches, opt, dev, args, data_mean=oracle.DATA_MEAN, data_std=oracle.DATA_STD)[0]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    # uint8 labels take the same path
    img8, tgt8 = m.prepare_batch(batches[0].to(torch.uint8), dev, oracle.DATA_MEAN, oracle.DATA_STD, True)
    img64, _ = m.prepare_batch(batches[0], dev, oracle.DATA_MEAN, oracle.DATA_STD, True)
    assert torch.equal(img8, img64) and tgt8.dtype == torch.int64 and m._staged is None
