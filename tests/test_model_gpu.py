"""GPU: the HIP VAE against (a) the reference-generated goldens and (b) the CPU oracle run in the same process.

Tolerances (written here as the contract):
  f32 mode  (exact-f32 MFMA, f32 activations): loss/nll rel 2e-5, kl/mmd rel 1e-4, mu/logvar abs 2e-4,
            recon abs 2e-3 (post-BatchNorm values are O(1)).  Gradients: the fp32 CPU reference is itself up to ~1e-2
            away from the fp64 evaluation of the same graph on these inputs (binary images put whole groups of ReLU
            pre-activations exactly at zero, so the mask side is decided by the last bit: a 1e-7 input perturbation moves
            the reference's own gradients by 1.2e-3; at N=8 one flipped element of a 16K-element layer moves every
            upstream gradient by ~1%).  Which side of a tie is taken is a valid sub-gradient either way, so the network-
            level gradient gates allow for a few flips: golden table within 2e-2 per tensor norm, and
            test_f32_gradients_within_reference_uncertainty: per tensor |g_hip - g_ref32| <= 3*|g_ref32 - g_ref64| + 3e-2
            relative with both oracles evaluated live.  The tie-free gate is the layer-level one (tests/test_ops_gpu.py:
            every conv / dgrad / wgrad within 3e-5 of torch fp32; typical network-level agreement away from ties is 1e-6).
  bf16 mode (bf16 activations + bf16 MFMA, f32 accumulate/statistics): loss/nll rel 1e-3 (BASELINE.json:
            "ELBO within 1e-3 of CPU reference", relative), kl rel 5e-2, mu/logvar abs 0.15.  Gradients of this
            BatchNorm-heavy net are inherently noisy in bf16 (torch's own CPU bf16 autocast of the oracle is 3%..40%
            off per tensor at the filler weights), so the golden table is only a sanity bound there (norms within
            50%) and the real gate is test_bf16_gradient_noise_not_worse_than_torch_autocast: per tensor,
            |g_hip - g_fp32| <= 1.5 * |g_autocast - g_fp32| + 2% of |g_fp32|, evaluated live against the oracle.
"""
import importlib
import os
import sys
from collections import OrderedDict

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import CASE_NAMES, TRAJ_NAMES, LabelLoader, case_inputs, load, make_args  # noqa: E402

pytestmark = pytest.mark.gpu

TOL = {
    "f32": dict(loss=2e-5, kl=1e-4, mmd=2e-4, lat=2e-4, recon=2e-3, erecon=2e-3, gnorm=2e-2, gval=5e-2, bn=1e-4),
    # (bf16 gradients have no absolute bound here: they are gated per tensor against torch's own bf16 autocast of the oracle, for every
    # golden case, in test_bf16_gradient_noise_not_worse_than_torch_autocast, and at full size against the f32 mode and the minibatch
    # sampling noise in tests/test_config2_gpu.py)
    "bf16": dict(loss=1e-3, kl=5e-2, mmd=5e-2, lat=0.15, recon=0.25, erecon=0.6, gnorm=None, gval=None, bn=2e-2),
}


def _M():
    return importlib.import_module("moving-mnist-vae_amd.model")


def build_model(cfg, dt, O):
    M = _M()
    m = M.VAE(cfg.get("in_ch", 1), 32, cfg["out_ch"], 2, cfg["z"], False, False, 4, "ReLu", 1, cfg["kl"], cfg["mmd"], cfg["rsample"], cfg["sigma"],
              cfg["S"], compute_dtype=dt)
    spec = O.state_spec(cfg.get("in_ch", 1), cfg["z"], cfg["out_ch"], cfg["S"], cfg["rsample"])
    m.load_state_dict(O.filled_state(spec, seed=0))
    return m.to("cuda").train(), spec


def run_case(name, dt, O, verbose=False):
    g, cfg = load(name)
    t = TOL[dt]
    n, z, S = cfg["N"], cfg["z"], cfg["S"]
    dev = torch.device("cuda")
    m, spec = build_model(cfg, dt, O)
    labels, image, categorical, target = case_inputs(O, cfg, int(g["labels_seed"]))
    target = target.to(dev)
    if cfg["rsample"]:
        m.injected_eps = torch.from_numpy(g["eps"]).view(n, z, 1, 1).to(dev)
    m.injected_true_samples = torch.from_numpy(g["true_samples"]).to(dev)
    args = make_args(cfg, dev)
    mu, lv, enc, rec = m(image.to(dev))
    loss, nll, kl, mmd = m.loss(target, mu, lv, enc, rec, dev, args)
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    rep = OrderedDict()

    def rel(a, b):
        return abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)

    rep["loss"] = (rel(loss.item(), g["loss"]), t["loss"])
    rep["nll"] = (rel(nll, g["nll"]), t["loss"])
    rep["kl"] = (abs(kl - float(g["kl"])) / max(abs(float(g["kl"])), 1.0), t["kl"])
    rep["mmd"] = (abs(mmd - float(g["mmd"])) / max(abs(float(g["mmd"])), 1.0), t["mmd"])
    rep["mu"] = (float(np.abs(mu.detach().view(n, -1)[:8].cpu().numpy() - g["mu"]).max()), t["lat"])
    if cfg["rsample"]:
        rep["logvar"] = (float(np.abs(lv.detach().view(n, -1)[:8].cpu().numpy() - g["logvar"]).max()), t["lat"])
    rc = rec.detach().cpu()
    assert tuple(rc.shape) == (n, cfg["out_ch"], S, S)
    rep["recon_sub"] = (float(np.abs(rc[:, :, ::8, ::8].numpy() - g["recon_sub"]).max()), t["recon"])
    rep["recon_sumsq"] = (rel((rc.double() ** 2).sum().item(), g["recon_sumsq"]), 10 * t["loss"] if dt == "f32" else 2e-2)
    names = [str(s) for s in g["grad_names"]]
    worst_n, worst_v, wn_name = 0.0, 0.0, ""
    params = dict(m.named_parameters())
    for i, k in enumerate(names):
        gr = params[k].grad
        assert gr is not None, k
        gn = float(g["grad_norm"][i])
        gmax = float(g["grad_norm"].max())
        if gn < 1e-5 * gmax:
            # analytically-zero gradient (decoder.conv2.bias sits in front of a BatchNorm): the reference value is
            # rounding noise; only require ours to be noise-sized too
            assert gr.double().norm().item() < 1e-3 * gmax, k
            continue
        e = abs(gr.double().norm().item() - gn) / max(gn, 1e-6 * gmax)
        if e > worst_n:
            worst_n, wn_name = e, k
        flat = gr.flatten().cpu()
        for j, idx in enumerate(g["grad_idx"][i]):
            if idx >= 0:
                ev = abs(flat[idx].item() - float(g["grad_val"][i][j])) / max(gn, 1e-6 * float(g["grad_norm"].max()))
                worst_v = max(worst_v, ev)
    if t["gnorm"] is not None:
        rep["grad_norm[" + wn_name + "]"] = (worst_n, t["gnorm"])
        rep["grad_val"] = (worst_v, t["gval"])
    sd = m.state_dict()
    wb = 0.0
    for i, k in enumerate([str(s) for s in g["bn_names"]]):
        wb = max(wb, rel(sd[k].double().norm().item(), g["bn_norm"][i]))
    rep["bn_running"] = (wb, t["bn"])
    assert int(sd["encoder.bn1.num_batches_tracked"]) == 1
    # eval-mode reconstruction (model.py:353-362) from the freshly updated running statistics
    m.eval()
    with torch.no_grad():
        ev = m.get_reconstruction(torch.from_numpy(g["eval_z"]).view(4, z, 1, 1).to(dev)).cpu()
    # (eval mode has no batch re-normalisation, so bf16 deviations are not rescaled away: looser bound there)
    rep["eval_recon"] = (float(np.abs(ev[:, :, ::8, ::8].numpy() - g["eval_recon_sub"]).max()), t["erecon"])
    if verbose:
        print(f"[{dt}] {name}: " + "  ".join(f"{k}={v[0]:.2e}{'' if v[0] <= v[1] else '(!>' + format(v[1], '.0e') + ')'}" for k, v in rep.items()),
              flush=True)
    return rep


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("name", CASE_NAMES)
def test_hip_model_matches_reference_golden(name, dt, oracle):
    rep = run_case(name, dt, oracle)
    bad = {k: v for k, v in rep.items() if not (v[0] == v[0] and v[0] <= v[1])}
    assert not bad, bad


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("name", TRAJ_NAMES)
def test_train_loop_trajectory_matches_reference(name, dt, oracle, pkg):
    """pkg.train + HIP VAE + FusedAdam vs the 3-step trajectory of train + reference model + torch.optim.Adam."""
    O = oracle
    g, cfg = load(name)
    M = _M()
    dev = torch.device("cuda")
    m, spec = build_model(cfg, dt, O)
    opt = M.FusedAdam(list(m.parameters()))
    eps, ts = g["eps"], g["true_samples"]
    step = {"i": 0}

    class Inject:   # per-step noise injection in the order the reference consumed its CPU generator
        def __iter__(self_inner):
            for b in LabelLoader(O, cfg["N"], cfg["S"], cfg["steps"], int(g["loader_seed0"])):
                i = step["i"]
                m.injected_eps = torch.from_numpy(eps[i]).view(cfg["N"], cfg["z"], 1, 1).to(dev)
                m.injected_true_samples = torch.from_numpy(ts[i]).to(dev)
                step["i"] += 1
                yield b

    out = pkg.train(m, Inject(), opt, dev, make_args(cfg, dev), epoch=0, data_mean=O.DATA_MEAN, data_std=O.DATA_STD)
    torch.cuda.synchronize()
    tl = 3e-4 if dt == "f32" else 5e-3     # step 2-3 include the Adam update (sign-like, amplifies tiny gradient differences)
    np.testing.assert_allclose(out[0], g["loss"], rtol=tl)
    np.testing.assert_allclose(out[1], g["nll"], rtol=tl)
    np.testing.assert_allclose(out[2], g["kl"], rtol=20 * tl, atol=1e-3)
    sd = m.state_dict()
    for i, k in enumerate([str(s) for s in g["param_names"]]):
        if k == "decoder.conv2.bias":
            continue    # analytically-zero gradient (bias in front of a BatchNorm): Adam turns rounding noise into +-lr steps
        pn = float(g["param_norm"][i])
        assert abs(sd[k].double().norm().item() - pn) <= (2e-3 if dt == "f32" else 2e-2) * max(pn, 1e-3), k
    assert int(sd["decoder.bn2.num_batches_tracked"]) == cfg["steps"]
    # gradients landed in the flat buffer without copies, optimiser state has the torch.optim.Adam layout
    p0 = next(m.parameters())
    assert p0.grad is not None and p0.grad.data_ptr() == m._G[0].data_ptr()
    st = opt.state_dict()["state"]
    assert len(st) == len(list(m.parameters())) and set(st[0].keys()) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("name", ["c1_gauss", "gauss_s28", "gauss_rgb"])
def test_eval_mode_forward_matches_oracle(name, dt, oracle):
    """model.eval() forward (the reference's plotting branch, main.py:401-424: encoder AND decoder with the running statistics, every
    BatchNorm folded to scale / shift in one launch per entry point) against the oracle in eval mode, after one training step has moved the
    running statistics away from their defaults."""
    O = oracle
    g, cfg = load(name)
    n, z, S = cfg["N"], cfg["z"], cfg["S"]
    dev = torch.device("cuda")
    m, spec = build_model(cfg, dt, O)
    labels, image, categorical, target = case_inputs(O, cfg, int(g["labels_seed"]))
    eps = torch.from_numpy(g["eps"]).view(n, z, 1, 1)
    m.injected_eps = eps.to(dev)
    m.injected_true_samples = torch.from_numpy(g["true_samples"]).to(dev)
    mu, lv, enc, rec = m(image.to(dev))                       # one training forward: running statistics updated
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m.eval()
    m.injected_eps = eps.to(dev)
    with torch.no_grad():
        mu, lv, enc, rec = m(image.to(dev))
    torch.cuda.synchronize()
    omu, olv, oenc, orec = O.vae_forward(sd, image, eps, S, False, True)
    t = TOL[dt]
    assert (mu.cpu() - omu).abs().max().item() <= t["lat"], "mu"
    assert (lv.cpu() - olv).abs().max().item() <= t["lat"], "logvar"
    # (one step has moved the running statistics only a tenth of the way to the batch's: the eval-mode activations are O(10) and not
    # re-normalised layer by layer, so the reconstruction is compared relative to its own range; bf16 in relative L2)
    rc, scale = rec.cpu(), orec.abs().max().item()
    rel_max = (rc - orec).abs().max().item() / scale
    rel_l2 = (rc - orec).norm().item() / orec.norm().item()
    print(f"\neval forward {name} {dt}: recon max-abs error / range {rel_max:.2e}, rel-L2 {rel_l2:.2e} (range {scale:.1f})")
    if dt == "f32":
        assert rel_max <= 2e-5, rel_max          # measured 1.4e-6 .. 3.0e-6
    else:
        assert rel_l2 <= 8e-2, rel_l2            # measured 2.9e-2 .. 3.9e-2
    # the statistics are untouched by an eval forward
    sd2 = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(sd2[k].cpu(), v), k


@pytest.mark.parametrize("name", CASE_NAMES)
def test_bf16_gradient_noise_not_worse_than_torch_autocast(name, oracle):
    O = oracle
    g, cfg = load(name)
    n, z, S = cfg["N"], cfg["z"], cfg["S"]
    dev = torch.device("cuda")
    rs = bool(cfg["rsample"])
    spec = O.state_spec(cfg.get("in_ch", 1), z, cfg["out_ch"], S, rs)
    pn = [k for k, _, kind in spec if kind in ("conv", "convT", "bias", "bn_w", "bn_b")]
    labels, image, categorical, target = case_inputs(O, cfg, int(g["labels_seed"]))
    eps = torch.from_numpy(g["eps"]).view(n, z, 1, 1) if "eps" in g.files else None      # (require_rsample=False: no noise)
    ts = torch.from_numpy(g["true_samples"])
    args = make_args(cfg)

    def oracle_grads(autocast):
        sd = O.filled_state(spec, seed=0)
        for k in pn:
            sd[k].requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            mu, lv, enc, rec = O.vae_forward(sd, image, eps, S, True, rs)
        loss = O.vae_loss(target, mu.float(), None if lv is None else lv.float(), enc.float(), rec.float(), ts, nll=1, kl=cfg["kl"], mmd=cfg["mmd"],
                          sigma_decoder=cfg["sigma"], categorical=categorical, class_weight=args.data_ratio_of_labels)[0]
        loss.backward()
        return {k: sd[k].grad.detach().clone() for k in pn}

    g32, g16 = oracle_grads(False), oracle_grads(True)
    m, _ = build_model(cfg, "bf16", O)
    m.injected_eps, m.injected_true_samples = None if eps is None else eps.to(dev), ts.to(dev)
    mu, lv, enc, rec = m(image.to(dev))
    loss = m.loss(target.to(dev), mu, lv, enc, rec, dev, make_args(cfg, dev))[0]
    loss.backward()
    torch.cuda.synchronize()
    bad = {}
    for k, p in m.named_parameters():
        ref = g32[k]
        if ref.norm().item() < 1e-3 * max(v.norm().item() for v in g32.values()) * 1e-3:
            continue                                  # analytically-zero gradients (conv bias in front of a BatchNorm)
        e_hip = (p.grad.cpu() - ref).norm().item()
        e_auto = (g16[k] - ref).norm().item()
        # the LAST up-block's bn1 gradients are the known weak spot of the bf16 mode (sums of a masked data gradient that cancels to ~0
        # behind the output BatchNorm; csrc/conv_joinbwd.hip, tests/test_config2_gpu.py): twice autocast's own error there
        last_bn1 = k.startswith(f"decoder.uplayer{5 if S > 32 else 4}.0.bn1.")
        if e_hip > (2.0 if last_bn1 else 1.5) * e_auto + 0.02 * ref.norm().item():
            bad[k] = (e_hip / ref.norm().item(), e_auto / ref.norm().item())
    assert not bad, bad


@pytest.mark.parametrize("name", ["c1_gauss", "c1_cat_w1", "gauss_z128", "gauss_s28"])
def test_f32_gradients_within_reference_uncertainty(name, oracle):
    O = oracle
    g, cfg = load(name)
    n, z, S = cfg["N"], cfg["z"], cfg["S"]
    dev = torch.device("cuda")
    spec = O.state_spec(1, z, cfg["out_ch"], S, True)
    pn = [k for k, _, kind in spec if kind in ("conv", "convT", "bias", "bn_w", "bn_b")]
    labels = O.synthetic_labels(n, S, seed=int(g["labels_seed"]))
    categorical = cfg["out_ch"] > 1
    eps = torch.from_numpy(g["eps"]).view(n, z, 1, 1)
    ts = torch.from_numpy(g["true_samples"])
    w = make_args(cfg).data_ratio_of_labels

    def oracle_grads(dtype):
        sd = O.filled_state(spec, seed=0)
        sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
        for k in pn:
            sd[k].requires_grad_(True)
        image = O.normalise(labels, S).to(dtype)
        target = labels if categorical else image
        mu, lv, enc, rec = O.vae_forward(sd, image, eps.to(dtype), S, True, True)
        loss = O.vae_loss(target, mu, lv, enc, rec, ts.to(dtype), nll=1, kl=cfg["kl"], mmd=cfg["mmd"], sigma_decoder=cfg["sigma"],
                          categorical=categorical, class_weight=None if w is None else w.to(dtype))[0]
        loss.backward()
        return {k: sd[k].grad.detach().double() for k in pn}

    g32, g64 = oracle_grads(torch.float32), oracle_grads(torch.float64)
    m, _ = build_model(cfg, "f32", O)
    m.injected_eps, m.injected_true_samples = eps.to(dev), ts.to(dev)
    image = O.normalise(labels, S)
    target = (labels if categorical else image).to(dev)
    mu, lv, enc, rec = m(image.to(dev))
    loss = m.loss(target, mu, lv, enc, rec, dev, make_args(cfg, dev))[0]
    loss.backward()
    torch.cuda.synchronize()
    gmax = max(v.norm().item() for v in g32.values())
    bad = {}
    for k, p in m.named_parameters():
        ref = g32[k]
        if ref.norm().item() < 1e-5 * gmax:
            continue
        e_hip = (p.grad.cpu().double() - ref).norm().item() / ref.norm().item()
        band = (ref - g64[k]).norm().item() / ref.norm().item()
        if e_hip > 3 * band + 3e-2:
            bad[k] = (e_hip, band)
    assert not bad, bad


def test_full_size_properties_bf16(oracle):
    """BASELINE config-2 shape class (many frames, z=128, bf16) through size-independent properties: finite outputs,
    per-channel BatchNorm statistics of the reconstruction (zero mean / unit variance by construction, model.py:193),
    loss decomposition loss == nll + kl (mmd coefficient 0), and decreasing loss over a few Adam steps."""
    import types
    M = _M()
    pkg = importlib.import_module("moving-mnist-vae_amd")
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = M.VAE(1, 32, 1, 2, 128, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
    opt = M.FusedAdam(list(m.parameters()))
    N = 20 * 64
    labels = oracle.synthetic_labels(N, 64, seed=2).view(64, 20, 64, 64)
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    losses, nlls, kls, mmds = pkg.train(m, [labels] * 6, opt, dev, args, data_mean=oracle.DATA_MEAN, data_std=oracle.DATA_STD)
    assert all(np.isfinite(losses)), losses
    for a, b, c in zip(losses, nlls, kls):
        assert abs(a - (b + c)) <= 1e-4 * abs(a)
    assert losses[-1] < losses[0]
    image = oracle.normalise(labels.view(-1, 64, 64), 64).to(dev)
    mu, lv, enc, rec = m(image)
    assert rec.shape == (N, 1, 64, 64)
    gamma, beta = m.decoder.bn2.weight.item(), m.decoder.bn2.bias.item()      # output BatchNorm: recon ~ (beta, gamma^2)
    assert abs(rec.mean().item() - beta) < 1e-3 and abs(rec.var(unbiased=False).item() - gamma * gamma) < 1e-3


if __name__ == "__main__":
    from oracle import vae_oracle as O
    only = sys.argv[1:] or CASE_NAMES
    for dt in ("f32", "bf16"):
        for name in only:
            try:
                run_case(name, dt, O, verbose=True)
            except Exception as ex:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                print(f"[{dt}] {name}: EXCEPTION {ex!r}"[:400], flush=True)


def test_deferred_side_join_decoder_only_backward():
    """A backward pass that stops at the decoder (leaf encoding, no mmvae_encoder_bwd to order the side stream): the
    end-of-backward callback must still put the decoder's weight gradients in front of the caller's stream.  Compared with the
    same pass with the join inside mmvae_decoder_bwd (mmvae_net_defer_join(net, 0)): equal to the bit (every reduction of the pass has a
    fixed order; a missing join leaves whole tensors at zero or half summed)."""
    M = _M()
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    dev = torch.device("cuda")
    torch.manual_seed(3)
    m = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
    N = 256
    enc = torch.randn(N, 32, 1, 1, device=dev, requires_grad=True)
    w = torch.randn(N, 1, 64, 64, device=dev)

    def grads(defer):
        old = M._DEFER_JOIN
        M._DEFER_JOIN = defer
        try:
            m.zero_grad()
            enc.grad = None
            (m._decode(enc) * w).sum().backward()
            out = [p.grad.detach().clone() for p in m._dec_params] + [enc.grad.detach().clone()]
        finally:
            M._DEFER_JOIN = old
        return out

    ref = grads(False)
    for _ in range(3):
        got = grads(True)
        names = [n for n, _ in m.named_parameters() if n.startswith("decoder.")] + ["d_encoding"]
        bad = {n: (float((a - b).abs().max()), float(b.abs().max())) for n, a, b in zip(names, got, ref) if not torch.equal(a, b)}
        assert not bad, bad
    assert any(float(g.abs().max()) > 0 for g in ref)


def test_loss_scalars_off_the_critical_stream_and_no_reference_cycle():
    """train() takes the step's loss scalars (KL, MMD, NLL sums, reference model.py:385-406) off the caller's stream: the decoder's backward
    enqueues them on the net's side stream (VAE.loss(deferred=True), _LossFn.forward).  (1) The values are the ones the in-stream form
    returns, to the bit (same kernels, same inputs), for every step of a short run; (2) scalars whose backward never ran are still
    computed when read; (3) the bookkeeping holds no reference cycle with the model -- `del model` frees it (and its multi-GB workspace)
    without the cycle collector."""
    import gc
    import types
    import weakref
    M = _M()
    pkg = importlib.import_module("moving-mnist-vae_amd")
    dev = torch.device("cuda")
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    g = torch.Generator().manual_seed(5)
    batches = [(torch.rand((2, 20, 64, 64), generator=g) < 0.05).long() for _ in range(3)]

    def run(deferred):
        torch.manual_seed(11)
        m = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1, 1, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
        opt = M.FusedAdam(list(m.parameters()))
        if deferred:
            out = pkg.train(m, batches, opt, dev, args, data_mean=0.05, data_std=0.22)
            vals = [[float(v) for v in col] for col in out]
        else:
            vals = [[], [], [], []]
            for b in batches:
                image, target = pkg.main.prepare_batch(m, b, dev, args, 0.05, 0.22)
                mu, lv, enc, rec = m(image)
                loss, nll, kl, mmd = m.loss(target, mu, lv, enc, rec, dev, args)          # in-stream form, Python floats
                for col, v in zip(vals, (loss.item(), nll, kl, mmd)):
                    col.append(float(v))
                opt.zero_grad()
                loss.backward()
                opt.step()
        return m, opt, vals

    # (the first optimizer a process constructs stays referenced by frames torch keeps from its lazy torch._dynamo import: not ours to test)
    warm = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1, 1, True, 0.1, 64, compute_dtype="bf16").to(dev)
    M.FusedAdam(list(warm.parameters()))
    del warm
    gc.collect()
    gc.disable()
    try:
        m1, o1, side_vals = run(True)
        m2, o2, main_vals = run(False)
        assert side_vals == main_vals, (side_vals, main_vals)
        # (2) a loss whose backward never runs: the first read launches the kernels
        image, target = pkg.main.prepare_batch(m1, batches[0], dev, args, 0.05, 0.22)
        mu, lv, enc, rec = m1(image)
        loss, nll, kl, mmd = m1.loss(target, mu, lv, enc, rec, dev, args, deferred=True)
        assert m1.__dict__.get("_pending_loss") is not None
        v = float(nll)
        assert m1.__dict__.get("_pending_loss") is None and v == v and v > 0
        del loss, nll, kl, mmd, mu, lv, enc, rec, image, target
        # (3) no cycle
        refs = [weakref.ref(m1), weakref.ref(m2)]
        del m1, o1, m2, o2
        assert all(r() is None for r in refs), "the model survives `del`: a reference cycle keeps it (and its workspace) alive"
    finally:
        gc.enable()


def test_join_gradient_form_matches_the_reduce_apply_form():
    """decoder.uplayer4's backward has two forms (mmvae_net_set_join_grad): the default hands the masked join gradient down from the block above
    and evaluates dy2 / dys inside the two fused ConvTranspose2d backward passes (no bn_bwd_apply launch, no dy tensors); the other is reduce ->
    apply -> consumers.  Same per-element arithmetic (A g + B y + C in f32, rounded to bf16 once; the mask from the stored join output instead of
    the recomputed sum), same kernels downstream: the block's own gradients agree to a bf16 rounding, the gradients upstream of it to the
    perturbation that rounding leaves after the remaining bf16 layers -- far inside the bf16-vs-f32 gates of the golden tests."""
    M = _M()
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    dev = torch.device("cuda")
    torch.manual_seed(21)
    m = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
    x = ((torch.rand(64, 1, 64, 64, device=dev) < 0.05).float() - 0.05) / 0.22
    eps = torch.randn(64, 32, 1, 1, device=dev)

    def grads(enable):
        L.check(L.lib().mmvae_net_set_join_grad(m._h, int(enable)), "mmvae_net_set_join_grad")
        m.zero_grad()
        m.injected_eps = eps
        mu, lv, enc, rec = m(x)
        loss = ((rec - x) ** 2).sum() * 50.0 + (mu ** 2).sum() + (lv ** 2).sum()
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    try:
        a, b = grads(True), grads(False)
    finally:
        L.lib().mmvae_net_set_join_grad(m._h, 1)
        m.injected_eps = None
    rels = {}
    for n in a:
        rels[n] = float((a[n] - b[n]).norm()) / (float(b[n].norm()) + 1e-30)
    order = sorted(rels, key=lambda n: -rels[n])
    print("join-gradient form vs reduce/apply form, worst rel-L2: " + ", ".join(f"{n} {rels[n]:.2e}" for n in order[:6]))
    print("  decoder.uplayer4: " + ", ".join(f"{n.split('uplayer4.0.')[1]} {rels[n]:.2e}" for n in rels if "uplayer4" in n))
    # the block's own parameters see the two forms directly: one bf16 rounding of dy apart at most
    assert all(rels[n] <= 2.0 ** -8 for n in rels if "uplayer4" in n), {n: rels[n] for n in rels if "uplayer4" in n}
    # everything upstream of it (the rest of the decoder, the encoder) inherits that perturbation through up to a dozen bf16 layers
    assert rels[order[0]] <= 4e-2, (order[0], rels[order[0]])
    assert any(float(v.abs().max()) > 0 for v in a.values())
