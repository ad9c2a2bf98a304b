"""CPU: the oracle restatement reproduces every golden the reference generated (bit-exact on
this torch build; tolerances below only absorb a different BLAS thread count on another host)."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from golden_util import CASE_NAMES, TRAJ_NAMES, LabelLoader, case_inputs, load, make_args

RTOL = 2e-6


def _close(a, b, rtol=RTOL, atol=1e-6):
    np.testing.assert_allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", CASE_NAMES)
def test_oracle_matches_reference_golden(name, oracle):
    O = oracle
    g, cfg = load(name)
    n, z, S = cfg["N"], cfg["z"], cfg["S"]
    spec = O.state_spec(cfg.get("in_ch", 1), z, cfg["out_ch"], S, cfg["rsample"])
    sd = O.filled_state(spec, seed=0)
    params = [k for k, _, kind in spec if kind in ("conv", "convT", "bias", "bn_w", "bn_b")]
    assert params == [str(s) for s in g["grad_names"]]
    for k in params:
        sd[k].requires_grad_(True)
    labels, image, categorical, target = case_inputs(O, cfg, int(g["labels_seed"]))
    eps = torch.from_numpy(g["eps"]).view(n, z, 1, 1) if cfg["rsample"] else None
    ts = torch.from_numpy(g["true_samples"])
    args = make_args(cfg)
    mu, lv, enc, rec = O.vae_forward(sd, image, eps, S, True, cfg["rsample"])
    loss, px, kl, mmd = O.vae_loss(target, mu, lv, enc, rec, ts, nll=1, kl=cfg["kl"], mmd=cfg["mmd"],
                                   sigma_decoder=cfg["sigma"], categorical=categorical,
                                   class_weight=args.data_ratio_of_labels)
    loss.backward()
    _close(loss.item(), g["loss"]); _close(px.item() / n, g["nll"]); _close(kl.item() / n, g["kl"], atol=1e-5)
    _close(mmd.item() / n, g["mmd"], atol=1e-5)
    _close(mu.detach().view(n, -1)[:8], g["mu"], rtol=1e-5, atol=1e-5)
    if cfg["rsample"]:
        _close(lv.detach().view(n, -1)[:8], g["logvar"], rtol=1e-5, atol=1e-5)
        _close(enc.detach().view(n, -1)[:8], g["encoding"], rtol=1e-5, atol=1e-5)
    _close(rec.detach()[:, :, ::8, ::8], g["recon_sub"], rtol=1e-4, atol=1e-4)
    _close(rec.detach().double().sum().item(), g["recon_sum"], rtol=1e-5, atol=1e-2)
    _close((rec.detach().double() ** 2).sum().item(), g["recon_sumsq"], rtol=1e-5)
    for i, k in enumerate(params):
        gr = sd[k].grad
        _close(gr.double().norm().item(), g["grad_norm"][i], rtol=1e-4, atol=1e-6)
        for j, idx in enumerate(g["grad_idx"][i]):
            if idx >= 0:
                _close(gr.flatten()[idx].item(), g["grad_val"][i][j], rtol=1e-3, atol=1e-3 * float(g["grad_norm"][i]) + 1e-7)
    bn = [k for k, _, kind in spec if kind in ("bn_rm", "bn_rv")]
    for i, k in enumerate(bn):
        _close(sd[k].double().norm().item(), g["bn_norm"][i], rtol=1e-5)
    # eval-mode reconstruction with the freshly updated running statistics
    with torch.no_grad():
        zfix = torch.from_numpy(g["eval_z"]).view(4, z, 1, 1)
        ev = O.get_reconstruction(OrderedDict((k, v.detach()) for k, v in sd.items()), zfix, S, False)
    _close(ev[:, :, ::8, ::8], g["eval_recon_sub"], rtol=1e-4, atol=1e-4)
    _close((ev.double() ** 2).sum().item(), g["eval_recon_sumsq"], rtol=1e-5)


@pytest.mark.parametrize("name", TRAJ_NAMES)
def test_train_loop_with_oracle_matches_reference_trajectory(name, oracle, pkg):
    """pkg.train (restatement of main.py:362-429) driving the oracle == driving the reference."""
    O = oracle
    g, cfg = load(name)
    spec = O.state_spec(1, cfg["z"], cfg["out_ch"], cfg["S"], cfg["rsample"])
    m = O.OracleVAE(1, 32, cfg["out_ch"], 2, cfg["z"], False, False, 4, "ReLu", 1, cfg["kl"], cfg["mmd"],
                    cfg["rsample"], cfg["sigma"], cfg["S"])
    m.load_state_dict(O.filled_state(spec, seed=0))
    opt = torch.optim.Adam(list(m.parameters()))
    torch.manual_seed(int(g["seed"]))
    out = pkg.train(m, LabelLoader(O, cfg["N"], cfg["S"], cfg["steps"], int(g["loader_seed0"])), opt,
                    torch.device("cpu"), make_args(cfg), epoch=0, data_mean=O.DATA_MEAN, data_std=O.DATA_STD)
    _close(out[0], g["loss"], rtol=1e-5); _close(out[1], g["nll"], rtol=1e-5)
    _close(out[2], g["kl"], rtol=1e-4); _close(out[3], g["mmd"], rtol=1e-4, atol=1e-5)
    sd = m.state_dict()
    for i, k in enumerate([str(s) for s in g["param_names"]]):
        _close(sd[k].double().norm().item(), g["param_norm"][i], rtol=1e-5)
    assert int(sd["encoder.bn1.num_batches_tracked"]) == int(g["nbt"]) == cfg["steps"]


def test_identities_from_reference_notebook(oracle):
    """The loss identities of test-output-models.ipynb cells 6/9 (no saved outputs there)."""
    O = oracle
    torch.manual_seed(3)
    mu, lv = torch.randn(8, 32), torch.randn(8, 32) * 0.3
    ref = torch.distributions.kl.kl_divergence(torch.distributions.Normal(mu, torch.exp(0.5 * lv)),
                                               torch.distributions.Normal(0., 1.)).sum()
    _close(O.kl_divergence(mu, lv).item(), ref.item(), rtol=1e-5)
    r, t = torch.randn(2, 1, 8, 8), torch.randn(2, 1, 8, 8)
    _close(O.gaussian_nll(r, t, 0.1).item(), -torch.distributions.Normal(r, 0.1).log_prob(t).sum().item(), rtol=1e-6)
    logits = torch.randn(3, 2, 8, 8); tg = (torch.rand(3, 8, 8) < 0.3).long()
    ce = torch.nn.functional.cross_entropy(logits, tg, reduction="none").sum()
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logits[:, 1] - logits[:, 0], tg.float(), reduction="sum")
    _close(ce.item(), bce.item(), rtol=1e-5)
