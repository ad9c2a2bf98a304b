"""BASELINE configs[4]: fp8 weights + activations through the CDNA4 fp8 MFMA (compute_dtype="fp8").

What runs in fp8: the FORWARD convolutions of the deep layers (channel counts that are multiples of 64: encoder layer2.conv2 ...
layer4, decoder conv1 ... uplayer2), v_mfma_f32_16x16x32_fp8_fp8 on OCP e4m3 weights (static power-of-two scale per layer, absorbed
exactly by the BatchNorm that follows every such conv) and e4m3 activations (quantised behind the fused BN+ReLU), f32 accumulation,
f32 BatchNorm statistics; storage stays bf16 and the backward pass runs the bf16 kernels on the scaled weights (straight-through).

Tolerance, RE-STATED for this mode (BASELINE: "tolerance re-stated"): ELBO within 2e-3 (relative) of the f32 CPU oracle (bf16 mode:
1e-3; measured 1.2e-4 at 64 frames), reconstruction within 25 % relative L2 (measured 17 %: e4m3 carries 3 mantissa bits, 2^-4
relative rounding per operand, and the deep layers feed everything downstream), gradients only as a direction check against the
bf16 mode (median cosine over the tensors >= 0.6; measured 0.74 at 64 frames).  The scaled-weight / BatchNorm algebra itself is exact: running
statistics are checked against the oracle at the bf16 tolerance, and the 3-step training loss must decrease like the bf16 mode's.
"""
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

pytestmark = pytest.mark.gpu


def _M():
    return importlib.import_module("moving-mnist-vae_amd.model")


def test_fp8_forward_against_oracle(oracle):
    O = oracle
    M = _M()
    dev = torch.device("cuda")
    N, z, S = 64, 128, 64
    torch.manual_seed(1)
    m = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype="fp8")
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    image = O.normalise(O.synthetic_labels(N, S, seed=8), S)
    g = torch.Generator().manual_seed(3)
    eps, ts = torch.randn(N, z, 1, 1, generator=g), torch.randn(N, z, generator=g)
    osd = {k: v.clone() for k, v in state.items()}
    mu, lv, enc, rec = O.vae_forward(osd, image, eps, S, True, True)
    oloss = O.vae_loss(image, mu, lv, enc, rec, ts, nll=1, kl=1, mmd=0, sigma_decoder=0.1)[0].item()
    out = {}
    for dt in ("fp8", "bf16"):
        mm = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype=dt)
        mm.load_state_dict(state)
        mm.to(dev).train()
        mm.injected_eps, mm.injected_true_samples = eps.to(dev), ts.to(dev)
        hmu, hlv, henc, hrec = mm(image.to(dev))
        loss = mm.loss(image.to(dev), hmu, hlv, henc, hrec, dev, types.SimpleNamespace())[0]
        loss.backward()
        torch.cuda.synchronize()
        sd = mm.state_dict()
        rs = max((sd[k].cpu() - osd[k]).abs().max().item() / max(1.0, osd[k].abs().max().item())
                 for k in sd if k.endswith("running_var") or k.endswith("running_mean"))
        out[dt] = (abs(loss.item() - oloss) / abs(oloss), (hmu.detach().cpu() - mu).abs().max().item(),
                   (hrec.detach().cpu() - rec).norm().item() / rec.norm().item(), rs,
                   {k: p.grad.detach().cpu() for k, p in mm.named_parameters()})
    print(f"\nfp8 : ELBO rel {out['fp8'][0]:.3e}  mu abs {out['fp8'][1]:.3e}  recon rel-L2 {out['fp8'][2]:.3e}  running stats {out['fp8'][3]:.3e}")
    print(f"bf16: ELBO rel {out['bf16'][0]:.3e}  mu abs {out['bf16'][1]:.3e}  recon rel-L2 {out['bf16'][2]:.3e}  running stats {out['bf16'][3]:.3e}")
    assert out["fp8"][0] <= 2e-3                       # the re-stated ELBO tolerance of the fp8 mode
    assert out["fp8"][2] <= 0.25 and out["fp8"][1] <= 1.0
    assert out["fp8"][3] <= 0.1                        # scaled weights are absorbed by the BatchNorm: running statistics in true units
    # gradients: same direction as the bf16 mode's (straight-through backward on the same graph)
    cos = []
    for k, gb in out["bf16"][4].items():
        g8 = out["fp8"][4][k]
        if gb.norm().item() > 1e-6 * max(v.norm().item() for v in out["bf16"][4].values()):
            cos.append((g8 * gb).sum().item() / (g8.norm().item() * gb.norm().item() + 1e-30))
    print(f"fp8 vs bf16 gradient cosine: min {min(cos):.3f} median {sorted(cos)[len(cos) // 2]:.3f}")
    assert sorted(cos)[len(cos) // 2] >= 0.6


def test_fp8_mode_trains():
    pkg = importlib.import_module("moving-mnist-vae_amd")
    O = importlib.import_module("oracle.vae_oracle")
    M = _M()
    dev = torch.device("cuda")
    labels = O.synthetic_labels(32 * 20, 64, seed=6).view(32, 20, 64, 64)
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    curves = {}
    for dt in ("bf16", "fp8"):
        torch.manual_seed(0)
        m = M.VAE(1, 32, 1, 2, 128, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype=dt).to(dev).train()
        opt = M.FusedAdam(list(m.parameters()))
        torch.manual_seed(77)
        curves[dt] = np.array(pkg.train(m, [labels] * 30, opt, dev, args, data_mean=O.DATA_MEAN, data_std=O.DATA_STD)[0])
    a, b = curves["bf16"], curves["fp8"]
    print(f"\n30 steps: bf16 {a[0]:.1f} -> {a[-1]:.1f}; fp8 {b[0]:.1f} -> {b[-1]:.1f}; max rel diff {np.max(np.abs(a - b) / a):.3e}")
    assert np.all(np.isfinite(b)) and b[-1] < b[0]
    # measured 2.5e-3 ... 1.0e-2 over runs and kernel paths; the bf16 curve ITSELF ends between 365.0 k and 367.3 k (6e-3) depending on
    # the run / accumulation order (30 Adam steps amplify rounding noise), so the two trajectories agree to within that spread
    assert np.max(np.abs(a - b) / a) <= 2e-2
