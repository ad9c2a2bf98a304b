"""Generate tests/golden/pixel_*.npz (PixelCNN / PixelVAE) from the REFERENCE implementation.  Build-container only.

Run:  python oracle/make_pixel_golden.py            (needs /root/reference; never runs on the GPU box)

Like oracle/make_golden.py: builds the reference ``model.VAE`` with ``pixelcnn=True`` (``only_pixelcnn`` or not), loads the name-keyed
deterministic state of ``oracle.vae_oracle.filled_state``, runs forward / loss / backward on CPU fp32, REQUIRES the oracle's restatement
(``pixelcnn_forward`` + the concat of model.py:331-336) to agree bit for bit, and stores KB-scale goldens (data only).
Weight gradients are stored MASKED (grad * mask): the reference's gradient at a masked tap is that of an unmasked convolution with a zeroed
weight -- non-zero, and irrelevant, because model.py:222 zeroes the tap again before every use.
"""
from __future__ import annotations

import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import vae_oracle as O  # noqa: E402

CASES = OrderedDict([
    # only_pixelcnn: model "pixelcnn_<layers>" (main.py:57-64): input = the normalised image itself
    ("pixel_only_3", dict(only=True, in_ch=1, mid=16, dec_out=0, pix_out=2, layers=3, z=32, S=16, N=4, kl=0, mmd=0)),
    # categorical_pixelvae_1_kl_0_mmd with decoder_out_channels = 2 (main.py:75-77, 118-123)
    ("pixel_vae_cat", dict(only=False, in_ch=1, mid=16, dec_out=2, pix_out=2, layers=3, z=32, S=64, N=4, kl=1, mmd=0)),
    # normal_pixelvae: decoder_out_channels = input_channels = 1; four layers, wider
    ("pixel_vae_norm", dict(only=False, in_ch=1, mid=32, dec_out=1, pix_out=2, layers=4, z=32, S=32, N=3, kl=1, mmd=0)),
])


def full_spec(cfg):
    pix_in = cfg["in_ch"] if cfg["only"] else cfg["dec_out"] + cfg["in_ch"]
    spec = O.pixelcnn_spec(pix_in, cfg["mid"], cfg["pix_out"], cfg["layers"])
    if not cfg["only"]:
        spec = spec + O.state_spec(cfg["in_ch"], cfg["z"], cfg["dec_out"], cfg["S"], True)
    return spec


def main():
    sys.path.insert(0, REF)
    import model as refmodel
    os.makedirs(OUT, exist_ok=True)
    for name, cfg in CASES.items():
        spec = full_spec(cfg)
        state = O.filled_state(spec, seed=0)
        m = refmodel.VAE(cfg["in_ch"], cfg["mid"], cfg["dec_out"], cfg["pix_out"], cfg["z"], True, cfg["only"], cfg["layers"], "ReLu", 1, cfg["kl"],
                         cfg["mmd"], True, 0.0, cfg["S"])
        assert list(m.state_dict().keys()) == [k for k, _, _ in spec], "state_dict order / keys differ from the oracle's spec"
        m.load_state_dict(state)
        m.train()
        N, S, z = cfg["N"], cfg["S"], cfg["z"]
        labels = O.synthetic_labels(N, S, seed=77)
        image = O.normalise(labels, S)
        torch.manual_seed(11)
        eps = torch.randn(N, z, 1, 1)
        ts = torch.randn(N, z)
        args = types.SimpleNamespace(data_ratio_of_labels=torch.ones(cfg["pix_out"]))
        # ---- reference
        torch.manual_seed(11)               # forward draws eps (= the tensor above), loss draws true_samples next
        mu, lv, enc, rec = m(image)
        loss, px, kl, mmd = m.loss(labels, mu, lv, enc, rec, torch.device("cpu"), args)
        m.zero_grad()
        loss.backward()
        ref_grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        # ---- oracle
        sd = {k: v.clone() for k, v in state.items()}
        pn = [k for k, _, kind in spec if kind in ("conv", "convT", "bias", "bn_w", "bn_b")]
        for k in pn:
            sd[k].requires_grad_(True)
        if cfg["only"]:
            omu = olv = oenc = None
            orec = O.pixelcnn_forward(sd, image, cfg["layers"])
        else:
            omu, olv, oenc, dec = O.vae_forward(sd, image, eps, S, True, True)
            orec = O.pixelcnn_forward(sd, torch.cat([dec, image], dim=1), cfg["layers"])          # model.py:331-336 (training)
        oloss, opx, okl, ommd = O.vae_loss(labels, omu, olv, oenc, orec, ts if not cfg["only"] else None, nll=1, kl=cfg["kl"], mmd=cfg["mmd"],
                                           sigma_decoder=0.0, categorical=True, class_weight=args.data_ratio_of_labels)
        oloss.backward()
        assert torch.equal(orec, rec), name
        assert torch.equal(oloss.detach(), loss.detach()), (name, oloss.item(), loss.item())
        gold = dict(cfg=np.array(repr(cfg)), loss=np.float64(loss.item()), px=np.float64(px), kl=np.float64(kl), mmd=np.float64(mmd),
                    recon_sub=rec.detach()[:, :, ::4, ::4].numpy().copy(), recon_sum=np.float64(rec.double().sum().item()),
                    recon_sq=np.float64((rec.double() ** 2).sum().item()), eps=eps.numpy(), true_samples=ts.numpy())
        names, gnorm, gvals, gidx = [], [], [], []
        for k in pn:
            g = ref_grads[k]
            og = sd[k].grad
            if k.startswith("pixelcnn.") and k.endswith(".weight"):
                mask = state[k[:-6] + "mask"]
                g = g * mask
                og = og * mask
            assert torch.equal(og, g), (name, k, (og - g).abs().max().item())
            names.append(k)
            gnorm.append(g.double().norm().item())
            flat = g.flatten()
            nz = torch.nonzero(flat).flatten()
            idx = nz[torch.linspace(0, max(len(nz) - 1, 0), 4).long()] if len(nz) else torch.zeros(4, dtype=torch.long)
            gidx.append(idx.numpy())
            gvals.append(flat[idx].numpy())
        gold.update(grad_names=np.array(names), grad_norm=np.array(gnorm), grad_idx=np.stack(gidx), grad_val=np.stack(gvals))
        if not cfg["only"]:
            gold.update(mu=mu.detach().numpy().copy(), logvar=lv.detach().numpy().copy())
        # the reference has masked its stored weights in place (model.py:222): part of the state after a forward
        for i in range(cfg["layers"]):
            w = m.state_dict()[f"pixelcnn.layers.{i}.weight"]
            assert torch.equal(w, state[f"pixelcnn.layers.{i}.weight"] * state[f"pixelcnn.layers.{i}.mask"])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **gold)
        print(f"{name}: loss {loss.item():.6f} px {px:.6f} kl {kl:.6f}  [oracle == reference bit-exact; {len(pn)} gradients]", flush=True)


if __name__ == "__main__":
    main()
