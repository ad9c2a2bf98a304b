"""Train-step driver with the reference's ``main.train`` signature.

Mirrors the loop body of the reference ``main.py:362-429`` (normalise ->
forward -> loss -> bookkeeping -> zero_grad / backward / step) so that a user
of the reference can swap ``from main import train`` for this one.  The
plotting / wandb branch (``main.py:401-424``) is out of scope; a callback can
be attached through ``args.plot_callback`` instead.

Nothing here does arithmetic on the hot path when the model is the HIP
``VAE`` of this package: batch preparation, forward, loss, backward and the
optimiser step all land in the C-ABI library (see ``model.py``).  The generic
torch fallback below exists so the same loop can drive the CPU oracle / the
reference model in parity tests; it is never used for the HIP model.
"""
from __future__ import annotations

import time
from typing import List, Tuple

import numpy as np
import torch


def _is_categorical(model) -> bool:
    # main.py:381 -- cross entropy when a PixelCNN exists or the decoder emits
    # more channels than the input has.
    return model.pixelcnn is not None or model.decoder_out_channels > model.in_channels


def prepare_batch(model, batch, device, args, data_mean, data_std):
    """main.py:374-388: returns (image, target)."""
    if getattr(args, "dataset", "MovingMNIST") == "MNIST":
        batch = batch[0]
    size = model.input_image_size
    hook = getattr(model, "prepare_batch", None)
    if hook is not None:
        # HIP model: one fused kernel, labels -> normalised frames (+ targets)
        return hook(batch, device, data_mean, data_std, _is_categorical(model))
    frames = batch.to(device)
    image = (frames.float().view(-1, 1, size, size) - data_mean) / data_std
    if _is_categorical(model):
        target = frames.view(-1, size, size).to(device).long()
    else:
        target = image
    return image, target


def train(model, data_loader, optimizer, device, args, epoch=0, data_mean=0, data_std=1, plot_every=200,
          directory="output/") -> Tuple[List[float], List[float], List[float], List[float]]:
    """One epoch.  Returns per-step lists (loss, nll, kl, mmd) like main.py:429."""
    losses: List[float] = []
    nlls: List[float] = []
    kls: List[float] = []
    mmds: List[float] = []
    t0 = time.time()
    model.train(True)
    plot_callback = getattr(args, "plot_callback", None)
    for index, batch in enumerate(data_loader):
        model.train(True)                                                    # main.py:372
        image, target = prepare_batch(model, batch, device, args, data_mean, data_std)
        mu, logvar, encoding, reconstruction = model(image)                  # main.py:389
        loss, nll_v, kl_v, mmd_v = model.loss(target, mu, logvar, encoding, reconstruction, device, args)
        losses.append(loss.item())                                           # main.py:393
        nlls.append(nll_v)
        kls.append(kl_v)
        mmds.append(mmd_v)
        optimizer.zero_grad()                                                # main.py:397-399
        loss.backward()
        optimizer.step()
        if plot_callback is not None and index % plot_every == 0:            # main.py:401
            plot_callback(model=model, image=image, reconstruction=reconstruction, encoding=encoding,
                          epoch=epoch, index=index, directory=directory,
                          running={"nll": np.mean(nlls), "kl": np.mean(kls), "mmd": np.mean(mmds)})
    elapsed = time.time() - t0
    if not getattr(args, "quiet", False) and losses:
        print("Epoch={:d}; Loss={:0.5f} NLL={:.3f}; KL={:.3f}; MMD={:.3f}; time_tr={:.1f}s;".format(
            epoch, np.mean(losses), np.mean(nlls), np.mean(kls), np.mean(mmds), elapsed))
    return losses, nlls, kls, mmds
