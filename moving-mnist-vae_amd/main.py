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
    # HIP model: the step scalars stay on the device (model._last_group) and are read back once, after the loop or
    # when the plot callback needs them -- the host never waits for the GPU in the middle of a step.
    pending = []                      # (position in the lists, scalar group)

    def materialise():
        sync = getattr(model, "_sync", None)
        if sync is not None and pending:       # data parallel: the logged values are means over ranks (one message)
            sync.reduce_step_scalars([g for _, g in pending])
        for pos, grp in pending:
            losses[pos], nlls[pos], kls[pos], mmds[pos] = grp.get(0), grp.get(1), grp.get(2), grp.get(3)
        pending.clear()

    for index, batch in enumerate(data_loader):
        model.train(True)                                                    # main.py:372
        image, target = prepare_batch(model, batch, device, args, data_mean, data_std)
        mu, logvar, encoding, reconstruction = model(image)                  # main.py:389
        if hasattr(model, "_last_group"):      # HIP model: scalars stay on the device until the end of the loop
            loss, nll_v, kl_v, mmd_v = model.loss(target, mu, logvar, encoding, reconstruction, device, args, deferred=True)
        else:
            loss, nll_v, kl_v, mmd_v = model.loss(target, mu, logvar, encoding, reconstruction, device, args)
        grp = getattr(model, "_last_group", None)
        if grp is not None and getattr(nll_v, "_g", None) is grp:
            pending.append((len(losses), grp))
            losses.append(None); nlls.append(None); kls.append(None); mmds.append(None)
        else:
            losses.append(loss.item())                                       # main.py:393
            nlls.append(nll_v)
            kls.append(kl_v)
            mmds.append(mmd_v)
        optimizer.zero_grad()                                                # main.py:397-399
        loss.backward()
        optimizer.step()
        if plot_callback is not None and index % plot_every == 0:            # main.py:401
            materialise()
            plot_callback(model=model, image=image, reconstruction=reconstruction, encoding=encoding,
                          epoch=epoch, index=index, directory=directory,
                          running={"nll": np.mean(nlls), "kl": np.mean(kls), "mmd": np.mean(mmds)})
    materialise()
    elapsed = time.time() - t0
    if not getattr(args, "quiet", False) and losses:
        print("Epoch={:d}; Loss={:0.5f} NLL={:.3f}; KL={:.3f}; MMD={:.3f}; time_tr={:.1f}s;".format(
            epoch, np.mean(losses), np.mean(nlls), np.mean(kls), np.mean(mmds), elapsed))
    return losses, nlls, kls, mmds


# ---------------------------------------------------------------------------------------------------------------------
# Drop-in contract around the loop: model-name grammar, checkpoint format, device-side input quantisation
# ---------------------------------------------------------------------------------------------------------------------
def _is_number(text) -> bool:
    try:
        float(text)
        return True
    except ValueError:
        return False


def select_model(args):
    """Model factory with the reference's name grammar and keyword mapping (main.py:41-147):
    ``pixelcnn_<layers>`` | ``<normal|categorical>_<vae|pixelvae>_<kl>_kl_<mmd>_mmd``.  Returns ``(model, model_params)`` with the reference's
    dict keys."""
    from .model import VAE
    parts = args.model.split("_")
    if len(parts) == 2:                                                         # main.py:57-66
        if parts[0] != "pixelcnn":
            raise AssertionError("It has to be only pixelcnn_2/4/7")
        if not _is_number(parts[1]):
            raise AssertionError("The number of layers has to be an int")
        only_pixelcnn = use_pixelcnn = True
        args.num_pixelcnn_layers = int(float(parts[1]))
        mp = {"model_name": "PixelCNN", "is_decoder_out_normal": False, "only_pixelcnn": True, "use_pixelcnn": True, "coeff_kl": 0., "coeff_mmd": 0.}
    else:                                                                       # main.py:69-87
        only_pixelcnn = False
        if not (len(parts) == 6 and "vae" in parts[1]):
            raise AssertionError("model name should be of the format normal_pixelvae_1_kl_10_mmd")
        if parts[1] not in ("pixelvae", "vae"):
            raise AssertionError("model should be vae or pixelvae")
        if not (_is_number(parts[2]) and _is_number(parts[4])):
            raise AssertionError("coefficients should be numeric")
        use_pixelcnn = parts[1] == "pixelvae"
        is_normal = parts[0] == "normal"
        mp = {"is_decoder_out_normal": is_normal, "only_pixelcnn": False, "use_pixelcnn": use_pixelcnn,
              "coeff_kl": float(parts[2]), "coeff_mmd": float(parts[4])}
        if use_pixelcnn:
            mp["model_name"] = "PixelVAE"
            if not is_normal and not (args.decoder_out_channels > args.input_channels):
                raise AssertionError("decoder_out_channels should be > input_channels when categorical_pixelvae else simply use normal_pixelvae")
        else:
            mp["model_name"] = "VAE"
    # main.py:91-93: normal_vae_* needs sigma_decoder != 0, normal_pixelvae_* needs sigma_decoder == 0
    if mp["is_decoder_out_normal"] and not (use_pixelcnn == (args.sigma_decoder == 0)):
        raise AssertionError("sigma_decoder should be 0 when using vae and non-zero when using pixelvae/pixelcnn")
    if use_pixelcnn:                                                            # main.py:95-100
        if not getattr(args, "num_pixelcnn_layers", 4) >= 2:
            raise AssertionError("num of pixelcnn layers should be greater than 2 when using pixelvae/pixelcnn")
        if getattr(args, "pixelcnn_activation", "ReLu") not in ("ReLu", "ELU"):
            raise AssertionError("Choose either Relu or ELU")
    mp.update({"input_channels": args.input_channels, "input_image_size": args.input_image_size,
               "intermediate_channels": args.intermediate_channels, "z_dimension": args.z_dimension,
               "sigma_decoder": args.sigma_decoder, "require_rsample": args.require_rsample,
               "num_pixelcnn_layers": getattr(args, "num_pixelcnn_layers", 4),
               "pixelcnn_activation": getattr(args, "pixelcnn_activation", "ReLu"), "coeff_nll": args.nll})
    if use_pixelcnn:                                                            # main.py:114-124
        mp["pixelcnn_out_channels"] = int(args.quantization)
        if not only_pixelcnn:
            mp["decoder_out_channels"] = args.input_channels if mp["is_decoder_out_normal"] else args.decoder_out_channels
        else:
            mp["decoder_out_channels"] = 0
    else:                                                                       # main.py:126-134
        mp["pixelcnn_out_channels"] = 0
        mp["decoder_out_channels"] = mp["input_channels"] if mp["is_decoder_out_normal"] else int(args.quantization)
    model = VAE(in_channels=mp["input_channels"], intermediate_channels=mp["intermediate_channels"],
                decoder_out_channels=mp["decoder_out_channels"], pixelcnn_out_channels=mp["pixelcnn_out_channels"],
                z_dimension=mp["z_dimension"], pixelcnn=mp["use_pixelcnn"], only_pixelcnn=mp["only_pixelcnn"],
                pixelcnn_layers=mp["num_pixelcnn_layers"], pixelcnn_activation=mp["pixelcnn_activation"],
                nll=mp["coeff_nll"], kl=mp["coeff_kl"], mmd=mp["coeff_mmd"], require_rsample=mp["require_rsample"],
                sigma_decoder=mp["sigma_decoder"], input_image_size=mp["input_image_size"],
                compute_dtype=getattr(args, "compute_dtype", None), blocks_per_stage=int(getattr(args, "blocks_per_stage", 1)))
    # build-defined extensions ride along in the parameter dict (and from there into the checkpoint)
    mp["compute_dtype"], mp["blocks_per_stage"] = model.compute_dtype, model.blocks_per_stage
    return model, mp


@torch.no_grad()
def generate_only_pixelcnn(sample, model, data_mean, data_std):
    """main.py:186-192: autoregressive sampling of a PixelCNN-only model, pixel by pixel (S * S forward passes; `sample` is updated in place)."""
    import torch.nn.functional as F
    out = None
    for i in range(model.input_image_size):
        for j in range(model.input_image_size):
            out = model.run_pixelcnn(sample)
            probs = F.softmax(out[:, :, i, j], dim=1)
            sample[:, :, i, j] = torch.multinomial(probs, 1).float() / data_std                      # (sic: the reference does not subtract the mean here)
    return out, sample


@torch.no_grad()
def generate(z_image, sample, model, data_mean, data_std):
    """main.py:195-202: autoregressive sampling of a PixelVAE's PixelCNN conditioned on the decoder image."""
    import torch.nn.functional as F
    output_ = None
    for i in range(model.input_image_size):
        for j in range(model.input_image_size):
            concat = torch.cat([z_image, sample], dim=1)
            output_ = model.run_pixelcnn(concat)
            probs = F.softmax(output_[:, :, i, j], dim=1)
            sample[:, :, i, j] = (torch.multinomial(probs, 1).float() - data_mean) / data_std
    return output_, sample


def save_checkpoint(model, optimizer, epoch, directory):
    """Same file and dict layout as main.py:522-526 (``latest-model.model`` = {'epoch','state_dict','optimizer'}), so
    checkpoints move between the reference and this package in both directions."""
    import os
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, "latest-model.model")
    ck = {"epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}
    # the reference network keeps the reference's three keys; a build-defined variant (deeper net, fp8 compute mode) adds what a loader
    # needs to rebuild it
    blocks, cdt = int(getattr(model, "blocks_per_stage", 1)), getattr(model, "compute_dtype", None)
    if blocks != 1 or cdt == "fp8":
        ck["mmvae"] = {"compute_dtype": cdt, "blocks_per_stage": blocks}
    torch.save(ck, path)
    return path


def checkpoint_variant(path, map_location=None):
    """The build-defined constructor keywords a checkpoint was saved with: {'compute_dtype', 'blocks_per_stage'} (reference
    checkpoints, which lack the entry: the reference network, default compute mode)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    v = dict(ck.get("mmvae") or {})
    return {"compute_dtype": v.get("compute_dtype"), "blocks_per_stage": int(v.get("blocks_per_stage", 1))}


def load_checkpoint(path, model, optimizer=None, map_location=None):
    """The resume path the reference lacks: restores parameters / BN buffers (and Adam moments for FusedAdam or
    torch.optim.Adam); returns the stored epoch."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    want = int((ck.get("mmvae") or {}).get("blocks_per_stage", 1))
    have = int(getattr(model, "blocks_per_stage", 1))
    if want != have:
        raise ValueError(f"checkpoint was saved from a net with blocks_per_stage={want}, the model has {have} "
                         "(build it with checkpoint_variant(path)['blocks_per_stage'])")
    model.load_state_dict(ck["state_dict"])
    if optimizer is not None and "optimizer" in ck:
        loader = getattr(optimizer, "load_flat_state", None)
        if loader is not None:
            loader(ck["optimizer"])
        else:
            optimizer.load_state_dict(ck["optimizer"])
    return ck.get("epoch", 0)


def quantise_frames(frames_u8, centres, data_mean, data_std):
    """Device-side input pipeline: uint8 frames -> k-means labels (int64) and normalised f32 image, one
    kernel (replaces ToTensor + kmeans.predict on the host, main.py:21-38, and the normalisation of main.py:383-387)."""
    from ._lib import check, lib, ptr
    f = frames_u8.contiguous()
    if f.dtype != torch.uint8 or not f.is_cuda:
        raise ValueError("quantise_frames expects a uint8 tensor on the GPU")
    c = torch.as_tensor(centres, dtype=torch.float32, device=f.device).contiguous()
    labels = torch.empty(f.shape, dtype=torch.int64, device=f.device)
    image = torch.empty(f.shape, dtype=torch.float32, device=f.device)
    check(lib().mmvae_quantise_normalise(ptr(f), f.numel(), ptr(c), c.numel(), float(data_mean), float(data_std), ptr(labels),
                                         ptr(image), torch.cuda.current_stream().cuda_stream), "mmvae_quantise_normalise")
    return labels, image


def clips_from_npz_array(arr) -> torch.Tensor:
    """File layout of movingmnist{train,test}.npz ['arr_0'] is (N, C, W, H) uint8.  The reference dataset transposes it to
    (N, H, W, C) (movingmnistdataset.py:15) and ToTensor turns every sample into (C, H, W) (main.py:29-31): net effect, the
    last two axes swap.  Returns the (N, C, H, W) uint8 tensor of clips the rest of the input pipeline works on (host memory)."""
    a = np.asarray(arr)
    if a.ndim != 4 or a.dtype != np.uint8:
        raise ValueError("expected a uint8 array of shape (N, C, W, H)")
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 1, 3, 2)))


class MovingMNISTClips:
    """Device-side replacement for ``DataLoader(MovingMNISTDataset(...), transform=ToTensor + kmeans.predict)``
    (movingmnistdataset.py:8-27, main.py:21-38, :491-500): the whole uint8 dataset sits in HBM (10 000 clips x 20 x 64 x 64 =
    819 MB), a batch is an index gather + ONE quantise kernel, and the iterator yields what the reference's loader yields --
    int64 k-means labels of shape (B, C*H*W) -- so ``train()`` consumes it unchanged.  No host work per step.

    source: a folder holding movingmnisttrain.npz / movingmnisttest.npz, or an (N, C, W, H) uint8 array in the file's layout.
    centres: the q k-means centres on the ToTensor scale [0, 1] (kmeans_dict['kmeans'].cluster_centers_.ravel())."""

    def __init__(self, source, centres, batch_size, device, train=True, shuffle=True, seed=None, drop_last=False):
        import os
        if isinstance(source, (str, os.PathLike)):
            path = os.path.join(source, "movingmnisttrain.npz" if train else "movingmnisttest.npz")
            if not os.path.isfile(path):
                raise FileNotFoundError(path)
            source = np.load(path)["arr_0"]
        self.device = torch.device(device)
        self.clips = clips_from_npz_array(source).to(self.device)
        self.train_data = self.clips                  # len(dataset.train_data) is read by main.py:506
        self.centres = torch.as_tensor(centres, dtype=torch.float32).reshape(-1)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self._gen = torch.Generator(device="cpu")
        if seed is not None:
            self._gen.manual_seed(int(seed))

    def __len__(self):
        n = self.clips.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.clips.shape[0]
        order = torch.randperm(n, generator=self._gen) if self.shuffle else torch.arange(n)
        order = order.to(self.device)
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            frames = self.clips.index_select(0, idx)
            labels, _ = quantise_frames(frames, self.centres, 0.0, 1.0)
            yield labels.view(labels.shape[0], -1)
