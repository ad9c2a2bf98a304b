"""CPU oracle for the conv-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional ``torch`` fp32 ops, the algorithm of
praateekmahajan/moving-mnist-vae's ``model.py`` VAE (encoder -> reparameterise
-> decoder -> ELBO) so that the HIP product path can be checked against it on
the GPU box, where ``/root/reference`` does not exist.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import this module;
  * the product package never imports it and never falls back to it.

Parity status: PINNED.  ``oracle/make_golden.py`` (run in the build
container, where the reference is importable) checks every function here
bit-for-bit against the reference ``model.py`` with torch 2.10.0 CPU and writes
the fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks
the oracle against those fixtures without the reference.

All ``file:line`` citations are into the reference repository.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm2d default, model.py:30,34,61,66,95,137,162,173,202
BN_MOMENTUM = 0.1

# --------------------------------------------------------------------------
# Parameter inventory (state_dict order of the reference VAE, pixelcnn=False)
# --------------------------------------------------------------------------

def _bn_entries(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    return [
        (prefix + ".weight", (c,), "bn_w"),
        (prefix + ".bias", (c,), "bn_b"),
        (prefix + ".running_mean", (c,), "bn_rm"),
        (prefix + ".running_var", (c,), "bn_rv"),
        (prefix + ".num_batches_tracked", (), "bn_nbt"),
    ]


def decoder_has_uplayer5(input_image_size: int) -> bool:
    # model.py:169,191
    return input_image_size > 32


def state_spec(in_channels: int, z: int, out_channels: int, input_image_size: int,
               need_logvar: bool = True, blocks: int = 1) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(key, shape, kind) for every state_dict entry, in the reference's order.

    Encoder: model.py:89-112 (+ _make_layer :132-146, BasicBlock :26-37).
    Decoder: model.py:154-179 (+ _make_up_block :196-209, DeconvBottleneck :58-68).

    ``blocks`` > 1 is the BUILD-DEFINED deeper variant (BASELINE configs[3]); parity UNPINNED by the reference, which
    hard-codes one block per stage (:98-101, :164-170).  Encoder: exactly what ``_make_layer(blocks)`` builds (:140-144:
    extra BasicBlocks with stride 1 and no downsample).  Decoder: ``_make_up_block(num_layer)`` puts the extra blocks FIRST
    (:204-206) but builds them as DeconvBottleneck(in, out) whose 2x ConvTranspose2d main path cannot be added to its
    identity shortcut; here an extra block keeps the stage's input shape: conv1x1 -> BN -> ReLU -> conv3x3 -> BN, + x, ReLU.
    """
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    e = "encoder."
    s.append((e + "conv1.weight", (32, in_channels, 5, 5), "conv"))
    s += _bn_entries(e + "bn1", 32)
    inpl = 32
    for li, planes in enumerate((32, 64, 128, 256), start=1):
        for b in range(blocks):
            p = f"{e}layer{li}.{b}."
            s.append((p + "conv1.weight", (planes, inpl, 3, 3), "conv"))
            s += _bn_entries(p + "bn1", planes)
            s.append((p + "conv2.weight", (planes, planes, 3, 3), "conv"))
            s += _bn_entries(p + "bn2", planes)
            if b == 0:
                s.append((p + "downsample.0.weight", (planes, inpl, 1, 1), "conv"))
                s += _bn_entries(p + "downsample.1", planes)
            inpl = planes
    s.append((e + "conv_mu.weight", (z, 256, 1, 1), "conv"))
    if need_logvar:
        s.append((e + "conv_logvar.weight", (z, 256, 1, 1), "conv"))
    d = "decoder."
    s.append((d + "conv1.weight", (z, 128, 2, 2), "convT"))
    s += _bn_entries(d + "bn1", 128)
    cin = 128
    ups = [128, 64, 32, 16] + ([16] if decoder_has_uplayer5(input_image_size) else [])
    for ui, planes in enumerate(ups, start=1):
        for b in range(blocks - 1):                     # build-defined shape-preserving blocks, first in the stage
            p = f"{d}uplayer{ui}.{b}."
            s.append((p + "conv1.weight", (cin, cin, 1, 1), "conv"))
            s += _bn_entries(p + "bn1", cin)
            s.append((p + "conv2.weight", (cin, cin, 3, 3), "conv"))
            s += _bn_entries(p + "bn2", cin)
        p = f"{d}uplayer{ui}.{blocks - 1}."
        s.append((p + "conv1.weight", (planes, cin, 1, 1), "conv"))
        s += _bn_entries(p + "bn1", planes)
        s.append((p + "conv2.weight", (planes, planes, 4, 4), "convT"))
        s += _bn_entries(p + "bn2", planes)
        s.append((p + "upsample.0.weight", (cin, planes, 4, 4), "convT"))
        s += _bn_entries(p + "upsample.1", planes)
        cin = planes
    s.append((d + "conv2.weight", (out_channels, 16, 3, 3), "conv"))
    s.append((d + "conv2.bias", (out_channels,), "bias"))
    s += _bn_entries(d + "bn2", out_channels)
    return s


def adjust_for(input_image_size: int) -> int:
    # model.py:307-310
    if input_image_size > 32:
        return (64 - input_image_size) // 2
    return (32 - input_image_size) // 2


# --------------------------------------------------------------------------
# Deterministic, name-keyed parameter filler (shared by oracle, fixtures, tests)
# --------------------------------------------------------------------------

def _gen_for(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def filled_state(spec, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Non-trivial but well-conditioned values for every state entry.

    conv / convT weights ~ N(0, 1.4/sqrt(fan_in)); BN gamma ~ U[0.5,1.5],
    beta ~ N(0,0.1); running_mean ~ N(0,0.1), running_var ~ U[0.5,1.5]
    (so eval-mode paths are non-trivial); num_batches_tracked = 0.
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape, kind in spec:
        g = _gen_for(name, seed)
        if kind == "conv":
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * (1.4 / math.sqrt(fan_in))
        elif kind == "convT":
            # effective fan-in of a stride-s transposed conv: Cin * (k/s)^2
            k = shape[2]
            stride = 2 if k == 4 else 1
            fan_in = shape[0] * max(1, (k // stride)) ** 2 if k == 4 else shape[0]
            t = torch.randn(shape, generator=g) * (1.4 / math.sqrt(fan_in))
        elif kind == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif kind == "bn_w":
            t = torch.rand(shape, generator=g) + 0.5
        elif kind == "bn_b":
            t = torch.randn(shape, generator=g) * 0.1
        elif kind == "bn_rm":
            t = torch.randn(shape, generator=g) * 0.1
        elif kind == "bn_rv":
            t = torch.rand(shape, generator=g) + 0.5
        elif kind == "bn_nbt":
            t = torch.tensor(0, dtype=torch.long)
        elif kind == "mask":
            # "pixelcnn.layers.<i>.mask": type A for the first layer, B for the others (model.py:234-239)
            t = pixelcnn_mask("A" if name.split(".")[-2] == "0" else "B", shape[0], shape[1])
        else:  # pragma: no cover
            raise ValueError(kind)
        sd[name] = t
    return sd


def synthetic_labels(n: int, size: int, seed: int = 1234, p: float = 0.0521) -> torch.Tensor:
    """k-means(q=2)-like labels: i.i.d. Bernoulli(p) per pixel, int64 (n, size, size).
    Statistics from test-output-models.ipynb cell 2 (label mean 0.052)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return (torch.rand((n, size, size), generator=g) < p).long()


DATA_MEAN = 0.0521
DATA_STD = 0.2222


def normalise(labels: torch.Tensor, size: int, data_mean: float = DATA_MEAN,
              data_std: float = DATA_STD) -> torch.Tensor:
    # main.py:383-387
    return (labels.float().view(-1, 1, size, size) - data_mean) / data_std


# --------------------------------------------------------------------------
# Functional forward
# --------------------------------------------------------------------------

class _BNState:
    """Carries train/eval flag; in train mode updates running stats in-place
    exactly like nn.BatchNorm2d (momentum 0.1, unbiased running_var)."""

    def __init__(self, sd: Dict[str, torch.Tensor], training: bool):
        self.sd = sd
        self.training = training

    def __call__(self, x: torch.Tensor, prefix: str) -> torch.Tensor:
        sd = self.sd
        if self.training:
            sd[prefix + ".num_batches_tracked"].add_(1)
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"],
                            self.training, BN_MOMENTUM, BN_EPS)


def encoder_forward(sd, x: torch.Tensor, training: bool = True, taps: Optional[dict] = None):
    """VAE_Encoder.forward, model.py:114-130; BasicBlock.forward model.py:39-55."""
    bn = _BNState(sd, training)
    e = "encoder."
    x = F.conv2d(x, sd[e + "conv1.weight"], None, 2, 2)          # :115 (5x5 s2 p2, :94)
    x = F.relu(bn(x, e + "bn1"))                                  # :116-117
    if taps is not None:
        taps[e + "stem"] = x
    for li in range(1, 5):
        b = 0
        while f"{e}layer{li}.{b}.conv1.weight" in sd:                 # one block in the reference (:98-101); more: build-defined
            p = f"{e}layer{li}.{b}."
            stride = 2 if b == 0 else 1
            out = F.conv2d(x, sd[p + "conv1.weight"], None, stride, 1)    # :42 (conv3x3, stride 2 in block 0, :14,:98-101)
            out = F.relu(bn(out, p + "bn1"))                          # :43-44
            out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1)   # :46
            out = bn(out, p + "bn2")                                  # :47
            idn = x                                                   # :40
            if (p + "downsample.0.weight") in sd:
                idn = F.conv2d(x, sd[p + "downsample.0.weight"], None, 2, 0)   # :50 (conv1x1 stride 2, :20,:135-138)
                idn = bn(idn, p + "downsample.1")
            x = F.relu(out + idn)                                     # :52-53
            b += 1
        if taps is not None:
            taps[f"{e}layer{li}"] = x
    x = F.adaptive_avg_pool2d(x, (1, 1))                          # :123
    mu = F.conv2d(x, sd[e + "conv_mu.weight"])                    # :125
    logvar = None
    if (e + "conv_logvar.weight") in sd:
        logvar = F.conv2d(x, sd[e + "conv_logvar.weight"])        # :128
    return mu, logvar


def rsample(mu: torch.Tensor, logvar: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """VAE_Encoder.rsample, model.py:148-150: Normal(mu, exp(0.5 logvar)).rsample()
    == loc + eps * scale with eps ~ N(0,1) drawn by the caller."""
    return mu + eps * torch.exp(logvar * 0.5)


def decoder_forward(sd, z: torch.Tensor, input_image_size: int, training: bool = True,
                    taps: Optional[dict] = None):
    """VAE_Decoder.forward, model.py:181-194; DeconvBottleneck.forward model.py:70-85."""
    bn = _BNState(sd, training)
    d = "decoder."
    x = F.conv_transpose2d(z, sd[d + "conv1.weight"], None, 1, 0)          # :182 (k2 s1 p0, :159-161)
    x = F.relu(bn(x, d + "bn1"))                                            # :183-184
    n_up = 5 if decoder_has_uplayer5(input_image_size) else 4
    for ui in range(1, n_up + 1):
        b = 0
        while (f"{d}uplayer{ui}.{b}.upsample.0.weight") not in sd:          # build-defined shape-preserving blocks (blocks > 1)
            p = f"{d}uplayer{ui}.{b}."
            out = F.conv2d(x, sd[p + "conv1.weight"])
            out = F.relu(bn(out, p + "bn1"))
            out = F.conv2d(out, sd[p + "conv2.weight"], None, 1, 1)
            out = bn(out, p + "bn2")
            x = F.relu(out + x)
            b += 1
        p = f"{d}uplayer{ui}.{b}."
        out = F.conv2d(x, sd[p + "conv1.weight"])                           # :73
        out = F.relu(bn(out, p + "bn1"))                                    # :74-75
        out = F.conv_transpose2d(out, sd[p + "conv2.weight"], None, 2, 1)   # :77 (k4 s2 p1, :62-65)
        out = bn(out, p + "bn2")                                            # :78
        sc = F.conv_transpose2d(x, sd[p + "upsample.0.weight"], None, 2, 1) # :81 (:197-203)
        sc = bn(sc, p + "upsample.1")
        x = F.relu(out + sc)                                                # :82-83
        if taps is not None:
            taps[f"{d}uplayer{ui}"] = x
    x = F.conv2d(x, sd[d + "conv2.weight"], sd[d + "conv2.bias"], 1, 1)     # :193 (:172)
    x = bn(x, d + "bn2")                                                    # :193 (BN on the output)
    return x


def crop(x: torch.Tensor, adjust: int) -> torch.Tensor:
    # model.py:328-329
    if adjust != 0:
        return x[:, :, adjust:-adjust, adjust:-adjust]
    return x


def vae_forward(sd, x: torch.Tensor, eps: Optional[torch.Tensor], input_image_size: int,
                training: bool = True, require_rsample: bool = True, taps: Optional[dict] = None):
    """VAE.forward, model.py:316-342 with pixelcnn=None, only_pixelcnn=False."""
    mu, logvar = encoder_forward(sd, x, training, taps)
    if require_rsample:
        encoding = rsample(mu, logvar, eps)                       # :322
    else:
        encoding = mu                                             # :324
    recon = crop(decoder_forward(sd, encoding, input_image_size, training, taps),
                 adjust_for(input_image_size))
    return mu, logvar, encoding, recon


# --------------------------------------------------------------------------
# PixelCNN / PixelVAE: MaskedConv2d model.py:212-224, PixelCNN :227-255, the concat of VAE.forward :331-336
# --------------------------------------------------------------------------
IN_EPS = 1e-5        # nn.InstanceNorm2d default (affine=False, no running statistics: instance statistics in train AND eval mode)


def pixelcnn_spec(in_channels: int, intermediate_channels: int, out_channels: int, layers: int,
                  prefix: str = "pixelcnn.") -> List[Tuple[str, Tuple[int, ...], str]]:
    """state_dict entries of PixelCNN in the reference's order (model.py:234-241): per layer weight, bias and the registered
    `mask` buffer (:216); the InstanceNorm2d modules have neither parameters nor buffers."""
    s: List[Tuple[str, Tuple[int, ...], str]] = []
    for i in range(layers):
        cin = in_channels if i == 0 else intermediate_channels
        cout = out_channels if i == layers - 1 else intermediate_channels
        p = f"{prefix}layers.{i}."
        s.append((p + "weight", (cout, cin, 7, 7), "conv"))
        s.append((p + "bias", (cout,), "bias"))
        s.append((p + "mask", (cout, cin, 7, 7), "mask"))
    return s


def pixelcnn_mask(mask_type: str, cout: int, cin: int) -> torch.Tensor:
    """model.py:216-220: ones; the centre row from the centre (type A) / right of the centre (type B) on and every row below are zero."""
    m = torch.ones(cout, cin, 7, 7)
    m[:, :, 3, 3 + (mask_type == "B"):] = 0
    m[:, :, 4:] = 0
    return m


def instance_norm(x: torch.Tensor) -> torch.Tensor:
    return F.instance_norm(x, eps=IN_EPS)


def pixelcnn_forward(sd, x: torch.Tensor, layers: int, prefix: str = "pixelcnn.") -> torch.Tensor:
    """PixelCNN.forward, model.py:248-255 with activation "ReLu": x = IN(x); (masked conv 7x7 pad 3 -> IN -> ReLU) x (layers - 1);
    masked conv.  The reference multiplies weight.data by the mask in place in every forward (:222-223); the functional form uses the
    masked weight (same values; the gradient w.r.t. the stored weight at masked taps differs: see tests)."""
    x = instance_norm(x)                                                          # :249
    for i in range(layers):
        w = sd[f"{prefix}layers.{i}.weight"] * sd[f"{prefix}layers.{i}.mask"]     # :222
        x = F.conv2d(x, w, sd[f"{prefix}layers.{i}.bias"], stride=1, padding=3)   # :223
        if i < layers - 1:
            x = F.relu(instance_norm(x))                                          # :252-253
    return x


def get_reconstruction(sd, encoding: torch.Tensor, input_image_size: int, training: bool = False):
    """VAE.get_reconstruction / get_z_image, model.py:344-362 (pixelcnn=None)."""
    return crop(decoder_forward(sd, encoding, input_image_size, training), adjust_for(input_image_size))


# --------------------------------------------------------------------------
# Loss
# --------------------------------------------------------------------------

def kl_divergence(mu: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    # model.py:364-365 (and :9-10)
    return -0.5 * torch.sum(logvar - logvar.exp() - mu.pow(2) + 1)


def compute_kernel(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    # model.py:367-376 : exp(-mean_d((x-y)^2)/d)
    dim = x.size(1)
    diff = x.unsqueeze(1) - y.unsqueeze(0)
    return torch.exp(-(diff.pow(2).mean(2) / float(dim)))


def compute_mmd(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    # model.py:378-383  (sums, not means)
    return compute_kernel(x, x).sum() + compute_kernel(y, y).sum() - 2 * compute_kernel(x, y).sum()


def compute_mmd_tiled(x: torch.Tensor, y: torch.Tensor, tile: int = 512) -> torch.Tensor:
    """Same quantity as compute_mmd without materialising (N,N,d); used by the
    CPU baseline at N=5120 where the reference formulation needs 3 x 13.4 GB."""
    d = x.size(1)

    def ksum(a, b):
        tot = torch.zeros((), dtype=a.dtype)
        for i in range(0, a.size(0), tile):
            ai = a[i:i + tile]
            d2 = (ai * ai).sum(1, keepdim=True) + (b * b).sum(1)[None, :] - 2.0 * ai @ b.t()
            tot = tot + torch.exp(-(d2.clamp_min(0) / d) / float(d)).sum()
        return tot
    return ksum(x, x) + ksum(y, y) - 2 * ksum(x, y)


def gaussian_nll(recon: torch.Tensor, target: torch.Tensor, sigma: float) -> torch.Tensor:
    """-Normal(recon, sigma).log_prob(target).sum(), model.py:403.
    torch.distributions.Normal broadcasts the float scale to an fp32 tensor, so
    var and log(scale) are evaluated in fp32:
    log_prob = -((v-loc)^2)/(2 var) - log(scale) - log(sqrt(2 pi))."""
    scale = torch.as_tensor(sigma, dtype=recon.dtype)
    var = scale ** 2
    lp = -((target - recon) ** 2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))
    return -lp.sum()


def vae_loss(target, mu, logvar, encoding, recon, true_samples, *, nll=1, kl=1, mmd=0,
             sigma_decoder=0.1, categorical=False, class_weight=None, tiled_mmd=False):
    """VAE.loss, model.py:385-406.  ``true_samples`` replaces the in-line
    torch.randn(N, z) of model.py:395 (drawn by the caller so that parity tests
    can inject it).  Returns (loss, px_sum, kl_sum, mmd_sum) as 0-d tensors; the
    reference reports the last three as ``t.item() / N`` Python floats (:406)."""
    n = target.shape[0]
    klv = torch.tensor(0.)
    mmdv = torch.tensor(0.)
    if mu is not None and logvar is not None:
        klv = kl_divergence(mu, logvar)                                       # :390-391
    if encoding is not None:
        enc2 = encoding.view(-1, encoding.shape[1])
        mmdv = (compute_mmd_tiled if tiled_mmd else compute_mmd)(true_samples, enc2)   # :395-396
    if categorical:
        px = nll * F.cross_entropy(recon, target, reduction="none", weight=class_weight).sum()   # :400-401
    else:
        px = nll * gaussian_nll(recon, target, sigma_decoder)                 # :403
    loss = (px + kl * klv + mmd * mmdv) / n                                    # :405
    return loss, px, klv, mmdv


# --------------------------------------------------------------------------
# Thin nn.Module shell so the train loop / torch.optim can drive the oracle
# --------------------------------------------------------------------------

class _Node(torch.nn.Module):
    pass


class OracleVAE(torch.nn.Module):
    """Same constructor keywords, attributes and state_dict keys as model.py:258-310
    (hot-path subset: pixelcnn=False, only_pixelcnn=False).  forward/loss delegate to
    the functional restatement above.  RNG: like the reference, forward draws eps with
    the default CPU generator and loss draws true_samples, unless injected."""

    def __init__(self, in_channels, intermediate_channels, decoder_out_channels=1, pixelcnn_out_channels=2,
                 z_dimension=32, pixelcnn=True, only_pixelcnn=True, pixelcnn_layers=4,
                 pixelcnn_activation="ReLu", nll=1, kl=1, mmd=0, require_rsample=True,
                 sigma_decoder=0.1, input_image_size=64, blocks_per_stage=1):
        super().__init__()
        if pixelcnn or only_pixelcnn:
            raise NotImplementedError("oracle covers the plain conv-VAE path only")
        self.in_channels = in_channels
        self.z_dimensions = z_dimension
        self.decoder_out_channels = decoder_out_channels
        self.pixelcnn_out_channels = pixelcnn_out_channels
        self.num_pixelcnn_layers = pixelcnn_layers
        self.require_rsample = require_rsample
        self.nll, self.kl, self.mmd = nll, kl, mmd
        self.sigma_decoder = sigma_decoder
        self.input_image_size = input_image_size
        self.only_pixelcnn = only_pixelcnn
        self.pixelcnn = None
        self.adjust = adjust_for(input_image_size)
        self.tiled_mmd = False
        self._spec = state_spec(in_channels, z_dimension, decoder_out_channels, input_image_size, require_rsample, blocks_per_stage)
        init = filled_state(self._spec, seed=0)
        for name, shape, kind in self._spec:
            parts = name.split(".")
            node = self
            for comp in parts[:-1]:
                if comp not in node._modules:
                    node.add_module(comp, _Node())
                node = node._modules[comp]
            if kind in ("conv", "convT", "bias", "bn_w", "bn_b"):
                node.register_parameter(parts[-1], torch.nn.Parameter(init[name].clone()))
            else:
                node.register_buffer(parts[-1], init[name].clone())
        self.injected_eps = None
        self.injected_true_samples = None

    def _live_state(self):
        sd = OrderedDict()
        for k, v in self.named_parameters():
            sd[k] = v
        for k, v in self.named_buffers():
            sd[k] = v
        return sd

    def forward(self, x, sample=None):
        sd = self._live_state()
        mu, logvar = encoder_forward(sd, x, self.training)
        if self.require_rsample:
            eps = self.injected_eps
            if eps is None:
                eps = torch.empty(mu.shape, dtype=mu.dtype).normal_()
            encoding = rsample(mu, logvar, eps)
        else:
            encoding = mu
        recon = crop(decoder_forward(sd, encoding, self.input_image_size, self.training), self.adjust)
        return mu, logvar, encoding, recon

    def get_reconstruction(self, encoding, sample=None):
        return get_reconstruction(self._live_state(), encoding, self.input_image_size, self.training)

    get_z_image = get_reconstruction

    def loss(self, target, encoding_mu, encoding_logvar, encoding, reconstruction, device, args):
        ts = self.injected_true_samples
        if ts is None and encoding is not None:
            ts = torch.randn(target.shape[0], encoding.shape[1])
        categorical = self.decoder_out_channels > self.in_channels
        w = getattr(args, "data_ratio_of_labels", None) if categorical else None
        loss, px, klv, mmdv = vae_loss(target, encoding_mu, encoding_logvar, encoding, reconstruction, ts,
                                       nll=self.nll, kl=self.kl, mmd=self.mmd, sigma_decoder=self.sigma_decoder,
                                       categorical=categorical, class_weight=w, tiled_mmd=self.tiled_mmd)
        n = target.shape[0]
        return loss, px.item() / n, klv.item() / n, mmdv.item() / n       # :406
