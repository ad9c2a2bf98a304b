"""Drop-in ``VAE`` for the reference ``model.VAE`` (model.py:258-439), MI355X-native.

Same constructor keywords, attributes, ``forward`` / ``loss`` / helper signatures, ``state_dict`` keys and
parameter registration order as the reference; every arithmetic op of the hot path runs in hand-written HIP
kernels behind the C ABI of ``include/mmvae.h``.  PyTorch is used for device memory, the ``nn.Module`` /
autograd shell and ``torch.distributed`` only.  There is NO CPU fallback: without the built library, or
without a GPU, the model raises.

Layout: all parameters live in ONE flat f32 device buffer (``nn.Parameter``s are views into it, in the
reference's registration order), likewise gradients, Adam moments and BN running statistics -- so the
optimiser step is one kernel and the data-parallel all-reduce is two bucket-sized collectives.
"""
from __future__ import annotations

import ctypes
import math
import os
import weakref
from typing import List, Optional

import torch
from torch import nn

from ._lib import MmvaeError, check, lib, ptr

_DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "fp8": 2}


def _stream():
    return torch.cuda.current_stream().cuda_stream


_DEFER_JOIN = os.environ.get("MMVAE_DEFER_JOIN", "1") != "0"   # developer A/B switch

class _Scope(nn.Module):
    """Name-space node of the module tree (mirrors the reference's nesting so state_dict keys match)."""

    def __init__(self, owner=None, role=None):
        super().__init__()
        self.__dict__["_owner_ref"] = weakref.ref(owner) if owner is not None else None
        self.__dict__["_role"] = role

    def forward(self, *a, **k):  # encoder(x) / decoder(z), like VAE_Encoder / VAE_Decoder (model.py:114,181)
        owner = self._owner_ref() if self._owner_ref else None
        if owner is None or self._role is None:
            raise MmvaeError("this sub-module is a parameter name-space; call the VAE instead")
        return owner._encode(*a, **k) if self._role == "encoder" else owner._decode_full(*a, **k)

    def rsample(self, mu, logvar):   # VAE_Encoder.rsample, model.py:148-150
        owner = self._owner_ref() if self._owner_ref else None
        if owner is None or self._role != "encoder":
            raise AttributeError("rsample")
        return owner._rsample(mu, logvar)


# ------------------------------------------------------------------------------------------ autograd glue
class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        N = x.shape[0]
        z = model.z_dimensions
        mu = torch.empty((N, z, 1, 1), device=x.device, dtype=torch.float32)
        logvar = torch.empty_like(mu) if model.require_rsample else None
        training = bool(model.training)
        ws = model._workspace(N, training)
        # prepare_batch staged this very tensor (same storage, untouched since) into this workspace: no conversion pass over x
        st_ = model._staged
        staged = st_ is not None and st_[0]() is x and st_[1] == x._version and st_[2] == ws.data_ptr() and st_[3] == N
        model._staged = None
        fn = lib().mmvae_encoder_fwd_staged if staged else lib().mmvae_encoder_fwd
        check(fn(model._h, N, ptr(x), model._net_ptr(model._flat), ptr(model._bnf), ptr(model._bni), ptr(ws), ws.numel(),
                 ptr(mu), ptr(logvar), int(training), _stream()), "mmvae_encoder_fwd")
        ctx.model, ctx.N, ctx.training = model, N, training
        ctx.token = model._stamp("enc", training)
        if logvar is None:
            return mu
        return mu, logvar

    @staticmethod
    def backward(ctx, d_mu, d_logvar=None):
        model = ctx.model
        model._check_stamp("enc", ctx.token, ctx.training)
        z, N = model.z_dimensions, ctx.N
        dev = model._flat.device
        d_mu = torch.zeros((N, z), device=dev) if d_mu is None else d_mu.contiguous().float()
        if model.require_rsample:
            d_logvar = torch.zeros((N, z), device=dev) if d_logvar is None else d_logvar.contiguous().float()
        G = model._grad_target()
        G[model._poff:model._dec_off].zero_()
        ws = model._workspace(N, True)
        check(lib().mmvae_encoder_bwd(model._h, N, ptr(d_mu), ptr(d_logvar), model._net_ptr(model._flat), model._net_ptr(G), ptr(ws), ws.numel(), _stream()),
              "mmvae_encoder_bwd")
        if model._sync is not None:
            model._sync.bucket_ready(G, 0, model._dec_off)      # (a PixelCNN's gradients sit in front and are complete by now: it ran first)
        return (None, None) + model._grad_views(G, 0)


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, encoding, *params):
        N = encoding.shape[0]
        training = bool(model.training)
        recon = torch.empty((N, model.decoder_out_channels, model._dec_side, model._dec_side), device=encoding.device,
                            dtype=torch.float32)
        ws = model._workspace(N, training)
        check(lib().mmvae_decoder_fwd(model._h, N, ptr(encoding), model._net_ptr(model._flat), ptr(model._bnf), ptr(model._bni), ptr(ws), ws.numel(),
                                      ptr(recon), int(training), _stream()), "mmvae_decoder_fwd")
        ctx.model, ctx.N, ctx.training = model, N, training
        ctx.token = model._stamp("dec", training)
        ctx.need_denc = bool(ctx.needs_input_grad[1])
        model._last_recon = (recon.data_ptr(), ctx.token) if training else None
        return recon

    @staticmethod
    def backward(ctx, d_recon):
        model = ctx.model
        model._check_stamp("dec", ctx.token, ctx.training)
        N = ctx.N
        # Gaussian loss tail (see _LossFn.backward): the incoming "gradient" is the loss function's token -- the real one is evaluated
        # inside the output BatchNorm's backward from the saved conv output and the target, and never stored
        tail = model._pending_tail
        model._pending_tail = None
        fused_tail = (tail is not None and tail[5] == ctx.token and d_recon.stride(0) == 0 and d_recon.data_ptr() == tail[0].data_ptr())
        if not fused_tail:
            d_recon = d_recon.contiguous().float()
        d_enc = torch.empty((N, model.z_dimensions, 1, 1), device=d_recon.device, dtype=torch.float32) if ctx.need_denc else None
        G = model._grad_target()
        G[model._dec_off:].zero_()
        ws = model._workspace(N, True)
        # The decoder's weight gradients stay in flight on the net's side stream while the encoder backward is enqueued;
        # mmvae_encoder_bwd orders them before this stream again, and so does the end-of-backward callback below when the graph
        # holds no encoder (a decoder driven from a leaf encoding).  With a GradSync attached the decoder bucket is all-reduced
        # from a communication stream that waits for this stream AND the side stream (GradSync.bucket_ready), so the caller's
        # stream is not held up there either.
        defer = _DEFER_JOIN
        check(lib().mmvae_net_defer_join(model._h, int(defer)), "mmvae_net_defer_join")
        if fused_tail:
            _, target, sigma, coef, gscale, _ = tail
            check(lib().mmvae_decoder_bwd_gauss(model._h, N, ptr(target), float(sigma), float(coef), ptr(gscale), model._net_ptr(model._flat), model._net_ptr(G), ptr(ws),
                                                ws.numel(), ptr(d_enc), _stream()), "mmvae_decoder_bwd_gauss")
        else:
            check(lib().mmvae_decoder_bwd(model._h, N, ptr(d_recon), model._net_ptr(model._flat), model._net_ptr(G), ptr(ws), ws.numel(), ptr(d_enc), _stream()),
                  "mmvae_decoder_bwd")
        if defer:
            h, st = model._h, _stream()
            torch.autograd.Variable._execution_engine.queue_callback(lambda: check(lib().mmvae_net_join(h, st), "mmvae_net_join"))
        if model._sync is not None:
            model._sync.bucket_ready(G, model._dec_off, model._n_params, side_of=model if defer else None)
        pend = model.__dict__.get("_pending_loss")
        if pend is not None:
            # the step's loss scalars (_LossFn.forward): behind the decoder's weight gradients on the side stream -- every fork of this pass
            # ordered that stream behind the forward pass already, so no new wait is needed, and the stream idles here until the encoder's
            # weight gradients arrive
            model.__dict__["_pending_loss"] = None
            side = lib().mmvae_net_side_stream(model._h)
            pend[0](side if side else _stream())
            if side and not defer:
                h, st = model._h, _stream()
                torch.autograd.Variable._execution_engine.queue_callback(lambda: check(lib().mmvae_net_join(h, st), "mmvae_net_join"))
        return (None, d_enc) + model._grad_views(G, 1)


class _PixelFn(torch.autograd.Function):
    """PixelCNN.forward (model.py:248-255) through mmvae_pixelcnn_fwd / _bwd."""

    @staticmethod
    def forward(ctx, model, x, *params):
        N, S = x.shape[0], x.shape[2]
        x = x.contiguous().float()
        with torch.no_grad():
            model._flat[:model._poff].mul_(model._pix_mask)          # model.py:222: weight.data *= mask in every forward
        out = torch.empty((N, model.pixelcnn_out_channels, S, S), device=x.device, dtype=torch.float32)
        ws = model._pixel_workspace(N, S)
        check(lib().mmvae_pixelcnn_fwd(model._hp, N, S, ptr(x), ptr(model._flat), ptr(ws), ws.numel(), ptr(out), _stream()), "mmvae_pixelcnn_fwd")
        ctx.model, ctx.N, ctx.S = model, N, S
        ctx.token = model._stamp("pix", True)
        ctx.need_dx = bool(ctx.needs_input_grad[1])
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, d_out):
        model, N, S = ctx.model, ctx.N, ctx.S
        if model._stamps.get(("pix", True)) != ctx.token:
            raise MmvaeError("the saved activations of this PixelCNN forward were overwritten by a later forward")
        (x,) = ctx.saved_tensors
        d_out = d_out.contiguous().float()
        G = model._grad_target()
        G[:model._poff].zero_()
        d_x = torch.empty_like(x) if ctx.need_dx else None
        ws = model._pixel_workspace(N, S)
        check(lib().mmvae_pixelcnn_bwd(model._hp, N, S, ptr(x), ptr(d_out), ptr(model._flat), ptr(G), ptr(ws), ws.numel(), ptr(d_x), _stream()),
              "mmvae_pixelcnn_bwd")
        tab = [e for e in model._ptable if e[2] < model._poff]
        chunks = G[:model._poff].split([e[3] for e in tab])
        return (None, d_x) + tuple(c.view(e[4]) for c, e in zip(chunks, tab))


class _StepScalars:
    """The four scalars of one step (loss, nll/N, kl/N, mmd/N) as a device tensor, copied to the host on first use.  `join` (optional):
    orders the stream that produced them (the net's side stream, _LossFn) before the current one -- called before any read."""
    __slots__ = ("dev", "vals", "join")

    def __init__(self, dev_tensor, join=None):
        self.dev, self.vals, self.join = dev_tensor, None, join

    def ready(self):
        """The device tensor, ordered before the current stream."""
        if self.join is not None:
            self.join()
            self.join = None
        return self.dev

    def get(self, i):
        if self.vals is None:
            self.vals = self.ready().tolist()    # the only device->host copy (and synchronisation) of the step
        return self.vals[i]


class DeferredScalar:
    """Float-like view of one step scalar.  ``VAE.loss`` returns these instead of Python floats so that the host does
    not have to wait for the forward pass before it enqueues the backward pass (the reference synchronises three
    times per step with ``.item()``); any arithmetic, comparison, formatting or ``float()`` reads the value."""
    __slots__ = ("_g", "_i")

    def __init__(self, group, index):
        self._g, self._i = group, index

    def __float__(self):
        return self._g.get(self._i)

    item = __float__

    def __repr__(self):
        return repr(float(self))

    def __format__(self, spec):
        return format(float(self), spec)

    def __bool__(self):
        return bool(float(self))

    def __hash__(self):
        return hash(float(self))

    def __array__(self, dtype=None, copy=None):
        import numpy as _np
        return _np.asarray(float(self), dtype=dtype)


def _forward_float(name, reflected=False):
    def op(self, *other):
        f = getattr(float, name)
        return f(float(self), *[float(o) if isinstance(o, DeferredScalar) else o for o in other])
    op.__name__ = name
    return op


for _n in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__", "__truediv__", "__rtruediv__", "__pow__", "__rpow__",
           "__neg__", "__pos__", "__abs__", "__lt__", "__le__", "__gt__", "__ge__", "__eq__", "__ne__", "__round__", "__int__",
           "__floordiv__", "__rfloordiv__", "__mod__", "__rmod__"):
    setattr(DeferredScalar, _n, _forward_float(_n))


class _RsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = mu.contiguous(), logvar.contiguous(), eps.contiguous()
        enc = torch.empty_like(mu)
        check(lib().mmvae_rsample_fwd(ptr(mu), ptr(logvar), ptr(eps), ptr(enc), mu.numel(), _stream()), "mmvae_rsample_fwd")
        ctx.save_for_backward(logvar, eps)
        return enc

    @staticmethod
    def backward(ctx, d_enc):
        logvar, eps = ctx.saved_tensors
        d_enc = d_enc.contiguous()
        d_mu, d_lv = torch.empty_like(d_enc), torch.empty_like(d_enc)
        check(lib().mmvae_rsample_bwd(ptr(d_enc), ptr(logvar), ptr(eps), ptr(d_mu), ptr(d_lv), d_enc.numel(), _stream()), "mmvae_rsample_bwd")
        return d_mu, d_lv, None


class _LossFn(torch.autograd.Function):
    """VAE.loss arithmetic (model.py:385-405): KL + MMD + Gaussian-NLL | weighted CE, all reductions on device."""

    @staticmethod
    def forward(ctx, model, target, mu, logvar, encoding, recon, true_samples, weight):
        L = lib()
        N = target.shape[0]
        dev = recon.device
        acc = torch.zeros(4, dtype=torch.float64, device=dev)      # px, kl, mmd
        out = torch.empty(4, dtype=torch.float32, device=dev)      # loss, px/N, kl/N, mmd/N
        # `model._loss_side` (set by VAE.loss(deferred=True) inside a train step): the loss scalars are logged values -- no gradient kernel
        # reads them -- so their kernels (KL, MMD, NLL / CE sums, the combine) leave the critical stream: they are enqueued on the net's side
        # stream by the decoder's backward pass, behind its weight gradients, where that stream idles while the caller's runs the latency-bound
        # kernels around the latent code (_DecoderFn.backward); the backward pass's own join (mmvae_encoder_bwd / the end-of-backward
        # mmvae_net_join) orders them before the caller's stream again.  Never launched by then (no decoder backward in the graph): the first
        # read of the scalars, or the next loss, launches them on the caller's stream (VAE._flush_pending_loss).
        side = bool(model.__dict__.get("_loss_side")) and model._h is not None
        model.__dict__["_loss_side"] = False
        model._flush_pending_loss()
        base = acc.data_ptr()
        recon = recon.contiguous()
        categorical = model.pixelcnn is not None or model.decoder_out_channels > model.in_channels      # model.py:398
        if mu is not None and logvar is not None:
            mu, logvar = mu.contiguous(), logvar.contiguous()
        scratch = None
        if encoding is not None:
            encoding = encoding.contiguous()
            scratch = torch.empty(2 * N, dtype=torch.float32, device=dev)
        target = target.contiguous()
        nll_c, kl_c, mmd_c, sigma = float(model.nll), float(model.kl), float(model.mmd), float(model.sigma_decoder)
        # the closure keeps every operand alive until it is dropped (VAE._loss_keep) -- as detached views: a tensor with a grad_fn would keep
        # the step's autograd graph, whose nodes hold the model, in a reference cycle with it (45 GB workspaces waiting for the cycle collector)
        det = lambda t: None if t is None else t.detach()
        k_mu, k_lv, k_enc, k_rec, k_tgt, k_ts, k_w = det(mu), det(logvar), det(encoding), det(recon), det(target), det(true_samples), det(weight)

        def launch(st):
            if k_mu is not None and k_lv is not None:
                check(L.mmvae_kl_fwd(ptr(k_mu), ptr(k_lv), k_mu.numel(), base + 8, st), "mmvae_kl_fwd")
            if k_enc is not None:
                check(L.mmvae_mmd_fwd(ptr(k_ts), ptr(k_enc), N, k_enc.shape[1], ptr(scratch), base + 16, st), "mmvae_mmd_fwd")
            if categorical:
                Q, HW = k_rec.shape[1], k_rec.shape[2] * k_rec.shape[3]
                check(L.mmvae_ce_fwd(ptr(k_rec), ptr(k_tgt), ptr(k_w), N, Q, HW, base, st), "mmvae_ce_fwd")
            else:
                check(L.mmvae_gauss_nll_fwd(ptr(k_rec), ptr(k_tgt), k_rec.numel(), sigma, base, st), "mmvae_gauss_nll_fwd")
            check(L.mmvae_loss_finish(base, ptr(out), nll_c, kl_c, mmd_c, float(N), st), "mmvae_loss_finish")
            _ = acc                     # (referenced: the accumulator lives as long as the closure)

        if side:
            model.__dict__["_pending_loss"] = (launch, out)
        else:
            launch(_stream())
        model.__dict__["_loss_keep"] = launch
        model._last_scalars = out
        model.__dict__["_last_on_side"] = side
        ctx.model, ctx.N, ctx.categorical = model, N, categorical
        # the reconstruction is the decoder's own output tensor (no crop, no copy): its gradient can be folded into the decoder's backward
        lr = model._last_recon
        ctx.direct_tail = (lr[1] if (lr is not None and lr[0] == recon.data_ptr() and not categorical and model.fuse_loss_tail and
                                     target.dtype == torch.float32 and target.shape == recon.shape) else None)
        ctx.save_for_backward(target, mu, logvar, encoding, recon, true_samples, weight)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        L = lib()
        model, N = ctx.model, ctx.N
        target, mu, logvar, encoding, recon, ts, weight = ctx.saved_tensors
        st = _stream()
        g = g.contiguous().float()
        if ctx.direct_tail is not None and model._stamps.get(("dec", True)) == ctx.direct_tail:
            # No d_recon tensor: hand the decoder's backward what it needs to evaluate coef / sigma^2 * g * (recon - target) itself, and a
            # token in place of the gradient -- a 0-strided NaN, so that any OTHER consumer of it fails loudly instead of silently.
            token = model._tail_token(recon)
            model._pending_tail = (token, target, float(model.sigma_decoder), float(model.nll) / N, g, ctx.direct_tail)
            d_recon = token
        else:
            d_recon = torch.empty_like(recon)
        if ctx.direct_tail is not None and d_recon.stride(0) == 0:
            pass
        elif ctx.categorical:
            Q, HW = recon.shape[1], recon.shape[2] * recon.shape[3]
            check(L.mmvae_ce_bwd(ptr(recon), ptr(target), ptr(weight), N, Q, HW, float(model.nll) / N, ptr(g), ptr(d_recon), st), "mmvae_ce_bwd")
        else:
            check(L.mmvae_gauss_nll_bwd(ptr(recon), ptr(target), recon.numel(), float(model.sigma_decoder), float(model.nll) / N, ptr(g),
                                        ptr(d_recon), st), "mmvae_gauss_nll_bwd")
        d_mu = d_lv = d_enc = None
        if mu is not None and logvar is not None:
            d_mu, d_lv = torch.empty_like(mu), torch.empty_like(logvar)
            check(L.mmvae_kl_bwd(ptr(mu), ptr(logvar), float(model.kl) / N, ptr(g), ptr(d_mu), ptr(d_lv), mu.numel(), st), "mmvae_kl_bwd")
        if encoding is not None and float(model.mmd) != 0.0:
            d_enc = torch.zeros_like(encoding)
            check(L.mmvae_mmd_bwd(ptr(ts), ptr(encoding), N, encoding.shape[1], float(model.mmd) / N, ptr(g), ptr(d_enc), st), "mmvae_mmd_bwd")
        return None, None, d_mu, d_lv, d_enc, d_recon, None, None


# ------------------------------------------------------------------------------------------ the model
class VAE(nn.Module):
    def __init__(self, in_channels, intermediate_channels, decoder_out_channels=1, pixelcnn_out_channels=2,
                 z_dimension=32,
                 pixelcnn=True, only_pixelcnn=True, pixelcnn_layers=4, pixelcnn_activation="ReLu", nll=1, kl=1, mmd=0,
                 require_rsample=True, sigma_decoder=0.1, input_image_size=64, compute_dtype=None, blocks_per_stage=1):
        """Same arguments as the reference (model.py:259-262) plus ``compute_dtype`` ("bf16" default, or "f32";
        also settable through the MMVAE_DTYPE environment variable): storage type of activations / MFMA inputs, and
        ``blocks_per_stage`` (default 1 = the reference network): residual blocks per encoder / decoder stage of the deeper
        build-defined variant (BASELINE configs[3]; include/mmvae.h: mmvae_net_create_ex)."""
        super().__init__()
        if (pixelcnn or only_pixelcnn) and pixelcnn_activation != "ReLu":
            # (the reference accepts "ELU" at the command line but its PixelCNN only knows "ReLu" / "Elu": model.py:243-246 vs main.py:100)
            raise NotImplementedError("PixelCNN is built with the ReLu activation only")
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
        self.adjust = (64 - input_image_size) // 2 if input_image_size > 32 else (32 - input_image_size) // 2   # model.py:307-310
        dt = compute_dtype or os.environ.get("MMVAE_DTYPE", "bf16")
        if dt not in _DTYPES:
            raise ValueError(f"compute_dtype must be one of {sorted(_DTYPES)}")
        self.compute_dtype = {0: "f32", 1: "bf16", 2: "fp8"}[_DTYPES[dt]]

        L = lib()
        d = self.__dict__
        self.blocks_per_stage = int(blocks_per_stage)
        d["_ptable"], d["_btable"] = [], []
        # ---- PixelCNN first: the reference registers it before the encoder / decoder (model.py:283-300), so its parameters lead the
        # parameter list and the state_dict
        d["_hp"], d["_poff"], d["_pix_ws"] = None, 0, None
        pix_tables = None
        if only_pixelcnn or pixelcnn:
            pix_in = in_channels if only_pixelcnn else decoder_out_channels + in_channels                      # model.py:288,312
            hp = ctypes.c_void_p()
            check(L.mmvae_pixelcnn_create(ctypes.byref(hp), pix_in, intermediate_channels, pixelcnn_out_channels, pixelcnn_layers,
                                          1 if _DTYPES[dt] != 0 else 0), "mmvae_pixelcnn_create")
            d["_hp"] = hp
            d["_poff"] = int(L.mmvae_pixelcnn_num_params(hp))
            pix_tables = (pix_in, intermediate_channels, pixelcnn_out_channels, pixelcnn_layers)
        # ---- encoder / decoder
        d["_h"] = None
        n_params, n_bnf, dec_off = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        n_bni, dec_side = ctypes.c_int32(0), ctypes.c_int32(0)
        if not only_pixelcnn:
            h = ctypes.c_void_p()
            check(L.mmvae_net_create_ex(ctypes.byref(h), in_channels, z_dimension, decoder_out_channels, input_image_size,
                                        int(bool(require_rsample)), _DTYPES[dt], self.blocks_per_stage), "mmvae_net_create_ex")
            d["_h"] = h
            check(L.mmvae_net_sizes(h, ctypes.byref(n_params), ctypes.byref(n_bnf), ctypes.byref(n_bni), ctypes.byref(dec_off),
                                    ctypes.byref(dec_side)), "mmvae_net_sizes")
        d["_n_params"], d["_n_bnf"], d["_n_bni"] = self._poff + n_params.value, n_bnf.value, n_bni.value
        d["_dec_off"], d["_dec_side"] = self._poff + dec_off.value, dec_side.value          # absolute: first decoder parameter of the flat buffer
        d["_flat"] = torch.zeros(self._n_params, dtype=torch.float32)
        d["_bnf"] = torch.zeros(self._n_bnf, dtype=torch.float32)
        d["_bni"] = torch.zeros(self._n_bni, dtype=torch.int64)
        d["_G"] = [None, None]
        d["_ws"] = {True: None, False: None}
        d["_stamps"] = {}
        d["_sync"] = None
        d["_last_scalars"] = None
        d["_last_group"] = None
        d["_staged"] = None                # (weakref to the image prepare_batch staged, its version, workspace data_ptr, N)
        d["_grad_key"] = None
        d["_last_recon"] = None            # (data_ptr, forward stamp) of the last train-mode reconstruction
        d["_pending_tail"] = None          # Gaussian loss gradient handed from _LossFn.backward to _DecoderFn.backward
        d["_nan_scalar"] = None
        # Gaussian NLL: fold the loss gradient into the decoder's backward (no d_recon tensor).  Set False when the reconstruction
        # tensor feeds anything besides VAE.loss in the autograd graph.
        d["fuse_loss_tail"] = os.environ.get("MMVAE_FUSE_LOSS_TAIL", "1") != "0"
        d["injected_eps"] = None           # parity tests: noise for rsample / loss instead of torch.randn
        d["injected_true_samples"] = None
        if pix_tables is not None:
            self._build_pixel_tree(*pix_tables)
        else:
            self.pixelcnn = None
        if self._h is not None:
            self._build_tree(L)
        self._reset_parameters()

    # ---- construction helpers
    def _build_tree(self, L):
        name_buf = ctypes.create_string_buffer(128)
        ndim, kind = ctypes.c_int32(), ctypes.c_int32()
        shape = (ctypes.c_int32 * 4)()
        off = ctypes.c_int64()
        for i in range(L.mmvae_net_num_entries(self._h)):
            check(L.mmvae_net_entry(self._h, i, name_buf, 128, ctypes.byref(ndim), shape, ctypes.byref(kind), ctypes.byref(off)), "mmvae_net_entry")
            name = name_buf.value.decode()
            shp = tuple(shape[k] for k in range(ndim.value))
            numel = 1
            for s in shp:
                numel *= s
            parts = name.split(".")
            node = self
            for depth, comp in enumerate(parts[:-1]):
                if comp not in node._modules:
                    role = comp if (depth == 0 and comp in ("encoder", "decoder")) else None
                    node.add_module(comp, _Scope(self, role))
                node = node._modules[comp]
            if kind.value == 0:
                o = self._poff + off.value
                p = nn.Parameter(self._flat[o:o + numel].view(shp))
                p._mmvae_owner = weakref.ref(self)
                node.register_parameter(parts[-1], p)
                self._ptable.append((name, p, o, numel, shp))
            elif kind.value == 1:
                node.register_buffer(parts[-1], self._bnf[off.value:off.value + numel].view(shp))
                self._btable.append((node, parts[-1], 1, off.value, numel, shp))
            else:
                node.register_buffer(parts[-1], self._bni[off.value:off.value + 1].view(()))
                self._btable.append((node, parts[-1], 2, off.value, 1, ()))
        self.__dict__["_enc_params"] = [p for (_, p, o, _, _) in self._ptable if self._poff <= o < self._dec_off]
        self.__dict__["_dec_params"] = [p for (_, p, o, _, _) in self._ptable if o >= self._dec_off]

    def _build_pixel_tree(self, pix_in, mid, pix_out, layers):
        """pixelcnn.layers.<i>.{weight, bias, mask}: the reference's names and order (model.py:234-241; `mask` is a registered buffer, :216)."""
        top = _Scope(self, None)
        self.add_module("pixelcnn", top)
        lay = _Scope(self, None)
        top.add_module("layers", lay)
        off = 0
        masks = []
        for i in range(layers):
            cin = pix_in if i == 0 else mid
            cout = pix_out if i == layers - 1 else mid
            node = _Scope(self, None)
            lay.add_module(str(i), node)
            for leaf, shp in (("weight", (cout, cin, 7, 7)), ("bias", (cout,))):
                n = 1
                for v in shp:
                    n *= v
                p = nn.Parameter(self._flat[off:off + n].view(shp))
                p._mmvae_owner = weakref.ref(self)
                node.register_parameter(leaf, p)
                self._ptable.append((f"pixelcnn.layers.{i}.{leaf}", p, off, n, shp))
                off += n
            m = torch.ones(cout, cin, 7, 7)
            m[:, :, 3, 3 + (0 if i == 0 else 1):] = 0                  # type 'A' first, 'B' afterwards (model.py:216-220, :234-239)
            m[:, :, 4:] = 0
            node.register_buffer("mask", m)
            masks += [m.reshape(-1), torch.ones(cout)]
        assert off == self._poff
        self.__dict__["_pix_mask"] = torch.cat(masks)                     # multiplies the PixelCNN's flat parameter region (biases: ones)
        self.__dict__["_pix_params"] = [e[1] for e in self._ptable]

    @torch.no_grad()
    def _reset_parameters(self):
        """PyTorch default initialisation in the reference's module-construction order, so that
        ``torch.manual_seed(s); VAE(...)`` yields the same parameters as the reference under the same seed:
        convs kaiming_uniform(a=sqrt(5)), conv bias U(+-1/sqrt(fan_in)), BN gamma=1 beta=0, running stats (0,1).
        Order: a block's shortcut conv is created before its main-path convs (model.py:132-141, :196-207)."""
        byname = {n: p for (n, p, _, _, _) in self._ptable}

        def conv(name):
            nn.init.kaiming_uniform_(byname[name], a=math.sqrt(5))

        # PixelCNN first (constructed first, model.py:283-300): nn.Conv2d.reset_parameters per layer -- weight, then bias
        i = 0
        while f"pixelcnn.layers.{i}.weight" in byname:
            w = byname[f"pixelcnn.layers.{i}.weight"]
            conv(f"pixelcnn.layers.{i}.weight")
            bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
            nn.init.uniform_(byname[f"pixelcnn.layers.{i}.bias"], -bound, bound)
            i += 1
        if self._h is None:
            return
        conv("encoder.conv1.weight")
        for i in range(1, 5):
            j = 0
            while f"encoder.layer{i}.{j}.conv1.weight" in byname:
                pre = f"encoder.layer{i}.{j}."
                if pre + "downsample.0.weight" in byname:
                    conv(pre + "downsample.0.weight")
                conv(pre + "conv1.weight"); conv(pre + "conv2.weight")
                j += 1
        conv("encoder.conv_mu.weight")
        if self.require_rsample:
            conv("encoder.conv_logvar.weight")
        conv("decoder.conv1.weight")
        i = 1
        while f"decoder.uplayer{i}.0.conv1.weight" in byname:
            # reference construction order of a stage: the upsample shortcut, then the blocks in order (model.py:196-209)
            j = 0
            while f"decoder.uplayer{i}.{j}.conv1.weight" in byname:
                j += 1
            conv(f"decoder.uplayer{i}.{j - 1}.upsample.0.weight")
            for b in range(j):
                pre = f"decoder.uplayer{i}.{b}."
                conv(pre + "conv1.weight"); conv(pre + "conv2.weight")
            i += 1
        conv("decoder.conv2.weight")
        w = byname["decoder.conv2.weight"]
        bound = 1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])
        nn.init.uniform_(byname["decoder.conv2.bias"], -bound, bound)
        for (name, p, _, _, shp) in self._ptable:
            if len(shp) == 1 and name != "decoder.conv2.bias" and not name.startswith("pixelcnn."):
                p.fill_(1.0 if name.endswith(".weight") else 0.0)
        for (node, leaf, k, _, _, _) in self._btable:
            b = node._buffers[leaf]
            if leaf == "running_var":
                b.fill_(1.0)
            else:
                b.zero_()

    # ---- flat storage management
    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)
        self._reflatten()
        return self

    @torch.no_grad()
    def _reflatten(self):
        dev = self._ptable[0][1].device
        flat = torch.empty(self._n_params, dtype=torch.float32, device=dev)
        for (_, p, off, n, shp) in self._ptable:
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(shp)
            p.grad = None
        bnf = torch.empty(self._n_bnf, dtype=torch.float32, device=dev)
        bni = torch.empty(self._n_bni, dtype=torch.int64, device=dev)
        for (node, leaf, k, off, n, shp) in self._btable:
            src = node._buffers[leaf]
            if k == 1:
                bnf[off:off + n].copy_(src.reshape(-1))
                node._buffers[leaf] = bnf[off:off + n].view(shp)
            else:
                bni[off:off + 1].copy_(src.reshape(-1))
                node._buffers[leaf] = bni[off:off + 1].view(())
        d = self.__dict__
        d["_flat"], d["_bnf"], d["_bni"] = flat, bnf, bni
        d["_G"] = [None, None]
        d["_ws"] = {True: None, False: None}
        d["_pix_ws"] = None
        if self._hp is not None:
            d["_pix_mask"] = self._pix_mask.to(dev)

    def _ensure_flat(self):
        first, last = self._ptable[0], self._ptable[-1]
        base, es = self._flat.data_ptr(), 4
        if first[1].data_ptr() != base + first[2] * es or last[1].data_ptr() != base + last[2] * es:
            self._reflatten()
        if not self._flat.is_cuda:
            raise MmvaeError("the HIP VAE only runs on a GPU: call model.to('cuda') first (there is no CPU fallback; "
                             "the CPU restatement under oracle/ is test infrastructure)")

    def _net_ptr(self, flat_like):
        """Device address of the encoder / decoder's region of a flat parameter-shaped buffer (behind the PixelCNN's, if any)."""
        return flat_like.data_ptr() + 4 * self._poff

    def _pixel_workspace(self, N, S):
        need = lib().mmvae_pixelcnn_workspace_bytes(self._hp, int(N), int(S))
        ws = self._pix_ws
        if ws is None or ws.numel() < need or ws.device != self._flat.device:
            ws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
            self.__dict__["_pix_ws"] = ws
        return ws

    def _workspace(self, N, training):
        need = lib().mmvae_net_workspace_bytes(self._h, int(N))
        ws = self._ws[training]
        if ws is None or ws.numel() < need or ws.device != self._flat.device:
            ws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
            self._ws[training] = ws
        return ws

    def _tail_token(self, like):
        if self._nan_scalar is None or self._nan_scalar.device != like.device:
            self.__dict__["_nan_scalar"] = torch.full((1,), float("nan"), device=like.device, dtype=torch.float32)
        return self._nan_scalar.expand(like.shape)

    def _ws_view(self, dev_ptr, n):
        """f32 view of n floats at a device address inside one of this model's workspaces (SyncBN all-reduce operands)."""
        for ws in (self._ws.values() if isinstance(self._ws, dict) else self._ws):
            if ws is not None and ws.data_ptr() <= dev_ptr < ws.data_ptr() + ws.numel():
                off = dev_ptr - ws.data_ptr()
                return ws[off:off + 4 * n].view(torch.float32)
        raise MmvaeError("sync_bn: operand outside the model's workspace")

    def _stamp(self, which, training):
        tok = self._stamps.get((which, training), 0) + 1
        self._stamps[(which, training)] = tok
        return tok

    def _check_stamp(self, which, tok, training):
        if not training:
            raise MmvaeError("backward through an eval-mode forward is not supported (BatchNorm batch statistics are needed)")
        if self._stamps.get((which, training)) != tok:
            raise MmvaeError("the saved activations of this forward were overwritten by a later forward; "
                             "call backward before running the model again in train mode")

    def _grad_target(self):
        """Flat gradient buffer to write into: buffer 0 normally; buffer 1 when .grad tensors already exist
        (so that autograd's accumulation into them stays correct).  The choice is keyed on the parameter whose .grad autograd
        writes LAST in a backward pass -- the encoder's first parameter (the PixelCNN's when there is no encoder): its .grad does not change
        between the backward nodes of one pass (PixelCNN -> decoder -> encoder), so all of them pick the same buffer."""
        key = self._grad_key
        if key is None:
            key = self._ptable[0][1]
            if self._h is not None:
                for e in self._ptable:
                    if e[2] == self._poff:
                        key = e[1]
                        break
            self.__dict__["_grad_key"] = key
        idx = 0
        if key.grad is not None and self._G[0] is not None and \
                self._G[0].data_ptr() <= key.grad.data_ptr() < self._G[0].data_ptr() + 4 * self._n_params:
            idx = 1
        if self._G[idx] is None or self._G[idx].device != self._flat.device:
            self._G[idx] = torch.zeros(self._n_params, dtype=torch.float32, device=self._flat.device)
        return self._G[idx]

    def _grad_views(self, G, part):
        tab = [e for e in self._ptable if ((self._poff <= e[2] < self._dec_off) if part == 0 else e[2] >= self._dec_off)]
        lo = tab[0][2]
        hi = tab[-1][2] + tab[-1][3]
        chunks = G[lo:hi].split([e[3] for e in tab])
        return tuple(c.view(e[4]) for c, e in zip(chunks, tab))

    # ---- the reference surface
    def _encode(self, x):
        self._ensure_flat()
        x = x.contiguous().float()
        S = self.input_image_size
        if x.dim() != 4 or x.shape[1] != self.in_channels or x.shape[2] != S or x.shape[3] != S:
            raise ValueError(f"expected input (N,{self.in_channels},{S},{S}), got {tuple(x.shape)}")
        out = _EncoderFn.apply(self, x, *self._enc_params)
        return out if self.require_rsample else (out, None)

    def _rsample(self, mu, logvar):
        eps = self.injected_eps
        if eps is None:
            eps = torch.randn(mu.shape, device=mu.device, dtype=mu.dtype)
        return _RsampleFn.apply(mu, logvar, eps.to(mu.device).view(mu.shape))

    def _decode_full(self, encoding):
        self._ensure_flat()
        enc = encoding.contiguous().float().view(-1, self.z_dimensions, 1, 1)
        return _DecoderFn.apply(self, enc, *self._dec_params)

    def _decode(self, encoding):
        out = self._decode_full(encoding)
        if self.adjust != 0:                                   # model.py:328-329
            out = out[:, :, self.adjust:-self.adjust, self.adjust:-self.adjust]
        return out

    def run_pixelcnn(self, concat):                            # model.py:350-351
        if self._hp is None:
            raise MmvaeError("this model has no PixelCNN")
        self._ensure_flat()
        x = concat.contiguous().float()
        if x.dim() != 4 or x.shape[2] != x.shape[3]:
            raise ValueError(f"expected a square (N, C, S, S) input, got {tuple(x.shape)}")
        return _PixelFn.apply(self, x, *self._pix_params_live())

    def _pix_params_live(self):
        return [e[1] for e in self._ptable if e[2] < self._poff]

    def forward(self, x, sample=None):
        """model.py:316-342: returns (mu, logvar, encoding, reconstruction).  With a PixelCNN the reconstruction is its output on
        concat([decoder_output, x]) in train mode / concat([decoder_output, sample]) in eval mode (:331-336); only_pixelcnn: on x itself."""
        if self.only_pixelcnn:
            return None, None, None, self.run_pixelcnn(x)                                   # :339-340
        mu, logvar = self._encode(x)
        encoding = self._rsample(mu, logvar) if self.require_rsample else mu
        decoder_output = self._decode(encoding)
        if self._hp is None:
            return mu, logvar, encoding, decoder_output
        other = x if self.training else sample
        if other is None:
            raise ValueError("a PixelVAE in eval mode needs `sample` (model.py:334-335)")
        return mu, logvar, encoding, self.run_pixelcnn(torch.cat([decoder_output, other.to(decoder_output.dtype)], dim=1))

    def get_z_image(self, encoding):                           # model.py:344-348
        return self._decode(encoding)

    def get_reconstruction(self, encoding, sample=None):       # model.py:353-362
        decoder_output = self._decode(encoding)
        if self._hp is None:
            return decoder_output
        return self.run_pixelcnn(torch.cat([decoder_output, sample.to(decoder_output.dtype)], dim=1))

    def kl_divergence(self, encoding_mu, encoding_logvar):     # model.py:364-365
        acc = torch.zeros(1, dtype=torch.float64, device=encoding_mu.device)
        mu, lv = encoding_mu.contiguous().float(), encoding_logvar.contiguous().float()
        check(lib().mmvae_kl_fwd(ptr(mu), ptr(lv), mu.numel(), ptr(acc), _stream()), "mmvae_kl_fwd")
        return acc[0].float()

    def compute_kernel(self, x, y):                            # model.py:367-376
        """(x_size, y_size) matrix exp(-mean_d((x_i - y_j)^2) / d) -- the helper itself; ``loss`` never materialises it."""
        x, y = x.contiguous().float(), y.contiguous().float()
        if x.dim() != 2 or y.dim() != 2 or x.shape[1] != y.shape[1]:
            raise ValueError("compute_kernel expects (n, d) and (m, d)")
        out = torch.empty((x.shape[0], y.shape[0]), dtype=torch.float32, device=x.device)
        check(lib().mmvae_rbf_kernel(ptr(x), ptr(y), x.shape[0], y.shape[0], x.shape[1], ptr(out), _stream()), "mmvae_rbf_kernel")
        return out

    def compute_mmd(self, x, y):                               # model.py:378-383
        acc = torch.zeros(1, dtype=torch.float64, device=x.device)
        x, y = x.contiguous().float(), y.contiguous().float()
        check(lib().mmvae_mmd_fwd(ptr(x), ptr(y), x.shape[0], x.shape[1], None, ptr(acc), _stream()), "mmvae_mmd_fwd")
        return acc[0].float()

    def _flush_pending_loss(self, only=None):
        """Launch loss-scalar kernels still waiting for a decoder backward pass (_LossFn.forward) on the current stream."""
        pend = self.__dict__.get("_pending_loss")
        if pend is not None and (only is None or pend[1] is only):
            self.__dict__["_pending_loss"] = None
            pend[0](_stream())

    def loss(self, target, encoding_mu, encoding_logvar, encoding, reconstruction, device, args, deferred=False):
        """model.py:385-406: returns (loss tensor with grad, nll/N, kl/N, mmd/N as Python floats) -- ONE device->host
        copy of the four scalars instead of the reference's three ``.item()`` calls.  ``deferred=True`` (used by this
        package's ``train``) returns float-like ``DeferredScalar``s instead, read back when first used, so that the host
        does not wait for the forward pass before it enqueues the backward pass."""
        N = target.shape[0]
        dev = reconstruction.device
        categorical = self.pixelcnn is not None or self.decoder_out_channels > self.in_channels          # model.py:398
        ts = None
        enc2 = None
        if encoding is not None:
            enc2 = encoding.view(-1, encoding.shape[1])
            ts = self.injected_true_samples
            if ts is None:
                ts = torch.randn(N, encoding.shape[1], device=dev)          # model.py:395
            ts = ts.to(dev).contiguous().float()
        weight = None
        if categorical:
            weight = getattr(args, "data_ratio_of_labels", None)
            if weight is not None:
                weight = weight.to(dev).contiguous().float()
            if target.dtype != torch.int64:
                target = target.long()
        # the side-stream form needs the backward pass that follows to join it (and to keep the operands alive): train steps only
        self.__dict__["_loss_side"] = bool(deferred and self.training and torch.is_grad_enabled() and reconstruction.requires_grad and self._h is not None)
        loss_t = _LossFn.apply(self, target, encoding_mu, encoding_logvar, enc2, reconstruction, ts, weight)
        join = None
        if self.__dict__.get("_last_on_side"):
            wr, mine = weakref.ref(self), self._last_scalars    # (weak: the group hangs off the model)

            def join():
                net = wr()
                if net is not None and net._h is not None:     # (a destroyed net has synchronised its side stream)
                    net._flush_pending_loss(mine)
                    check(lib().mmvae_net_join(net._h, _stream()), "mmvae_net_join")
        grp = _StepScalars(self._last_scalars, join)
        self._last_group = grp
        if deferred:
            return loss_t, DeferredScalar(grp, 1), DeferredScalar(grp, 2), DeferredScalar(grp, 3)
        return loss_t, grp.get(1), grp.get(2), grp.get(3)

    # ---- train-loop hook (main.py:374-388): labels -> normalised frames, one kernel
    def prepare_batch(self, batch, device, data_mean, data_std, categorical):
        S = self.input_image_size
        labels = batch.to(device)
        if labels.dtype not in (torch.int64, torch.uint8):
            labels = labels.long()
        labels = labels.contiguous()
        N = labels.numel() // (S * S * self.in_channels)
        image = torch.empty((N, self.in_channels, S, S), device=labels.device, dtype=torch.float32)
        self._ensure_flat()
        if self._h is None:
            # a PixelCNN on its own (main.py:50-58 "pixelcnn_N"): no encoder workspace to stage into, the f32 image only
            if labels.dtype != torch.int64:
                labels = labels.long()
            check(lib().mmvae_normalise_labels(ptr(labels), image.numel(), float(data_mean), float(data_std), ptr(image), _stream()),
                  "mmvae_normalise_labels")
            self.__dict__["_staged"] = None
        else:
            # one pass: the f32 image (network input, Gaussian target) and its storage-type copy straight into the workspace the forward
            # pass is about to use (train() calls the model right after; _EncoderFn checks that it really is this tensor object, unmodified)
            ws = self._workspace(N, bool(self.training))
            check(lib().mmvae_net_stage_labels(self._h, N, ptr(labels), labels.element_size(), float(data_mean), float(data_std), ptr(image), ptr(ws),
                                               ws.numel(), _stream()), "mmvae_net_stage_labels")
            self.__dict__["_staged"] = (weakref.ref(image), image._version, ws.data_ptr(), N)
        if categorical:
            target = labels.view(-1, S, S)
            if target.dtype != torch.int64:
                target = target.long()
        else:
            target = image
        return image, target

    def __repr__(self):
        """The reference's description text (model.py:408-439)."""
        size = str(self.input_image_size) + "x" + str(self.input_image_size) + "x"
        string = ""
        pixelcnn_input = size
        if self.only_pixelcnn:
            pixelcnn_used = "by itself"
            pixelcnn_input += str(self.in_channels)
        else:
            pixelcnn_used = "in the decoder"
            pixelcnn_input += str(self.in_channels + self.decoder_out_channels)
            rsample_text = " Where Z is rsampled from a Normal Distribution." if self.require_rsample else ""
            string += ("We are using an encoder which takes input of " + size + str(self.in_channels) + " and encodes into " +
                       str(self.z_dimensions) + " dimensional latent space." + rsample_text +
                       " \nIt is then pushed into a decoder which outputs an image of dimension " + size +
                       str(self.decoder_out_channels) + ".\n")
        if self.pixelcnn is None:
            if self.decoder_out_channels == self.in_channels:
                string += "We assume p(x/z) follows a normal distribution with mean x_recon and sigma " + str(self.sigma_decoder) + ".\n"
            else:
                string += "We assume p(x/z) follows a categorical distribution. \n"
        else:
            string += ("We are using PixelCNN " + pixelcnn_used + " which takes an input of " + pixelcnn_input + " dimension goes through " +
                       str(self.num_pixelcnn_layers) + " layers and outputs a " + size + str(self.pixelcnn_out_channels) + " dimensions image")
        return string

    def extra_repr(self):
        return f"MI355X/HIP backend, compute_dtype={self.compute_dtype}"

    def __del__(self):
        try:
            h = self.__dict__.get("_h")
            if h:
                lib().mmvae_net_destroy(h)
            hp = self.__dict__.get("_hp")
            if hp:
                lib().mmvae_pixelcnn_destroy(hp)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------ optimiser
class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (defaults of main.py:468) as ONE HIP kernel over the model's flat parameter buffer.
    Drop-in: ``FusedAdam(list(model.parameters()))``; ``state_dict()`` has the torch.optim.Adam layout."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        """capturable=True keeps the step count on the device (mmvae_adam_step_dev), so that a train step captured in a HIP graph
        (torch.cuda.graph) replays with advancing bias corrections; the eager default computes them on the host."""
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        owner = getattr(params[0], "_mmvae_owner", None)
        model = owner() if owner is not None else None
        if model is None or [id(p) for p in params] != [id(e[1]) for e in model._ptable]:
            raise MmvaeError("FusedAdam needs exactly list(model.parameters()) of one HIP VAE, in order")
        self._model = weakref.ref(model)
        self._t = 0
        self._m = self._v = None
        self._capturable = bool(capturable)
        self._step_dev = None

    def _moments(self, model):
        flat = model._flat
        if self._m is None or self._m.device != flat.device:
            old_m, old_v = self._m, self._v
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
            if old_m is not None:
                self._m.copy_(old_m); self._v.copy_(old_v)
            for (_, p, off, n, shp) in model._ptable:
                self.state[p] = {"step": torch.tensor(float(self._t)), "exp_avg": self._m[off:off + n].view(shp),
                                 "exp_avg_sq": self._v[off:off + n].view(shp)}
        if self._capturable and (self._step_dev is None or self._step_dev.device != flat.device):
            # the device-side step counter is created HERE, outside step(): the first step() may run inside torch.cuda.graph, and
            # an allocation + fill captured there would reset the counter on every replay
            self._step_dev = torch.full((1,), float(self._t), dtype=torch.float64, device=flat.device)
        return self._m, self._v

    def prepare_capture(self):
        """Allocate the moments and the device-side step counter now (call once before capturing a step in a HIP graph)."""
        model = self._model()
        model._ensure_flat()
        self._moments(model)
        return self

    def _sync_step_from_device(self):
        """capturable: the authoritative step count lives on the device (graph replays advance it without the host seeing
        a step() call); read it back -- one small D2H copy -- whenever the host needs it (state_dict, resume)."""
        if self._capturable and self._step_dev is not None:
            self._t = int(round(float(self._step_dev.item())))
            for st_ in self.state.values():
                st_["step"] = torch.tensor(float(self._t))

    def state_dict(self):
        self._sync_step_from_device()
        return super().state_dict()

    def _flat_grads(self, model):
        first, last = model._ptable[0][1], model._ptable[-1][1]
        if first.grad is None:
            return None
        for G in model._G:
            if G is not None and first.grad.data_ptr() == G.data_ptr() and last.grad is not None and \
                    last.grad.data_ptr() == G.data_ptr() + model._ptable[-1][2] * 4:
                return G
        G = model._grad_target()                     # slow path: gradients live elsewhere -> gather
        for (_, p, off, n, _) in model._ptable:
            if p.grad is None:
                G[off:off + n].zero_()
            else:
                G[off:off + n].copy_(p.grad.reshape(-1))
        return G

    @torch.no_grad()
    def load_flat_state(self, state_dict):
        """Resume from a torch.optim.Adam-layout state_dict (this class's own or the reference's ``optimizer`` entry)."""
        model = self._model()
        model._ensure_flat()
        m, v = self._moments(model)
        st = state_dict.get("state", {})
        for i, (_, p, off, n, shp) in enumerate(model._ptable):
            e = st.get(i)
            if e is None:
                continue
            m[off:off + n].copy_(e["exp_avg"].reshape(-1).to(m.device))
            v[off:off + n].copy_(e["exp_avg_sq"].reshape(-1).to(v.device))
            self._t = int(float(e["step"]))
        for st_ in self.state.values():
            st_["step"] = torch.tensor(float(self._t))
        if self._step_dev is not None:            # capturable: the device counter follows the restored count (in place: a captured
            self._step_dev.fill_(float(self._t))  # graph keeps pointing at this tensor)
        if state_dict.get("param_groups"):
            g = state_dict["param_groups"][0]
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in g:
                    self.param_groups[0][k] = g[k]

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        model = self._model()
        model._ensure_flat()
        G = self._flat_grads(model)
        if G is None:
            return loss
        scale = 1.0
        if model._sync is not None:
            scale = model._sync.finish(G)
        g = self.param_groups[0]
        m, v = self._moments(model)
        self._t += 1
        b1, b2 = g["betas"]
        if self._capturable:
            check(lib().mmvae_adam_step_dev(ptr(model._flat), ptr(G), ptr(m), ptr(v), model._n_params, float(g["lr"]), float(b1), float(b2),
                                            float(g["eps"]), float(g["weight_decay"]), ptr(self._step_dev), scale, _stream()),
                  "mmvae_adam_step_dev")
            # state[p]["step"] is refreshed from the device counter in state_dict(): under graph replay the host does not see the steps
            return loss
        bc1 = 1.0 - b1 ** self._t
        bc2s = math.sqrt(1.0 - b2 ** self._t)
        check(lib().mmvae_adam_step(ptr(model._flat), ptr(G), ptr(m), ptr(v), model._n_params, float(g["lr"]), float(b1), float(b2),
                                    float(g["eps"]), float(g["weight_decay"]), bc1, bc2s, scale, _stream()), "mmvae_adam_step")
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._t))
        return loss


# ------------------------------------------------------------------------------------------ data parallel
class Communicator:
    """RCCL communicator behind the C ABI (``mmvae_comm_*``, include/mmvae.h): in-place f32 sum all-reduce enqueued on a HIP
    stream by the library itself.  ``Communicator.from_torch_distributed()`` builds one next to an initialised
    torch.distributed job (the 128-byte RCCL id travels through the job's own object broadcast)."""

    def __init__(self, world, rank, unique_id: bytes):
        self._h = ctypes.c_void_p()
        self.world, self.rank = int(world), int(rank)
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(lib().mmvae_comm_init(ctypes.byref(self._h), self.world, self.rank, ctypes.cast(buf, ctypes.c_void_p)), "mmvae_comm_init")

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(128)
        check(lib().mmvae_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), "mmvae_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group=None):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(world, rank, box[0])

    def all_reduce_(self, t: torch.Tensor, stream=None):
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise MmvaeError("Communicator.all_reduce_ expects a contiguous f32 GPU tensor")
        st = _stream() if stream is None else stream
        check(lib().mmvae_comm_allreduce(self._h, ptr(t), t.numel(), st), "mmvae_comm_allreduce")
        return t

    def destroy(self):
        h, self._h = self._h, None
        if h:
            check(lib().mmvae_comm_destroy(h), "mmvae_comm_destroy")

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class GradSync:
    """Data-parallel gradient exchange: one process per GPU, sum-all-reduce over RCCL/xGMI of the flat gradient in two
    buckets -- decoder gradients are complete first and are reduced while the encoder backward still runs -- and 1/world
    scaling folded into Adam.  ``comm="torch"`` (default) issues the buckets through torch.distributed (backend "nccl" = RCCL;
    "gloo" in the CPU tests); ``comm="rccl"`` (or a ``Communicator``) issues them through this library's own RCCL communicator
    (``mmvae_comm_allreduce``) from a communication stream, with no Python in the SyncBN path.
    BatchNorm statistics stay per-rank (like DistributedDataParallel's default) unless sync_bn=True: then every train-mode
    BatchNorm normalises with the statistics of the GLOBAL batch (the reference's semantics at the global batch size):
    the library sums its per-channel partial sums over the ranks -- one small all-reduce per BatchNorm and direction,
    stream-ordered (in-stream RCCL call with a Communicator, host callback into torch.distributed otherwise), equal shards per
    rank assumed."""

    def __init__(self, model, group=None, broadcast=True, sync_bn=False, comm=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.handles = []
        self.reduced = []
        self._comm_stream = None
        self._comm_pending = False
        model._sync = self
        self._cb = None
        comm = comm if comm is not None else os.environ.get("MMVAE_COMM", "torch")
        if isinstance(comm, str):
            if comm not in ("torch", "rccl"):
                raise ValueError("comm must be 'torch', 'rccl' or a Communicator")
            comm = Communicator.from_torch_distributed(group) if comm == "rccl" else None
        self.comm = comm
        self._bn_comms = ()
        if sync_bn and self.comm is not None:
            # the SyncBN rows get communicators of their own, one per stream they are issued from (caller's stream, library side
            # stream): RCCL orders the collectives of ONE communicator, so sharing the bucket communicator would queue the encoder's
            # rows behind the decoder's gradient bucket and lean on RCCL serialising concurrent streams (mmvae.h)
            self._bn_comms = (Communicator.from_torch_distributed(group), Communicator.from_torch_distributed(group))
            check(lib().mmvae_net_set_sync_bn_comm2(model._h, self._bn_comms[0]._h, self._bn_comms[1]._h), "mmvae_net_set_sync_bn_comm2")
        elif sync_bn:
            model._ensure_flat()

            def _allreduce(buf, n, stream, user):
                try:
                    t = model._ws_view(buf, n)
                    st = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream(t.device)
                    with torch.cuda.stream(st):
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    return 0
                except Exception:  # noqa: BLE001 -- must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return -1

            self._cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)(_allreduce)
            check(lib().mmvae_net_set_sync_bn(model._h, ctypes.cast(self._cb, ctypes.c_void_p), None, self.world), "mmvae_net_set_sync_bn")
        if broadcast and self.world > 1:
            model._ensure_flat() if model._flat.is_cuda else None
            dist.broadcast(model._flat, 0, group=group)
            dist.broadcast(model._bnf, 0, group=group)

    def _all_reduce(self, t):
        """One bucket, on the CURRENT stream context."""
        if self.comm is not None:
            self.comm.all_reduce_(t)
        else:
            self.handles.append(self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def bucket_ready(self, G, lo, hi, side_of=None):
        """Start the all-reduce of G[lo:hi].  side_of: the model whose library side stream still carries part of this bucket
        (deferred join): the collective is then issued from a communication stream that waits for the caller's stream and for
        that side stream, and the caller's stream goes on with the encoder backward."""
        if self.world == 1:
            return
        if side_of is None and self.comm is None:
            self._all_reduce(G[lo:hi])
        else:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=G.device)
            cs = self._comm_stream
            cs.wait_stream(torch.cuda.current_stream(G.device))
            with torch.cuda.stream(cs):
                if side_of is not None:
                    check(lib().mmvae_net_join(side_of._h, cs.cuda_stream), "mmvae_net_join")
                self._all_reduce(G[lo:hi])
            self._comm_pending = True
        self.reduced.append((lo, hi))

    def reduce_step_scalars(self, groups):
        """Logged scalars of data-parallel training: ONE all-reduce of the stacked (steps, 4) device tensor, mean over
        ranks.  `groups` are the _StepScalars of the steps not yet read back."""
        if self.world == 1 or not groups:
            return
        stacked = torch.stack([g.ready() for g in groups]).contiguous()
        if self.comm is not None:
            self.comm.all_reduce_(stacked)
        else:
            self.dist.all_reduce(stacked, op=self.dist.ReduceOp.SUM, group=self.group)
        vals = (stacked / self.world).tolist()
        for g, v in zip(groups, vals):
            g.vals = v

    def finish(self, G):
        """Wait for the in-flight buckets; returns the factor Adam applies to the summed gradient."""
        for h in self.handles:
            h.wait()
        if self._comm_pending:                                   # buckets issued from the communication stream
            torch.cuda.current_stream(G.device).wait_stream(self._comm_stream)
            self._comm_pending = False
        covered = sorted(self.reduced)
        self.handles, self.reduced = [], []
        if self.world > 1:
            pos = 0
            for lo, hi in covered:
                if lo > pos:
                    self._all_reduce_now(G[pos:lo])
                pos = max(pos, hi)
            if pos < G.numel():
                self._all_reduce_now(G[pos:])
        return 1.0 / self.world

    def _all_reduce_now(self, t):
        if self.comm is not None:
            self.comm.all_reduce_(t)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
