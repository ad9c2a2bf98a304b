"""CPU: the C-ABI library loads and exports every symbol include/mmvae.h declares; host-side logic of the
product package (inventory, flat storage, state_dict parity with the oracle spec) needs no GPU."""
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mmvae.h")).read()
    return sorted(set(re.findall(r"MMVAE_API[^;(]*?\b(mmvae_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    lib = L.lib()
    syms = _header_symbols()
    assert len(syms) >= 29
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(L.PROTOTYPES) == syms, "ctypes prototypes and header drifted apart"
    assert lib.mmvae_abi_version() == 3


def test_inventory_matches_oracle_spec(pkg, oracle):
    M = importlib.import_module("moving-mnist-vae_amd.model")
    for (z, oc, S, rs) in [(32, 1, 64, True), (128, 2, 64, True), (32, 1, 28, False), (512, 1, 32, True)]:
        m = M.VAE(1, 32, oc, 2, z, False, False, 4, "ReLu", 1, 1, 0, rs, 0.1, S)
        spec = oracle.state_spec(1, z, oc, S, rs)
        sd = m.state_dict()
        assert list(sd.keys()) == [k for k, _, _ in spec]
        for k, shape, _ in spec:
            assert tuple(sd[k].shape) == tuple(shape), k
        assert [k for k, _ in m.named_parameters()] == [k for k, _, kind in spec if kind in ("conv", "convT", "bias", "bn_w", "bn_b")]
        # parameters are views of ONE flat buffer, in registration order
        off = 0
        for p in m.parameters():
            assert p.data_ptr() == m._flat.data_ptr() + 4 * off
            off += p.numel()
        assert off == m._flat.numel()
        state = oracle.filled_state(spec, seed=3)
        m.load_state_dict(state)
        for k, v in m.state_dict().items():
            assert torch.equal(v, state[k]), k
        assert m.adjust == oracle.adjust_for(S)


def test_unsupported_configurations_fail_loudly(pkg):
    M = importlib.import_module("moving-mnist-vae_amd.model")
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    with pytest.raises(NotImplementedError):
        M.VAE(1, 32, pixelcnn_activation="Elu")       # PixelCNN: ReLu only (the reference's ELU branch is unreachable from its CLI, main.py:100)
    with pytest.raises(L.MmvaeError):
        M.VAE(1, 24)                                  # PixelCNN intermediate_channels not a multiple of 16
    with pytest.raises(L.MmvaeError):
        M.VAE(5, 32, 4, 2, 32, False, False)          # in_channels > 4
    with pytest.raises(L.MmvaeError):
        M.VAE(1, 32, 1, 2, 20, False, False)          # z not a multiple of 8
    m = M.VAE(1, 32, 1, 2, 32, False, False)
    with pytest.raises(L.MmvaeError):                  # no CPU fallback
        m(torch.zeros(2, 1, 64, 64))


def test_product_never_imports_the_oracle():
    pkgdir = os.path.join(ROOT, "moving-mnist-vae_amd")
    for base, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(base, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle." not in txt.replace("oracle/", ""), (base, f)
