"""GPU: mmvae_mmd_fwd (C ABI) against a float64 torch evaluation of the reference's MMD (model.py compute_kernel /
compute_mmd: k(a, b) = exp(-|a - b|^2 / d^2), mmd = mean k(x,x) + mean k(y,y) - 2 mean k(x,y); the entry point returns the
un-normalised sum, divided by n^2 by loss_finish).  Sizes cover one tile, ragged last tiles, even and odd tile counts (the
symmetric sums pair tile rows), the register-pipelined path (d <= 32, d % 4 == 0) and the chunked path (d = 128, d = 30).
Tolerance: rel 2e-5 of the three-term sum (f32 distances through exact-f32 MFMA, double accumulation)."""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _ref(x, y):
    x, y = x.double(), y.double()
    d = x.shape[1]
    k = lambda a, b: torch.exp(-((a[:, None, :] - b[None, :, :]) ** 2).sum(-1) / (d * d))
    return (k(x, x).sum() + k(y, y).sum() - 2 * k(x, y).sum()).item(), (k(x, x).sum() + k(y, y).sum() + 2 * k(x, y).sum()).item()


@pytest.mark.parametrize("n,d", [(8, 32), (64, 32), (100, 32), (200, 32), (300, 32), (333, 16), (450, 32), (700, 8),
                                 (130, 128), (257, 30), (1500, 32)])
def test_mmd_fwd_matches_float64(n, d):
    L = importlib.import_module("moving-mnist-vae_amd._lib")
    lib = L.lib()
    g = torch.Generator().manual_seed(n * 7 + d)
    x = torch.randn(n, d, generator=g) * 1.5
    y = torch.randn(n, d, generator=g) + 0.3
    xd, yd = x.cuda(), y.cuda()
    scratch = torch.zeros(2 * n, device="cuda")
    acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.mmvae_mmd_fwd(L.ptr(xd), L.ptr(yd), n, d, L.ptr(scratch), L.ptr(acc), st), "mmd_fwd")
    torch.cuda.synchronize()
    want, scale = _ref(x, y)
    assert abs(acc.item() - want) <= 2e-5 * scale, (n, d, acc.item(), want)
