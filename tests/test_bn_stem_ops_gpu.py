"""Op-level parity of the BatchNorm and stem entry points of include/mmvae.h (mmvae_batchnorm_fwd / _bwd, mmvae_stem_fwd / _bwd) against
plain PyTorch fp32 on the CPU: what a maintainer would bind instead of torch.nn.BatchNorm2d (model.py:14,20,30,95) and of
encoder.conv1 + encoder.bn1 + ReLU (model.py:94-95,103)."""
import importlib
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
BN_SCRATCH = 16 << 20


def _L():
    return importlib.import_module("moving-mnist-vae_amd._lib")


def _tol(dt):
    return 2e-5 if dt == "f32" else 2e-2


def _dev(t, dt):
    t = t.cuda()
    return t.to(torch.bfloat16) if dt == "bf16" else t


def _rnd(t, dt):
    return t.to(torch.bfloat16).float() if dt == "bf16" else t


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(3, 16, 16, 32), (5, 4, 4, 128), (2, 64, 64, 16), (7, 2, 2, 256)], ids=str)
@pytest.mark.parametrize("relu", [0, 1])
def test_batchnorm_fwd_bwd(shape, dt, relu):
    L = _L(); lib = L.lib()
    N, H, W, C = shape
    g = torch.Generator().manual_seed(N * 1000 + C + relu)
    y = _rnd(torch.randn(N, C, H, W, generator=g) * 1.7 + 0.3, dt)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dout = _rnd(torch.randn(N, C, H, W, generator=g), dt)
    # reference
    yr = y.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    o = F.batch_norm(yr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    if relu:
        o = F.relu(o)
    o.backward(dout)
    mean = y.mean((0, 2, 3)); var = y.var((0, 2, 3), unbiased=False)
    # device
    st = torch.cuda.current_stream().cuda_stream
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    dti = 0 if dt == "f32" else 1
    npix = N * H * W
    yd = _dev(y.permute(0, 2, 3, 1).contiguous(), dt)
    od = torch.empty((N, H, W, C), device="cuda", dtype=tdt)
    gd, bd, rmd, rvd = gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda()
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    sm, si = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    scratch = torch.empty(BN_SCRATCH, dtype=torch.uint8, device="cuda")
    L.check(lib.mmvae_batchnorm_fwd(dti, L.ptr(yd), npix, C, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), L.ptr(nbt), 0.1, 1e-5, relu, L.ptr(od),
                                    L.ptr(sm), L.ptr(si), L.ptr(scratch), st), "batchnorm_fwd")
    dyd = torch.empty((N, H, W, C), device="cuda", dtype=tdt)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    doutd = _dev(dout.permute(0, 2, 3, 1).contiguous(), dt)
    L.check(lib.mmvae_batchnorm_bwd(dti, L.ptr(doutd), L.ptr(yd), L.ptr(od) if relu else None, npix, C, L.ptr(gd), L.ptr(sm), L.ptr(si), L.ptr(dyd),
                                    L.ptr(dg), L.ptr(db), L.ptr(scratch), st), "batchnorm_bwd")
    torch.cuda.synchronize()
    tol = _tol(dt)
    assert _rel(od.float().cpu().permute(0, 3, 1, 2), o.detach()) < tol
    assert _rel(sm.cpu(), mean) < 1e-5 and _rel(si.cpu(), 1.0 / torch.sqrt(var + 1e-5)) < 1e-4
    assert _rel(rmd.cpu(), rm_r) < 1e-5 and _rel(rvd.cpu(), rv_r) < 1e-5 and int(nbt.item()) == 1
    if dt == "f32" or not relu:          # bf16 + ReLU: the mask of elements that round to 0 differs, covered by the f32 case
        assert _rel(dyd.float().cpu().permute(0, 3, 1, 2), yr.grad) < (tol if dt == "f32" else 4e-2)
        assert _rel(dg.cpu(), gr.grad) < (1e-4 if dt == "f32" else 2e-2) and _rel(db.cpu(), br.grad) < (1e-4 if dt == "f32" else 2e-2)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,S", [(3, 64), (5, 32), (2, 16)])
def test_stem_fwd_bwd(N, S, dt):
    L = _L(); lib = L.lib()
    g = torch.Generator().manual_seed(N * 100 + S)
    x = _rnd((torch.rand(N, 1, S, S, generator=g) < 0.15).float() * 4.3 - 0.23, dt)
    w = torch.randn(32, 1, 5, 5, generator=g) * 0.2
    gamma, beta = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.2
    H1 = S // 2
    gout = _rnd(torch.randn(N, 32, H1, H1, generator=g), dt)
    wq = _rnd(w, dt)
    wr = wq.clone().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    y0 = F.conv2d(x, wr, None, 2, 2)
    y0q = _rnd(y0.detach(), dt)
    a = F.relu(F.batch_norm(y0, None, None, gr, br, True, 0.1, 1e-5))
    a.backward(gout)
    mean = y0.detach().mean((0, 2, 3)); istd = 1.0 / torch.sqrt(y0.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    st = torch.cuda.current_stream().cuda_stream
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    dti = 0 if dt == "f32" else 1
    xd = _dev(x.view(N, S, S), dt)
    yd = torch.empty((N, H1, H1, 32), device="cuda", dtype=tdt)
    stats = torch.zeros(4096 * 64, device="cuda")
    scratch = torch.empty(BN_SCRATCH, dtype=torch.uint8, device="cuda")
    wd0 = w.cuda()
    rows = L.check(lib.mmvae_stem_fwd(dti, L.ptr(xd), L.ptr(wd0), L.ptr(yd), N, S, L.ptr(stats), L.ptr(scratch), st), "stem_fwd")
    torch.cuda.synchronize()
    tol = _tol(dt)
    assert _rel(yd.float().cpu().permute(0, 3, 1, 2), y0.detach()) < tol
    sh = stats[: rows * 64].view(rows, 2, 32).sum(0).cpu()
    assert _rel(sh[0], y0.detach().sum((0, 2, 3))) < 5 * tol and _rel(sh[1], (y0.detach() ** 2).sum((0, 2, 3))) < 5 * tol
    # backward from the reference's own forward values (rounded to the storage type)
    sc = (gamma * istd).cuda(); shf = (beta - mean * gamma * istd).cuda()
    dw, dg, db = torch.zeros(32, 1, 5, 5, device="cuda"), torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
    gd, y0d = _dev(gout.permute(0, 2, 3, 1).contiguous(), dt), _dev(y0q.permute(0, 2, 3, 1).contiguous(), dt)     # keep every operand alive
    wd, gmd, md, isd = w.cuda(), gamma.cuda(), mean.cuda(), istd.cuda()
    L.check(lib.mmvae_stem_bwd(dti, L.ptr(gd), L.ptr(y0d), L.ptr(xd), L.ptr(wd), L.ptr(gmd), L.ptr(sc), L.ptr(shf), L.ptr(md), L.ptr(isd), L.ptr(dw),
                               L.ptr(dg), L.ptr(db), N, S, L.ptr(scratch), st), "stem_bwd")
    torch.cuda.synchronize()
    # f32 is the gate.  bf16: g and y0 are rounded to 8 significant bits before the BatchNorm backward's cancelling sums (tiny batches
    # here), while the sum y0 (x) patch term comes from the exact w x R identity: a few per cent of the largest element is rounding, not error
    lim = 2e-4 if dt == "f32" else 0.15
    errs = (_rel(dw.cpu(), wr.grad), _rel(dg.cpu(), gr.grad), _rel(db.cpu(), br.grad))
    assert max(errs) < lim, errs
