"""GPU: single-op parity of the HIP conv kernels (through the C ABI) against torch CPU fp32 functional ops.

Every distinct (Cin, Cout, k, stride, pad, transposed) of the network (SURVEY.md section 8a layer table), odd
spatial sizes, ragged pixel counts (M not a multiple of the 128-pixel tile), a >1024-tile case for the persistent
loop, the fused BN+ReLU load prologue and the fused per-channel statistics epilogue.

Tolerances: f32 mode (exact-f32 MFMA) 2e-5 of the output scale; bf16 mode compares against the SAME bf16-rounded
inputs evaluated in fp32, so only the bf16 output rounding (2^-9 relative) and accumulation order remain: 1e-2.
Run as a script (python tests/test_ops_gpu.py) for a full error table without stopping at the first failure.
"""
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

# (Cin, Cout, k, s, p, transposed, H, N)
CONFIGS = [
    (32, 32, 3, 2, 1, 0, 16, 3), (32, 32, 3, 1, 1, 0, 8, 3), (32, 32, 1, 2, 0, 0, 16, 3),
    (32, 64, 3, 2, 1, 0, 8, 5), (64, 64, 3, 1, 1, 0, 8, 2), (32, 64, 1, 2, 0, 0, 8, 5),
    (64, 128, 3, 2, 1, 0, 8, 3), (128, 128, 3, 1, 1, 0, 4, 7), (64, 128, 1, 2, 0, 0, 8, 3),
    (128, 256, 3, 2, 1, 0, 4, 9), (256, 256, 3, 1, 1, 0, 2, 11), (128, 256, 1, 2, 0, 0, 4, 9),
    (32, 32, 3, 2, 1, 0, 7, 3), (32, 32, 1, 2, 0, 0, 7, 3), (64, 64, 3, 1, 1, 0, 5, 3),       # odd sizes
    (128, 128, 1, 1, 0, 0, 2, 9), (128, 64, 1, 1, 0, 0, 4, 5), (64, 32, 1, 1, 0, 0, 8, 3),
    (32, 16, 1, 1, 0, 0, 16, 3), (16, 16, 1, 1, 0, 0, 32, 2),
    (128, 128, 4, 2, 1, 1, 2, 9), (128, 64, 4, 2, 1, 1, 4, 5), (64, 64, 4, 2, 1, 1, 4, 5),
    (64, 32, 4, 2, 1, 1, 8, 3), (32, 32, 4, 2, 1, 1, 8, 3), (32, 16, 4, 2, 1, 1, 16, 2),
    (16, 16, 4, 2, 1, 1, 16, 2), (16, 16, 4, 2, 1, 1, 32, 1),
    (32, 128, 2, 2, 0, 1, 1, 37), (128, 128, 2, 2, 0, 1, 1, 300),                               # decoder stem (z -> 128)
    (16, 16, 1, 1, 0, 0, 64, 40), (16, 16, 4, 2, 1, 1, 32, 36),                                  # > 1024 pixel tiles
    (16, 16, 3, 1, 1, 0, 64, 2), (32, 32, 4, 2, 1, 1, 16, 4), (16, 32, 3, 2, 1, 0, 64, 2),       # 256/512-pixel row tiles, LDS-staged epilogue
    (32, 16, 3, 1, 1, 0, 32, 3), (16, 16, 4, 2, 1, 0, 64, 2),
    (32, 16, 4, 2, 1, 1, 32, 5), (16, 16, 4, 2, 1, 1, 32, 70),                                   # streaming weight gradient (32x32 -> 64x64)
    (32, 32, 3, 2, 1, 0, 32, 3), (32, 32, 3, 1, 1, 0, 16, 5), (32, 32, 3, 2, 1, 0, 32, 41),      # streaming 3x3 forward (encoder.layer1)
    # position-major kernel (conv_pos.inc): several 16 / 32 / 64-image tiles with a ragged last one
    (128, 128, 3, 1, 1, 0, 4, 37), (256, 256, 3, 1, 1, 0, 2, 70), (128, 256, 3, 2, 1, 0, 4, 45), (64, 128, 3, 2, 1, 0, 8, 19),
    (128, 128, 4, 2, 1, 1, 2, 50), (128, 64, 4, 2, 1, 1, 4, 21), (64, 64, 4, 2, 1, 1, 4, 33), (128, 256, 1, 2, 0, 0, 4, 40),
    (128, 128, 1, 1, 0, 0, 2, 41), (128, 64, 1, 1, 0, 0, 4, 18), (128, 128, 2, 2, 0, 1, 1, 100),
]


def _lib():
    return importlib.import_module("moving-mnist-vae_amd._lib")


def _to_dev(t_nchw, dt):
    x = t_nchw.permute(0, 2, 3, 1).contiguous().cuda()
    return x.to(torch.bfloat16) if dt == "bf16" else x


def _from_dev(t_nhwc):
    return t_nhwc.float().cpu().permute(0, 3, 1, 2).contiguous()


def _round(t, dt):
    return t.to(torch.bfloat16).float() if dt == "bf16" else t


_WSCRATCH = []


def _wgrad_scratch():
    """Caller-owned partial-image scratch of mmvae_conv2d_wgrad (MMVAE_WGRAD_SCRATCH_BYTES), allocated once per process."""
    if not _WSCRATCH:
        _WSCRATCH.append(torch.empty(64 << 20, dtype=torch.uint8, device="cuda"))
    return _WSCRATCH[0]


def run_config(cfg, dt, prologue=False):
    """Returns dict of relative errors (max-abs error / max-abs reference) for fwd, dgrad, wgrad, stats."""
    L = _lib()
    lib = L.lib()
    Cin, Cout, k, s, p, tr, H, N = cfg
    g = torch.Generator().manual_seed(hash(cfg) & 0xFFFF)
    dti = 0 if dt == "f32" else 1
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    x = _round(torch.randn(N, Cin, H, H, generator=g), dt)
    wshape = (Cin, Cout, k, k) if tr else (Cout, Cin, k, k)
    fan = (Cin * k * k) / (s * s if tr else 1)
    w = torch.randn(wshape, generator=g) / (fan ** 0.5)
    wq = _round(w, dt)
    ps = pb = None
    xin = x
    if prologue:
        ps = torch.rand(Cin, generator=g) + 0.5
        pb = torch.randn(Cin, generator=g) * 0.3
        xin = _round(F.relu(x * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1)), dt)
    xr = xin.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    y = F.conv_transpose2d(xr, wr, None, s, p) if tr else F.conv2d(xr, wr, None, s, p)
    Ho = y.shape[2]
    dy = _round(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    # device
    st = torch.cuda.current_stream().cuda_stream
    xd = _to_dev(x, dt)
    wd = w.cuda()
    yd = torch.empty((N, Ho, Ho, Cout), device="cuda", dtype=tdt)
    scratch = torch.empty(2 * w.numel() * 4 + 256, dtype=torch.uint8, device="cuda")
    stats = torch.zeros(4096 * 2 * Cout, device="cuda")
    psd = ps.cuda() if prologue else None
    pbd = pb.cuda() if prologue else None
    rows = L.check(lib.mmvae_conv2d_fwd(dti, tr, L.ptr(xd), L.ptr(wd), L.ptr(yd), N, H, H, Cin, Cout, k, s, p, L.ptr(psd), L.ptr(pbd), 1,
                                        L.ptr(stats), L.ptr(scratch), st), "conv2d_fwd")
    dyd = _to_dev(dy, dt)
    dxd = torch.full((N, H, H, Cin), float("nan"), device="cuda", dtype=tdt)
    L.check(lib.mmvae_conv2d_dgrad(dti, tr, L.ptr(dyd), L.ptr(wd), L.ptr(dxd), N, H, H, Cin, Cout, k, s, p, L.ptr(scratch), st), "conv2d_dgrad")
    dwd = torch.zeros(wshape, device="cuda")
    L.check(lib.mmvae_conv2d_wgrad(dti, tr, L.ptr(xd), L.ptr(dyd), L.ptr(dwd), N, H, H, Cin, Cout, k, s, p, L.ptr(psd), L.ptr(pbd), 1,
                                   L.ptr(_wgrad_scratch()), st), "conv2d_wgrad")
    torch.cuda.synchronize()
    out = {}
    yh = _from_dev(yd)
    out["fwd"] = ((yh - y.detach()).abs().max() / y.detach().abs().max()).item()
    st_h = stats[: rows * 2 * Cout].view(rows, 2, Cout).sum(0).cpu()
    yref = y.detach()
    out["stat_sum"] = ((st_h[0] - yref.sum((0, 2, 3))).abs().max() / (yref.abs().sum((0, 2, 3)).max() + 1e-9)).item()
    out["stat_sq"] = ((st_h[1] - (yref ** 2).sum((0, 2, 3))).abs().max() / (yref ** 2).sum((0, 2, 3)).max()).item()
    if not prologue:   # dgrad w.r.t. the post-prologue activation is what the network uses; check it on the plain case
        out["dgrad"] = ((_from_dev(dxd) - xr.grad).abs().max() / xr.grad.abs().max()).item()
    out["wgrad"] = ((dwd.cpu() - wr.grad).abs().max() / wr.grad.abs().max()).item()
    return out


def _tol(dt, key):
    if dt == "f32":
        return 3e-5 if key != "wgrad" else 1e-4
    return {"fwd": 1e-2, "dgrad": 1e-2, "wgrad": 1.5e-2, "stat_sum": 1e-2, "stat_sq": 1e-2}[key]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: "x".join(map(str, c)))
def test_conv_ops_match_torch(cfg, dt):
    errs = run_config(cfg, dt)
    for k, v in errs.items():
        assert v == v and v < _tol(dt, k), (cfg, dt, errs)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [CONFIGS[1], CONFIGS[4], CONFIGS[17], CONFIGS[22], CONFIGS[26], CONFIGS[12], CONFIGS[27], CONFIGS[-5], CONFIGS[-4], CONFIGS[-3], CONFIGS[-2]],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_ops_with_fused_bn_relu_prologue(cfg, dt):
    errs = run_config(cfg, dt, prologue=True)
    for k, v in errs.items():
        assert v == v and v < 2 * _tol(dt, k), (cfg, dt, errs)


@pytest.mark.parametrize("N", [1, 5, 37])
@pytest.mark.parametrize("prologue", [False, True])
@pytest.mark.parametrize("second", [False, True])
@pytest.mark.parametrize("C,H", [(16, 32), (16, 16), (32, 16), (32, 32)])
def test_convT_backward_in_one_pass(N, prologue, second, C, H):
    """mmvae_convT_bwd_fused: weight gradient and data gradient of a ConvTranspose2d(16 -> 16, k4 s2 p1) on 32x32 -> 64x64 maps from ONE
    pass over dy (wgrad_stream_kernel with DG), optionally with the 1x1 shortcut's share x2 (x) w2 added to dx -- against PyTorch fp32."""
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(100 + N)
    CO = 16
    x = _round(torch.randn(N, C, H, H, generator=g), "bf16")
    w = torch.randn(C, CO, 4, 4, generator=g) / 8.0
    wq = _round(w, "bf16")
    ps = pb = None
    xin = x
    if prologue:
        ps = torch.rand(C, generator=g) + 0.5
        pb = torch.randn(C, generator=g) * 0.3
        xin = _round(F.relu(x * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1)), "bf16")
    xr = xin.clone().requires_grad_(True)
    wr = wq.clone().requires_grad_(True)
    y = F.conv_transpose2d(xr, wr, None, 2, 1)
    dy = _round(torch.randn(y.shape, generator=g), "bf16")
    y.backward(dy)
    dx_ref = xr.grad.clone()
    x2 = w2 = None
    if second:
        x2 = _round(torch.randn(N, 16, H, H, generator=g), "bf16")
        w2 = torch.randn(C, 16, generator=g) / 4.0
        dx_ref = dx_ref + torch.einsum("nkhw,ak->nahw", x2, _round(w2, "bf16"))
        dw2_ref = torch.einsum("nkhw,nahw->ka", x2, xin)          # the 1x1 conv's own weight gradient, (out = 16, in = C)
    st = torch.cuda.current_stream().cuda_stream
    xd, dyd = _to_dev(x, "bf16"), _to_dev(dy, "bf16")
    wd = w.cuda()
    dwd = torch.zeros(C, CO, 4, 4, device="cuda")
    dxd = torch.full((N, H, H, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    x2d = _to_dev(x2, "bf16") if second else None
    w2d = w2.cuda() if second else None
    psd = ps.cuda() if prologue else None
    pbd = pb.cuda() if prologue else None
    scratch = torch.empty(65536, dtype=torch.uint8, device="cuda")
    dw2d = torch.zeros(16, C, device="cuda") if second else None
    L.check(lib.mmvae_convT_bwd_fused(1, L.ptr(xd), L.ptr(dyd), L.ptr(wd), L.ptr(dwd), L.ptr(dxd), N, H, H, C, CO, 4, 2, 1, L.ptr(psd), L.ptr(pbd), 1,
                                      L.ptr(x2d), L.ptr(w2d), L.ptr(dw2d), L.ptr(scratch), L.ptr(_wgrad_scratch()), st), "convT_bwd_fused")
    torch.cuda.synchronize()
    if second:
        e_dw2 = ((dw2d.cpu() - dw2_ref).abs().max() / dw2_ref.abs().max()).item()
        assert e_dw2 == e_dw2 and e_dw2 < 1.5e-2, e_dw2
    e_dx = ((_from_dev(dxd) - dx_ref).abs().max() / dx_ref.abs().max()).item()
    e_dw = ((dwd.cpu() - wr.grad).abs().max() / wr.grad.abs().max()).item()
    assert e_dx == e_dx and e_dx < 1e-2 and e_dw < 1.5e-2, (e_dx, e_dw)
    # shapes the kernel does not take are refused, not mis-computed
    rc = lib.mmvae_convT_bwd_fused(1, L.ptr(xd), L.ptr(dyd), L.ptr(wd), L.ptr(dwd), L.ptr(dxd), N, 8, 8, C, CO, 4, 2, 1, None, None, 0, None, None, None,
                                   L.ptr(scratch), L.ptr(_wgrad_scratch()), st)
    assert rc < 0


if __name__ == "__main__":
    bad = 0
    for dt in ("f32", "bf16"):
        for pro in (False, True):
            for cfg in CONFIGS:
                try:
                    e = run_config(cfg, dt, pro)
                    flag = "" if all(v == v and v < (2 if pro else 1) * _tol(dt, k) for k, v in e.items()) else "  <<<<<< FAIL"
                    bad += bool(flag)
                    print(dt, "pro" if pro else "   ", cfg, " ".join(f"{k}={v:.2e}" for k, v in e.items()), flag, flush=True)
                except Exception as ex:  # noqa: BLE001
                    bad += 1
                    print(dt, pro, cfg, "EXCEPTION", repr(ex)[:300], flush=True)
    print("FAILURES:", bad)


@pytest.mark.parametrize("N", [2, 37])
@pytest.mark.parametrize("prologue", [False, True])
def test_conv_wgrad_pair_one_pass(N, prologue):
    """mmvae_conv2d_wgrad_pair (wgrad_stream_kernel with the centre-tap companion): the weight gradients of encoder.layer1's conv1 (3x3 s2 p1,
    32 -> 32) and of its 1x1 stride-2 shortcut from ONE pass over the block input, against torch fp32 on the same bf16-rounded operands
    (model.py:29,135-138)."""
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(4000 + N)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(N, 32, 32, 32, generator=g))
    ps = pb = None
    xin = x
    if prologue:
        ps = torch.rand(32, generator=g) + 0.5
        pb = torch.randn(32, generator=g) * 0.3
        xin = bf(F.relu(x * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1)))
    w1 = torch.zeros(32, 32, 3, 3, requires_grad=True)
    ws = torch.zeros(32, 32, 1, 1, requires_grad=True)
    dy1, dys = bf(torch.randn(N, 32, 16, 16, generator=g)), bf(torch.randn(N, 32, 16, 16, generator=g))
    F.conv2d(xin, w1, None, 2, 1).backward(dy1)
    F.conv2d(xin, ws, None, 2, 0).backward(dys)
    xd, d1, d2 = _to_dev(x, "bf16"), _to_dev(dy1, "bf16"), _to_dev(dys, "bf16")
    dw1 = torch.zeros(32, 32, 3, 3, device="cuda")
    dws = torch.zeros(32, 32, 1, 1, device="cuda")
    psd = ps.cuda() if prologue else None
    pbd = pb.cuda() if prologue else None
    L.check(lib.mmvae_conv2d_wgrad_pair(1, L.ptr(xd), L.ptr(d1), L.ptr(d2), L.ptr(dw1), L.ptr(dws), N, 32, 32, 32, 32, L.ptr(psd), L.ptr(pbd), 1,
                                        L.ptr(_wgrad_scratch()), torch.cuda.current_stream().cuda_stream), "conv2d_wgrad_pair")
    torch.cuda.synchronize()
    e1 = ((dw1.cpu() - w1.grad).abs().max() / w1.grad.abs().max()).item()
    es = ((dws.cpu() - ws.grad).abs().max() / ws.grad.abs().max()).item()
    assert e1 < 2e-4 and es < 2e-4, (N, prologue, e1, es)
