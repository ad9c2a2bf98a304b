"""GPU: the fused last-up-block kernels (through the C ABI) against torch CPU fp32 autograd of the unfused graph.

The graph (reference model.py:86-88 residual join, :193 tail conv):
    x = relu(y2 * s2 + b2 + ys * ss + bs);  r = conv2d(x, w, bias, padding=1)
  mmvae_tail_join_fwd         r (+ per-image sum / sum of squares) without storing x
  mmvae_tail_join_bwd_reduce  with g = dL/dx masked by x > 0: sum g, sum g*y2, sum g*ys per channel (+ dL/dw partials)
  mmvae_tail_join_bwd_apply   dy2 = A2*g + B2*y2 + C2, dys = As*g + Bs*ys + Cs

Tolerances: f32 3e-5 of the output scale.  bf16: the kernels read bf16 y2 / ys and keep x, g in f32, while weights are
rounded to bf16 for the products that the MFMA path would do in bf16 (g); the reference uses the same rounded inputs and
weights in fp32, so 1e-2 covers the bf16 output rounding of dy2 / dys (2^-9) with margin; reductions are f32: 2e-4.
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

# (N, H, W, out_planes)
SHAPES = [(3, 64, 64, 1), (5, 32, 32, 1), (2, 16, 16, 1), (2, 64, 64, 3), (3, 32, 32, 8), (1100, 8, 8, 1)]


def _lib():
    return importlib.import_module("moving-mnist-vae_amd._lib")


def _round(t, dt):
    return t.to(torch.bfloat16).float() if dt == "bf16" else t


def _nhwc(t, dt):
    x = t.permute(0, 2, 3, 1).contiguous().cuda()
    return x.to(torch.bfloat16) if dt == "bf16" else x


def _case(shape, dt):
    N, H, W, OC = shape
    g = torch.Generator().manual_seed(N * 1000 + H + OC)
    y2 = _round(torch.randn(N, 16, H, W, generator=g), dt)
    ys = _round(torch.randn(N, 16, H, W, generator=g), dt)
    s2, ss = torch.rand(16, generator=g) + 0.5, torch.rand(16, generator=g) + 0.5
    b2, bs = torch.randn(16, generator=g) * 0.3, torch.randn(16, generator=g) * 0.3
    w = torch.randn(OC, 16, 3, 3, generator=g) / 12.0
    bias = torch.randn(OC, generator=g)
    d_raw = torch.randn(N, OC, H, W, generator=g)
    return y2, ys, s2, b2, ss, bs, w, bias, d_raw


def _supported(shape, dt):
    N, H, W, OC = shape
    pt = 64 if dt == "f32" else 128
    return W <= pt and (H * W) % pt == 0


def _join(y2, s2, b2, ys, ss, bs):
    v = lambda t: t.view(1, -1, 1, 1)
    return F.relu((y2 * v(s2) + v(b2)) + (ys * v(ss) + v(bs)))


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [s for s in SHAPES if s[3] == 1], ids=lambda s: "x".join(map(str, s)))
def test_tail_join_fwd(shape, dt):
    L = _lib()
    lib = L.lib()
    N, H, W, OC = shape
    y2, ys, s2, b2, ss, bs, w, bias, _ = _case(shape, dt)
    dti = 0 if dt == "f32" else 1
    st = torch.cuda.current_stream().cuda_stream
    y2d, ysd = _nhwc(y2, dt), _nhwc(ys, dt)
    dev = [t.cuda() for t in (s2, b2, ss, bs, w, bias)]
    r = torch.full((N, 1, H, W), float("nan"), device="cuda")
    stats = torch.zeros(N, 2, device="cuda")
    rc = lib.mmvae_tail_join_fwd(dti, L.ptr(y2d), L.ptr(dev[0]), L.ptr(dev[1]), L.ptr(ysd), L.ptr(dev[2]), L.ptr(dev[3]), L.ptr(dev[4]),
                                 L.ptr(dev[5]), L.ptr(r), L.ptr(stats), N, H, W, st)
    pt = 64 if dt == "f32" else 128
    if W > pt or H % (pt // W):
        assert rc < 0, "unsupported geometry must be refused, not computed"
        return
    assert L.check(rc, "tail_join_fwd") == N
    torch.cuda.synchronize()
    ref = F.conv2d(_join(y2, s2, b2, ys, ss, bs), _round(w, dt), bias, padding=1)
    tol = 3e-5 if dt == "f32" else 2e-4
    err = ((r.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < tol, (shape, dt, err)
    sh = stats.cpu()
    assert torch.allclose(sh[:, 0], ref.sum((1, 2, 3)), rtol=1e-4, atol=1e-3 * H * W ** 0.5)
    assert torch.allclose(sh[:, 1], (ref ** 2).sum((1, 2, 3)), rtol=2e-4)


@pytest.mark.parametrize("N", [1, 3, 40])
def test_tail_join_fwd_stream(N):
    """The MFMA stream form the network runs (bf16, 64x64, one plane): joined activation and weights are rounded to bf16 on their way
    to the MFMA, hence 1e-2 of the largest output; the partial (sum, sumsq) rows add up to the sums of the result it wrote."""
    L = _lib()
    lib = L.lib()
    shape = (N, 64, 64, 1)
    y2, ys, s2, b2, ss, bs, w, bias, _ = _case(shape, "bf16")
    st = torch.cuda.current_stream().cuda_stream
    y2d, ysd = _nhwc(y2, "bf16"), _nhwc(ys, "bf16")
    dev = [t.cuda() for t in (s2, b2, ss, bs, w, bias)]
    r = torch.full((N, 1, 64, 64), float("nan"), device="cuda")
    stats = torch.zeros(max(N, 1), 2, device="cuda")
    rows = L.check(lib.mmvae_tail_join_fwd_stream(1, L.ptr(y2d), L.ptr(dev[0]), L.ptr(dev[1]), L.ptr(ysd), L.ptr(dev[2]), L.ptr(dev[3]), L.ptr(dev[4]),
                                                  L.ptr(dev[5]), L.ptr(r), L.ptr(stats), N, 64, 64, st), "tail_join_fwd_stream")
    torch.cuda.synchronize()
    assert 1 <= rows <= N
    ref = F.conv2d(_join(y2, s2, b2, ys, ss, bs), _round(w, "bf16"), bias, padding=1)
    err = ((r.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-2, err
    sh = stats[:rows].sum(0).cpu()
    got = r.cpu()
    assert abs(sh[0].item() - got.sum().item()) <= 1e-4 * got.abs().sum().item()
    assert abs(sh[1].item() - (got ** 2).sum().item()) <= 1e-4 * (got ** 2).sum().item()
    assert lib.mmvae_tail_join_fwd_stream(1, L.ptr(y2d), L.ptr(dev[0]), L.ptr(dev[1]), L.ptr(ysd), L.ptr(dev[2]), L.ptr(dev[3]), L.ptr(dev[4]), L.ptr(dev[5]),
                                          L.ptr(r), L.ptr(stats), N, 32, 32, st) < 0


@pytest.mark.parametrize("wgrad", [False, True])
@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES + [(2, 128, 128, 1)], ids=lambda s: "x".join(map(str, s)))     # + one-row tiles of the MFMA form
def test_tail_join_bwd(shape, dt, wgrad):
    L = _lib()
    lib = L.lib()
    N, H, W, OC = shape
    if wgrad and OC != 1:
        pytest.skip("fused weight gradient is the one-plane case")
    y2, ys, s2, b2, ss, bs, w, bias, d_raw = _case(shape, dt)
    dti = 0 if dt == "f32" else 1
    st = torch.cuda.current_stream().cuda_stream
    y2d, ysd = _nhwc(y2, dt), _nhwc(ys, dt)
    s2d, b2d, ssd, bsd, wd, drd = [t.cuda() for t in (s2, b2, ss, bs, w, d_raw)]
    partials = torch.zeros(1024, 3, 16, device="cuda")
    wpart = torch.zeros(1024, 16, 9, device="cuda") if wgrad else None
    rows = lib.mmvae_tail_join_bwd_reduce(dti, L.ptr(drd), L.ptr(wd), OC, L.ptr(y2d), L.ptr(s2d), L.ptr(b2d), L.ptr(ysd), L.ptr(ssd),
                                          L.ptr(bsd), L.ptr(partials), L.ptr(wpart), N, H, W, st)
    if not _supported(shape, dt):
        assert rows < 0, "unsupported geometry must be refused, not computed"
        return
    rows = L.check(rows, "tail_join_bwd_reduce")
    assert 1 <= rows <= 1024
    # reference: autograd through x = join(...), r = conv(x, w)
    wq = _round(w, dt)
    x = _join(y2, s2, b2, ys, ss, bs).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wq, None, padding=1).backward(d_raw)
    g = x.grad * (x.detach() > 0)
    F.conv2d(x.detach(), wr, None, padding=1).backward(d_raw)
    sums = partials[:rows].sum(0).cpu()
    ref = torch.stack([g.sum((0, 2, 3)), (g * y2).sum((0, 2, 3)), (g * ys).sum((0, 2, 3))])
    scale = torch.stack([g.abs().sum((0, 2, 3)), (g * y2).abs().sum((0, 2, 3)), (g * ys).abs().sum((0, 2, 3))])
    tol = 3e-5 if dt == "f32" else 2e-4
    err = ((sums - ref).abs() / scale).max().item()
    assert err < tol, (shape, dt, "sums", err)
    if wgrad:
        dw = wpart[:rows].sum(0).cpu().view(1, 16, 3, 3)
        err = ((dw - wr.grad).abs().max() / wr.grad.abs().max()).item()
        assert err < (1e-4 if dt == "f32" else 2e-4), (shape, dt, "wgrad", err)
    # apply
    gen = torch.Generator().manual_seed(7)
    co = [torch.randn(16, generator=gen) for _ in range(6)]
    cod = [c.cuda() for c in co]
    tdt = torch.float32 if dt == "f32" else torch.bfloat16
    dy2 = torch.full((N, H, W, 16), float("nan"), device="cuda", dtype=tdt)
    dys = torch.full((N, H, W, 16), float("nan"), device="cuda", dtype=tdt)
    L.check(lib.mmvae_tail_join_bwd_apply(dti, L.ptr(drd), L.ptr(wd), OC, L.ptr(y2d), L.ptr(s2d), L.ptr(b2d), L.ptr(ysd), L.ptr(ssd), L.ptr(bsd),
                                          L.ptr(cod[0]), L.ptr(cod[1]), L.ptr(cod[2]), L.ptr(cod[3]), L.ptr(cod[4]), L.ptr(cod[5]),
                                          L.ptr(dy2), L.ptr(dys), N, H, W, st), "tail_join_bwd_apply")
    torch.cuda.synchronize()
    v = lambda t: t.view(1, -1, 1, 1)
    r2 = v(co[0]) * g + v(co[1]) * y2 + v(co[2])
    rs = v(co[3]) * g + v(co[4]) * ys + v(co[5])
    tol = 3e-5 if dt == "f32" else 1e-2
    for got, want in ((dy2, r2), (dys, rs)):
        err = ((got.float().cpu().permute(0, 3, 1, 2) - want).abs().max() / want.abs().max()).item()
        assert err < tol, (shape, dt, "apply", err)


@pytest.mark.parametrize("N", [1, 3, 19])
@pytest.mark.parametrize("pro_x", [False, True])
def test_upblock_backward_in_one_pass(N, pro_x):
    """mmvae_upblock_bwd_fused + mmvae_conv1x1_bwd_fused (conv_joinbwd.hip) against the composition they replace, every piece of which
    has its own parity test: mmvae_tail_join_bwd_apply (dy2, dys stored) -> mmvae_convT_bwd_fused over dy2 (conv2) and over dys
    (upsample, with conv1's share x2 (x) w2 and conv1's weight gradient).  The fused pass rounds the dy rows to bf16 in LDS as the
    stored tensors are, and multiplies in the same order: the data gradients agree element for element (up to rare last-bit differences
    of a dy value before its rounding), the weight gradients up to that and the grouping of their f32 partial sums.  bn1's backward sums against torch (from the f32 data gradient the kernel holds)."""
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(500 + N)
    H, W = 64, 64
    bf = lambda t: t.to(torch.bfloat16).float()
    y2, ys = bf(torch.randn(N, 16, H, W, generator=g)), bf(torch.randn(N, 16, H, W, generator=g))
    vec = lambda lo, hi: torch.rand(16, generator=g) * (hi - lo) + lo
    s2, ss, b2, bs = vec(0.5, 1.5), vec(0.5, 1.5), vec(-0.3, 0.3), vec(-0.3, 0.3)
    A2, B2, C2, As, Bs, Cs = vec(0.5, 1.5), vec(-0.05, 0.05), vec(-0.02, 0.02), vec(0.5, 1.5), vec(-0.05, 0.05), vec(-0.02, 0.02)
    tw = torch.randn(1, 16, 3, 3, generator=g) / 12.0
    d_raw = torch.randn(N, 1, H, W, generator=g)
    y1, xin = bf(torch.randn(N, 16, 32, 32, generator=g)), bf(torch.randn(N, 16, 32, 32, generator=g))
    s1, b1 = vec(0.5, 1.5), vec(-0.3, 0.3)
    sx, bx = (vec(0.5, 1.5), vec(-0.3, 0.3)) if pro_x else (None, None)
    w2, wu = torch.randn(16, 16, 4, 4, generator=g) / 8.0, torch.randn(16, 16, 4, 4, generator=g) / 8.0
    w1 = torch.randn(16, 16, 1, 1, generator=g) / 4.0
    A1, B1, C1 = vec(0.5, 1.5), vec(-0.05, 0.05), vec(-0.02, 0.02)
    st = torch.cuda.current_stream().cuda_stream
    cu = lambda t: None if t is None else t.cuda()
    nh = lambda t: _nhwc(t, "bf16")
    y2d, ysd, y1d, xind = nh(y2), nh(ys), nh(y1), nh(xin)
    dv = {k: cu(v) for k, v in dict(s2=s2, ss=ss, b2=b2, bs=bs, A2=A2, B2=B2, C2=C2, As=As, Bs=Bs, Cs=Cs, tw=tw, d_raw=d_raw, s1=s1, b1=b1, sx=sx, bx=bx,
                                      w2=w2, wu=wu, w1=w1, A1=A1, B1=B1, C1=C1).items()}
    P = L.ptr
    big = lambda: torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    # ---- reference composition
    dy2 = torch.empty(N, H, W, 16, device="cuda", dtype=torch.bfloat16)
    dys = torch.empty_like(dy2)
    L.check(lib.mmvae_tail_join_bwd_apply(1, P(dv["d_raw"]), P(dv["tw"]), 1, P(y2d), P(dv["s2"]), P(dv["b2"]), P(ysd), P(dv["ss"]), P(dv["bs"]),
                                          P(dv["A2"]), P(dv["B2"]), P(dv["C2"]), P(dv["As"]), P(dv["Bs"]), P(dv["Cs"]), P(dy2), P(dys), N, H, W, st), "apply")
    dw2_r, dwu_r, dw1_r = torch.zeros(16, 16, 4, 4, device="cuda"), torch.zeros(16, 16, 4, 4, device="cuda"), torch.zeros(16, 16, device="cuda")
    da1_r = torch.empty(N, 32, 32, 16, device="cuda", dtype=torch.bfloat16)
    gin_r = torch.empty_like(da1_r)
    sc, wsc = torch.empty(65536, dtype=torch.uint8, device="cuda"), big()
    L.check(lib.mmvae_convT_bwd_fused(1, P(y1d), P(dy2), P(dv["w2"]), P(dw2_r), P(da1_r), N, 32, 32, 16, 16, 4, 2, 1, P(dv["s1"]), P(dv["b1"]), 1,
                                      None, None, None, P(sc), P(wsc), st), "c2")
    # dy1 from the reference d_a1 (torch, bf16-rounded like the kernel's LDS row)
    da1_f = da1_r.float().cpu().permute(0, 3, 1, 2)
    v4 = lambda t: t.view(1, -1, 1, 1)
    m1 = (y1 * v4(s1) + v4(b1)) > 0
    dy1 = bf(v4(A1) * (da1_f * m1) + v4(B1) * y1 + v4(C1))
    dy1d = nh(dy1)
    w1t = w1.view(16, 16).t().contiguous().cuda()            # the (Cin, 16) form mmvae_convT_bwd_fused takes for w2
    L.check(lib.mmvae_convT_bwd_fused(1, P(xind), P(dys), P(dv["wu"]), P(dwu_r), P(gin_r), N, 32, 32, 16, 16, 4, 2, 1, P(dv["sx"]), P(dv["bx"]), 1,
                                      P(dy1d), P(w1t), P(dw1_r), P(sc), P(wsc), st), "cs")
    # ---- the one-pass form
    dw2_f, dwu_f, dw1_f = torch.zeros_like(dw2_r), torch.zeros_like(dwu_r), torch.zeros(16, 16, 1, 1, device="cuda")
    da1_f2 = torch.full((N, 32, 32, 16), float("nan"), device="cuda", dtype=torch.bfloat16)
    gin_f = torch.full((N, 32, 32, 16), float("nan"), device="cuda", dtype=torch.bfloat16)
    sums = torch.full((2, 16), float("nan"), device="cuda")
    wsc2 = big()
    L.check(lib.mmvae_upblock_bwd_fused(P(dv["d_raw"]), P(dv["tw"]), P(y2d), P(dv["s2"]), P(dv["b2"]), P(ysd), P(dv["ss"]), P(dv["bs"]), P(dv["A2"]),
                                        P(dv["B2"]), P(dv["C2"]), P(dv["As"]), P(dv["Bs"]), P(dv["Cs"]), P(y1d), P(dv["s1"]), P(dv["b1"]), P(dv["w2"]),
                                        P(dw2_f), P(da1_f2), P(sums), P(xind), P(dv["sx"]), P(dv["bx"]), P(dv["wu"]), P(dwu_f), P(gin_f), N, P(wsc2),
                                        st), "upblock_bwd_fused")
    L.check(lib.mmvae_conv1x1_bwd_fused(P(da1_f2), P(y1d), P(dv["s1"]), P(dv["b1"]), P(dv["A1"]), P(dv["B1"]), P(dv["C1"]), P(xind), P(dv["sx"]),
                                        P(dv["bx"]), P(dv["w1"]), P(dw1_f), P(gin_f), N * 32, P(wsc2), st), "conv1x1_bwd_fused")
    torch.cuda.synchronize()
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    # (the two kernels contract a*g + b*y + c into FMAs independently: a last-bit difference of a dy element before its bf16 rounding is
    # possible, so "almost every element equal, the rest one bf16 step apart" instead of torch.equal)
    neq = (da1_f2 != da1_r).float().mean().item()
    assert neq < 2e-3 and rel(da1_f2.float(), da1_r.float()) < 1e-2, (neq, rel(da1_f2.float(), da1_r.float()))
    assert rel(dw2_f, dw2_r) < 1e-4 and rel(dwu_f, dwu_r) < 1e-4, (rel(dw2_f, dw2_r), rel(dwu_f, dwu_r))
    # g_in: the two-pass form adds conv1's share inside the MFMA accumulator, the one-pass form rounds the shortcut's share to bf16 first
    assert rel(gin_f.float(), gin_r.float()) < 1.2e-2
    assert rel(dw1_f.view(16, 16), dw1_r) < 1e-3, rel(dw1_f.view(16, 16), dw1_r)
    gm = da1_r.float().cpu().permute(0, 3, 1, 2) * m1
    ref_sums = torch.stack([gm.sum((0, 2, 3)), (gm * y1).sum((0, 2, 3))])
    assert ((sums.cpu() - ref_sums).abs().max() / ref_sums.abs().max()).item() < 1e-2


@pytest.mark.parametrize("N", [1, 3, 19])
@pytest.mark.parametrize("pro_x", [False, True])
def test_upblock_tail_forward_recomputed(N, pro_x):
    """mmvae_upblock_tail_fwd (up5_tail_fwd_kernel: the join + tail conv with both ConvTranspose2d branch outputs recomputed from their
    inputs) against torch fp32 on the same bf16-rounded operands: r_raw = conv2d(relu(bn2(convT(relu(bn1(y1)))) + bns(convT(x))), w) + bias
    (model.py:70-85,193), and the output statistics.  The kernel keeps the recomputed branch outputs in f32 and rounds the two activations
    it feeds to the MFMA (the ConvTranspose2d inputs, the joined rows) to bf16: tolerance 1e-2 of the output's range, like the stored-tensor
    form's test above."""
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(900 + N)
    bf = lambda t: t.to(torch.bfloat16).float()
    vec = lambda lo, hi: torch.rand(16, generator=g) * (hi - lo) + lo
    y1, xin = bf(torch.randn(N, 16, 32, 32, generator=g)), bf(torch.randn(N, 16, 32, 32, generator=g))
    s1, b1 = vec(0.5, 1.5), vec(-0.3, 0.3)
    sx, bx = (vec(0.5, 1.5), vec(-0.3, 0.3)) if pro_x else (None, None)
    w2, wu = torch.randn(16, 16, 4, 4, generator=g) / 8.0, torch.randn(16, 16, 4, 4, generator=g) / 8.0
    s2, ss, b2, bs = vec(0.5, 1.5), vec(0.5, 1.5), vec(-0.3, 0.3), vec(-0.3, 0.3)
    tw, tb = torch.randn(1, 16, 3, 3, generator=g) / 12.0, torch.randn(1, generator=g)
    v4 = lambda t: t.view(1, -1, 1, 1)
    a1 = bf(torch.relu(y1 * v4(s1) + v4(b1)))
    ax = bf(torch.relu(xin * v4(sx) + v4(bx))) if pro_x else xin
    y2 = F.conv_transpose2d(a1, bf(w2), None, stride=2, padding=1)
    ys = F.conv_transpose2d(ax, bf(wu), None, stride=2, padding=1)
    x = bf(torch.relu(y2 * v4(s2) + v4(b2) + ys * v4(ss) + v4(bs)))
    ref = F.conv2d(x, bf(tw), tb, padding=1)
    cu = lambda t: None if t is None else t.cuda()
    P = L.ptr
    d = {k: cu(v) for k, v in dict(s1=s1, b1=b1, sx=sx, bx=bx, w2=w2, wu=wu, s2=s2, b2=b2, ss=ss, bs=bs, tw=tw, tb=tb).items()}
    y1d, xind = _nhwc(y1, "bf16"), _nhwc(xin, "bf16")
    out = torch.full((N, 1, 64, 64), float("nan"), device="cuda")
    stats = torch.full((max(N, 8), 2), float("nan"), device="cuda")
    sc = torch.empty(65536, dtype=torch.uint8, device="cuda")
    rows = lib.mmvae_upblock_tail_fwd(P(y1d), P(d["s1"]), P(d["b1"]), P(d["w2"]), P(xind), P(d["sx"]), P(d["bx"]), P(d["wu"]), P(d["s2"]), P(d["b2"]),
                                      P(d["ss"]), P(d["bs"]), P(d["tw"]), P(d["tb"]), P(out), P(stats), N, P(sc), torch.cuda.current_stream().cuda_stream)
    L.check(rows, "upblock_tail_fwd")
    torch.cuda.synchronize()
    assert 0 < rows <= stats.shape[0]
    got = out.cpu()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-2, (N, pro_x, err)
    sums = stats[:rows].sum(0).cpu()
    assert abs(sums[0].item() - got.double().sum().item()) <= 1e-3 * got.double().abs().sum().item()
    assert abs(sums[1].item() - (got.double() ** 2).sum().item()) <= 1e-4 * (got.double() ** 2).sum().item()


@pytest.mark.parametrize("C,N,H", [(16, 3, 32), (16, 70, 32), (32, 5, 16), (32, 257, 16), (16, 2, 64)])
@pytest.mark.parametrize("with_stats", [True, False])
def test_join_fused_with_the_next_blocks_conv1(C, N, H, with_stats):
    """mmvae_join_conv1x1_fwd (conv_joinfwd.hip): a DeconvBottleneck's residual join (reference model.py:86-88) fused with the next block's
    conv1 (1x1, C -> 16, no bias; model.py:72) and bn1's batch statistics, against torch CPU fp32 of the unfused graph on the same bf16 inputs
    and bf16-rounded weights.  `out` is a bf16 rounding of an f32 value (2^-8 of its scale); y1 is a bf16 rounding of a C-term f32 sum of
    bf16 products of the ROUNDED out (the kernel multiplies what it stores), so it is compared with the reference conv of the kernel's own
    `out` (2^-8) and, loosely, of the reference `out`; the statistics come from the f32 accumulators: 2e-3 of their scale.  Ragged N: the
    last steps belong to fewer waves than the grid has."""
    L = _lib()
    lib = L.lib()
    torch.manual_seed(C * 1000 + N)
    y2 = torch.randn(N, H, H, C).to(torch.bfloat16)
    ys = torch.randn(N, H, H, C).to(torch.bfloat16)
    s2, ss = torch.rand(C) + 0.5, torch.rand(C) + 0.5
    b2, bs = torch.randn(C) * 0.3, torch.randn(C) * 0.3
    w = torch.randn(16, C) * 0.2
    ref_out = torch.relu(y2.float() * s2 + b2 + ys.float() * ss + bs)
    wr = w.to(torch.bfloat16).float()
    d = torch.device("cuda")
    g = lambda t: t.contiguous().to(d)
    y2d, ysd, s2d, b2d, ssd, bsd, wd = g(y2), g(ys), g(s2), g(b2), g(ss), g(bs), g(w.view(16, C, 1, 1))
    out = torch.empty(N, H, H, C, device=d, dtype=torch.bfloat16)
    y1 = torch.empty(N, H, H, 16, device=d, dtype=torch.bfloat16)
    stats = torch.full((1024 * 32,), float("nan"), device=d)
    scratch = torch.empty(4096, dtype=torch.uint8, device=d)
    st = torch.cuda.current_stream().cuda_stream
    rows = lib.mmvae_join_conv1x1_fwd(L.ptr(y2d), L.ptr(s2d), L.ptr(b2d), L.ptr(ysd), L.ptr(ssd), L.ptr(bsd), L.ptr(wd), C, L.ptr(out), L.ptr(y1),
                                      L.ptr(stats) if with_stats else None, N * H * H, L.ptr(scratch), st)
    L.check(rows, "mmvae_join_conv1x1_fwd")
    torch.cuda.synchronize()
    got_out, got_y1 = out.float().cpu(), y1.float().cpu()
    scale_o = float(ref_out.abs().max())
    assert float((got_out - ref_out).abs().max()) <= 2.0 ** -8 * scale_o + 1e-6
    ref_y1_own = got_out @ wr.t()                        # the conv of what the kernel stored
    scale_y = float(ref_y1_own.abs().max())
    assert float((got_y1 - ref_y1_own).abs().max()) <= 2.0 ** -8 * scale_y + 1e-5
    assert float((got_y1 - ref_out @ wr.t()).abs().max()) <= 2.0 ** -6 * scale_y
    if with_stats:
        assert 0 < rows <= 1024
        part = stats[:rows * 32].view(rows, 2, 16).double().sum(0).cpu()
        assert torch.isfinite(part).all()
        ref = ref_y1_own.double().view(-1, 16)
        s_ref, q_ref = ref.sum(0), (ref * ref).sum(0)
        assert float((part[0] - s_ref).abs().max()) <= 2e-3 * float(ref.abs().sum(0).max())
        assert float((part[1] - q_ref).abs().max()) <= 2e-3 * float(q_ref.max())
