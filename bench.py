#!/usr/bin/env python3
"""Benchmark of the hot path: conv-VAE train step (encoder -> reparameterise -> decoder -> ELBO -> backward -> Adam)
on synthetic 20x64x64 Moving-MNIST clips (BASELINE.json configs[1]: 256 clips = 5120 frames per GPU, z=128, bf16).

  python bench.py --gpus N --steps K --warmup W
      N>1: one rank per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process is one of the N ranks;
      started plainly, it launches `python -m torch.distributed.run --nproc-per-node N ... bench.py` as a child BEFORE
      touching the GPU itself and exits with the child's code (fails loudly when fewer than N GPUs are visible).

A "step" is one pass of this repo's ``train`` loop body (main.py:371-399 restated) over one batch that is already
resident in HBM: labels -> normalise -> forward -> loss -> zero_grad -> backward -> FusedAdam.step.  The four scalars of
every step (loss, nll, kl, mmd) stay on the device and are read back once, inside the timed region, when ``train``
returns its per-step lists -- the host never waits for the GPU in the middle of a step.  Prints ONE JSON line (rank 0).
Besides the contract's keys the line carries ``roofline`` (the largest kernel family of the step by GPU time -- wgrad2_kernel -- at its
slowest layer, isolated launches timed with events on the launch stream), ``roofline_largest_launch`` (the single longest launch of the
step, join_bwd_stream_kernel, same method), ``top_kernels`` (the committed per-family table: ms per step, algorithmic and measured bytes),
``step_hbm_roofline`` (whole step against the ideal-fusion byte count), ``config4_deeper`` / ``config5_fp8`` (short runs of BASELINE
configs[3] / [4]) and ``cpu_baseline`` (the CPU oracle on a bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "moving-mnist-vae_amd"
METRIC = "frames/sec/GPU VAE train step, 20×64×64 batch; ELBO vs CPU ref"
DATA_MEAN, DATA_STD, P_ON = 0.0521, 0.2222, 0.0521
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
TRAIN_BYTES_PER_FRAME = 3.40e6  # ideal-fusion bf16 activation traffic of one train step, per frame
TRAIN_FLOP_PER_FRAME = 227409920  # z=128, SURVEY.md section 8(d)
C4_TRAIN_FLOP_PER_FRAME = {512: 409862144}   # tools/count_flops.py: z=512, 2 blocks per stage (forward 137 166 848)


def synthetic_clips(clips, seed, device):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    lab = (torch.rand((clips, 20, 64, 64), generator=g) < P_ON).long()
    return lab.to(device)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _time_oracle(batch, z, warm, steps):
    """`warm` + `steps` train steps of the CPU oracle (bit-identical restatement of the reference model.py, torch CPU ops,
    torch.optim.Adam) on one resident batch; returns (mean, min) seconds per step."""
    from oracle import vae_oracle as O
    pkg = importlib.import_module(PKG)
    torch.manual_seed(0)
    m = O.OracleVAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64)
    m.tiled_mmd = True
    opt = torch.optim.Adam(list(m.parameters()))
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    dev = torch.device("cpu")
    if warm > 0:
        pkg.train(m, [batch] * warm, opt, dev, args, data_mean=DATA_MEAN, data_std=DATA_STD)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        pkg.train(m, [batch], opt, dev, args, data_mean=DATA_MEAN, data_std=DATA_STD)
        ts.append(time.perf_counter() - t0)
    return sum(ts) / len(ts), min(ts)


def cpu_baseline(z):
    """the oracle timed on this host's cores, fp32, all threads, >= 3 warm-up + >= 20 timed steps of
    config 1 (32 frames, z=32), plus a bounded sample of the benchmarked shape (config 2: 20-frame clips, z as benchmarked)."""
    # all the cores this process may use (cpu affinity / container share), never more threads than torch's own default
    # (physical cores): oversubscribed oneDNN threads spin and the baseline collapses
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, torch.get_num_threads())))
    print(f"[bench] cpu baseline: {torch.get_num_threads()} threads on {_cpu_model()}", file=sys.stderr, flush=True)
    g = torch.Generator(device="cpu"); g.manual_seed(1234)
    c1_batch = (torch.rand((32, 64, 64), generator=g) < P_ON).long()                 # config 1: 32 single frames
    # config 1 is tiny (32 frames): all cores oversubscribe it, so it is also timed on 16 threads and the faster one is reported
    all_threads = torch.get_num_threads()
    c1_mean, c1_min = _time_oracle(c1_batch, 32, 3, 20)
    c1_threads = all_threads
    if all_threads > 16:
        torch.set_num_threads(16)
        m16, n16 = _time_oracle(c1_batch, 32, 3, 20)
        torch.set_num_threads(all_threads)
        if m16 < c1_mean:
            c1_mean, c1_min, c1_threads = m16, n16, 16
    print(f"[bench] cpu baseline config 1: {1e3 * c1_mean:.1f} ms/step on {c1_threads} threads", file=sys.stderr, flush=True)
    clips, steps = 32, 3
    c2_mean, c2_min = _time_oracle(synthetic_clips(clips, 1234, "cpu"), z, 1, steps)
    frames = clips * 20
    return {"value": frames / c2_mean, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": _cpu_model(),
            "sample": f"{steps} timed train steps (1 warm-up) of {clips} clips x 20 frames = {frames} frames (z={z}, fp32, torch CPU ops, "
                      f"tiled MMD), oracle/vae_oracle.py: the benchmarked shape at 1/8 of its batch",
            "ms_per_step": 1e3 * c2_mean, "ms_per_step_min": 1e3 * c2_min,
            "config1": {"value": 32 / c1_mean, "unit": "frames/s", "cores": c1_threads, "ms_per_step": 1e3 * c1_mean, "ms_per_step_min": 1e3 * c1_min,
                        "sample": "BASELINE configs[0]: 32 frames x 64x64, z=32, fp32, 3 warm-up + 20 timed steps"}}


def elbo_check(M, device, dtype):
    """Small live parity sample: ELBO of the HIP model vs the CPU oracle on the same weights / inputs / noise."""
    from oracle import vae_oracle as O
    N, z = 40, 128
    spec = O.state_spec(1, z, 1, 64, True)
    torch.manual_seed(0)
    m = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype=dtype)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(device).train()
    labels = O.synthetic_labels(N, 64, seed=3)
    image = O.normalise(labels, 64)
    eps, ts = torch.randn(N, z, 1, 1), torch.randn(N, z)
    mu, lv, enc, rec = O.vae_forward({k: v.clone() for k, v in state.items()}, image, eps, 64, True, True)
    ref = O.vae_loss(image, mu, lv, enc, rec, ts, nll=1, kl=1, mmd=0, sigma_decoder=0.1)[0].item()
    m.injected_eps, m.injected_true_samples = eps.to(device), ts.to(device)
    with torch.no_grad():
        hmu, hlv, henc, hrec = m(image.to(device))
        got = m.loss(image.to(device), hmu, hlv, henc, hrec, device, types.SimpleNamespace())[0].item()
    return abs(got - ref) / abs(ref)


def pmc_traffic(kernel_key, N):
    """HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes, tools/profile_round.sh) from the
    committed table profiles/*_pmc_hbm.csv -- only rows measured on THIS build (source hash) and batch count; else None."""
    import csv
    import glob
    L = importlib.import_module(PKG + "._lib")
    build = L.build_hash()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.csv")), reverse=True):
        for r in csv.DictReader(open(path)):
            if r["build"] == build and int(r["frames"]) == N and r["key"] == kernel_key:
                return float(r["fetch_bytes"]) + float(r["write_bytes"]), f"{os.path.relpath(path, ROOT)} (build {build})"
    return None, None


def committed_top_kernels(N):
    """The step's kernel families (launches, ms per step, stream, algorithmic MB as declared by the launchers, measured HBM MB, in-step
    TB/s, traffic ratio) and its twelve longest launches, from the newest profiles/*_top_kernels.json (tools/top_kernels.py: rocprofv3
    kernel trace + launch census + per-dispatch PMC table of one run).  `same_build` says whether it was taken on this source hash."""
    import glob
    L = importlib.import_module(PKG + "._lib")
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_top_kernels.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if int(d.get("frames", -1)) != N:
            continue
        d["source"] = os.path.relpath(path, ROOT)
        d["same_build"] = d.get("build") == L.build_hash()
        return d
    return None


def _time_calls(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# The layers whose weight gradient runs on wgrad2_kernel in the step (the others: uplayer5 inside join_bwd_stream_kernel, uplayer4 and
# encoder.layer1's 3x3 convs on wgrad_stream_kernel, the layers whose small-side map is at most 4x4 on wgrad_pos_kernel since round 4, the
# stem / tail / heads on their own kernels):
# name, transposed, Cin, Cout, k, stride, pad, H (input side of the forward op), BatchNorm+ReLU prologue on x
WGRAD2_LAYERS = [
    ("encoder.layer2.0.conv1", 0, 32, 64, 3, 2, 1, 16, 0), ("encoder.layer2.0.conv2", 0, 64, 64, 3, 1, 1, 8, 1),
    ("encoder.layer2.0.downsample.0", 0, 32, 64, 1, 2, 0, 16, 0),
    ("encoder.layer3.0.downsample.0", 0, 64, 128, 1, 2, 0, 8, 0),
    ("encoder.layer4.0.downsample.0", 0, 128, 256, 1, 2, 0, 4, 0),
    ("decoder.uplayer1.0.conv1", 0, 128, 128, 1, 1, 0, 2, 1),
    ("decoder.uplayer2.0.conv1", 0, 128, 64, 1, 1, 0, 4, 0),
    ("decoder.uplayer3.0.conv1", 0, 64, 32, 1, 1, 0, 8, 0), ("decoder.uplayer3.0.conv2", 1, 32, 32, 4, 2, 1, 8, 1),
    ("decoder.uplayer3.0.upsample.0", 1, 64, 32, 4, 2, 1, 8, 0),
]
# ... and the eight that moved to wgrad_pos_kernel in round 4 (small-side map up to 4x4, large-side up to 8x8; reported beside the family)
WGRAD_POS_LAYERS = [
    ("encoder.layer3.0.conv1", 0, 64, 128, 3, 2, 1, 8, 0), ("encoder.layer3.0.conv2", 0, 128, 128, 3, 1, 1, 4, 1),
    ("encoder.layer4.0.conv1", 0, 128, 256, 3, 2, 1, 4, 0), ("encoder.layer4.0.conv2", 0, 256, 256, 3, 1, 1, 2, 1),
    ("decoder.uplayer1.0.conv2", 1, 128, 128, 4, 2, 1, 2, 1), ("decoder.uplayer1.0.upsample.0", 1, 128, 128, 4, 2, 1, 2, 1),
    ("decoder.uplayer2.0.conv2", 1, 64, 64, 4, 2, 1, 4, 1), ("decoder.uplayer2.0.upsample.0", 1, 128, 64, 4, 2, 1, 4, 0),
]


def wgrad2_family_roofline(M, device, N, reps=10):
    """The wgrad2_kernel family: the weight gradient of the channel-heavy layers on 8x8 / 16x16 maps and of the 1x1 convs (side stream).  Every
    layer that runs on it is timed in isolation through the C ABI (mmvae_conv2d_wgrad: kernel + partial-image reduce, events on the launch
    stream), the SLOWEST one is the family's largest instance and is reported against the roofline that bounds it: algorithmic bytes = x + dy
    (bf16) + the weight gradient (f32), flops = 2 * pixels * Cin * Cout * k^2; bound = mfma when flops / bytes exceeds the bf16 ridge
    (2.5 PF / 8 TB/s)."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    wsc = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    rows = []
    for name, tr, Cin, Cout, k, sd, p, H, pro in WGRAD2_LAYERS:
        Ho = (H - 1) * sd - 2 * p + k if tr else (H + 2 * p - k) // sd + 1
        x = torch.randn(N, H, H, Cin, device=device).to(torch.bfloat16)
        dy = torch.randn(N, Ho, Ho, Cout, device=device).to(torch.bfloat16)
        dw = torch.zeros((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=device)
        sc, sh = torch.rand(Cin, device=device) + 0.5, torch.randn(Cin, device=device) * 0.1

        def fn():
            L.check(lib.mmvae_conv2d_wgrad(1, tr, L.ptr(x), L.ptr(dy), L.ptr(dw), N, H, H, Cin, Cout, k, sd, p, L.ptr(sc) if pro else None,
                                           L.ptr(sh) if pro else None, 1, L.ptr(wsc), st), name)
        ms = _time_calls(fn, 4, warm=1)
        pix = N * (H * H if tr else Ho * Ho)
        rows.append(dict(layer=name, ms=ms, bytes=(x.numel() + dy.numel()) * 2 + dw.numel() * 4, flop=2.0 * pix * Cin * Cout * k * k))
        del x, dy, dw
    # the layers that left the family for wgrad_pos_kernel in round 4, timed the same way (kernel + ordered reduce)
    pos_rows = {}
    for name, tr, Cin, Cout, k, sd, p, H, pro in WGRAD_POS_LAYERS:
        Ho = (H - 1) * sd - 2 * p + k if tr else (H + 2 * p - k) // sd + 1
        x = torch.randn(N, H, H, Cin, device=device).to(torch.bfloat16)
        dy = torch.randn(N, Ho, Ho, Cout, device=device).to(torch.bfloat16)
        dw = torch.zeros((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=device)
        sc, sh = torch.rand(Cin, device=device) + 0.5, torch.randn(Cin, device=device) * 0.1

        def fnp():
            L.check(lib.mmvae_conv2d_wgrad(1, tr, L.ptr(x), L.ptr(dy), L.ptr(dw), N, H, H, Cin, Cout, k, sd, p, L.ptr(sc) if pro else None,
                                           L.ptr(sh) if pro else None, 1, L.ptr(wsc), st), name)
        ms = _time_calls(fnp, 4, warm=1)
        flop = 2.0 * N * (H * H if tr else Ho * Ho) * Cin * Cout * k * k
        pos_rows[name] = {"ms": round(ms, 4), "TFLOPs": round(flop / (ms * 1e-3) / 1e12, 1), "frac_of_mfma_peak": round(flop / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 3)}
        del x, dy, dw
    worst = max(rows, key=lambda r: r["ms"])
    name = worst["layer"]
    spec = next(l for l in WGRAD2_LAYERS if l[0] == name)
    _, tr, Cin, Cout, k, sd, p, H, pro = spec
    Ho = (H - 1) * sd - 2 * p + k if tr else (H + 2 * p - k) // sd + 1
    x = torch.randn(N, H, H, Cin, device=device).to(torch.bfloat16)
    dy = torch.randn(N, Ho, Ho, Cout, device=device).to(torch.bfloat16)
    dw = torch.zeros((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=device)
    sc, sh = torch.rand(Cin, device=device) + 0.5, torch.randn(Cin, device=device) * 0.1
    ms = _time_calls(lambda: L.check(lib.mmvae_conv2d_wgrad(1, tr, L.ptr(x), L.ptr(dy), L.ptr(dw), N, H, H, Cin, Cout, k, sd, p, L.ptr(sc) if pro else None,
                                                            L.ptr(sh) if pro else None, 1, L.ptr(wsc), st), name), reps)
    alg, flop = worst["bytes"], worst["flop"]
    gbs, tfs = alg / (ms * 1e-3) / 1e9, flop / (ms * 1e-3) / 1e12
    mfma_bound = flop / alg > MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    top = committed_top_kernels(N)
    fam = None if top is None else next((f for f in top["families"] if f["family"] == "wgrad2_kernel"), None)
    traffic, tsrc = pmc_traffic(f"wgrad2.{name}", N)       # kernel + its partial-image reduce inside the step (tools/pmc_hbm_csv.py)
    return {"bound": "mfma" if mfma_bound else "hbm", "achieved": tfs if mfma_bound else gbs, "peak": MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": (tfs / MFMA_PEAK_TFLOPS) if mfma_bound else (gbs / HBM_PEAK_GBS),
            "traffic": traffic, "traffic_source": tsrc,
            "kernel": f"wgrad2_kernel (+ wgrad_reduce_kernel) @ {name}: weight gradient of {'ConvTranspose2d' if tr else 'Conv2d'}({Cin} -> {Cout}, k{k} s{sd} p{p}) "
                      f"on {H}x{H} inputs, the slowest of the {len(WGRAD2_LAYERS)} layers of the step's largest kernel family by GPU time "
                      "(isolated mmvae_conv2d_wgrad calls: tap-split MFMA tiles, per-block partial images, ordered reduce)",
            "algorithmic_bytes_per_launch": alg, "algorithmic_flop_per_launch": flop, "avg_launch_ms": ms,
            "achieved_GBs": gbs, "achieved_TFLOPs": tfs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "frac_of_mfma_peak": tfs / MFMA_PEAK_TFLOPS,
            "family_in_step": fam, "family_layers_isolated_ms": {r["layer"]: round(r["ms"], 4) for r in rows},
            "family_isolated_ms_sum": sum(r["ms"] for r in rows), "wgrad_pos_layers_isolated": pos_rows}


def convt4_family_roofline(M, device, N, reps=20):
    """The convT4_stream_kernel family (ConvTranspose2d 16 -> 16, k4 s2 p1 forward as per-wave streams: decoder.uplayer5.conv2 / .upsample at
    32x32 -> 64x64, uplayer4.conv2 at 16x16 -> 32x32): its largest instance, uplayer5.conv2, timed in isolation through the C ABI (events on the
    launch stream, weights packed once outside the timed launches).  Algorithmic bytes = input + output activations (bf16) + weights, once."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    rows = {}
    keep = None
    for name, H in (("decoder.uplayer5.0.conv2", 32), ("decoder.uplayer5.0.upsample.0", 32), ("decoder.uplayer4.0.conv2", 16)):
        x = torch.randn(N, H, H, 16, device=device).to(torch.bfloat16)
        w = torch.randn(16, 16, 4, 4, device=device) * 0.1
        y = torch.empty(N, 2 * H, 2 * H, 16, device=device, dtype=torch.bfloat16)
        scratch = torch.empty(2 * w.numel() * 2 + 256, dtype=torch.uint8, device=device)
        stats = torch.zeros(4096 * 2 * 16, device=device)
        sc, sh = torch.rand(16, device=device) + 0.5, torch.randn(16, device=device) * 0.1
        pro = not name.endswith("upsample.0")

        def launch(weights=None):
            L.check(lib.mmvae_conv2d_fwd(1, 1, L.ptr(x), L.ptr(weights), L.ptr(y), N, H, H, 16, 16, 4, 2, 1, L.ptr(sc) if pro else None,
                                         L.ptr(sh) if pro else None, 1 if pro else 0, L.ptr(stats), L.ptr(scratch), st), name)
        launch(w)                                            # packs the weights into `scratch`; the timed launches reuse the pack
        ms = _time_calls(launch, reps if H == 32 else 6, warm=2)
        alg = x.numel() * 2 + y.numel() * 2 + w.numel() * 2
        rows[name] = {"ms": round(ms, 4), "algorithmic_bytes": alg, "GBs": round(alg / (ms * 1e-3) / 1e9, 1)}
        if keep is None:
            keep = (ms, alg)
        del x, y
    ms, alg = keep
    ach = alg / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic("uplayer5.conv2.fwd", N)
    top = committed_top_kernels(N)
    fam = None if top is None else next((f for f in top["families"] if f["family"] == "convT4_stream_kernel"), None)
    # the same launch inside the step (committed rocprofv3 trace): it runs beside its twin, the shortcut branch's convT4_stream_kernel<32, false>
    # on the side stream (same bytes), so its own rate there is about half of the pair's
    in_step = None
    if top is not None:
        me = next((l for l in top.get("largest", []) if "convT4_stream_kernel<32, true" in l["kernel"]), None)
        twin = next((l for l in top.get("largest", []) if "convT4_stream_kernel<32, false" in l["kernel"]), None)
        if me is not None:
            in_step = {"ms": me["ms"], "GBs": round(alg / (me["ms"] * 1e-3) / 1e9, 1), "frac": round(alg / (me["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                       "measured_hbm_MB": me.get("hbm_MB"), "source": top["source"]}
            if twin is not None:
                span = max(me["ms"], twin["ms"])
                in_step["with_twin_on_side_stream"] = {"twin_ms": twin["ms"], "pair_GBs": round(2 * alg / (span * 1e-3) / 1e9, 1),
                                                       "pair_frac": round(2 * alg / (span * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)}
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "kernel": "convT4_stream_kernel<32> @ decoder.uplayer5.0.conv2 (ConvTranspose2d 16 -> 16 k4 s2, 32x32 -> 64x64 with the fused BatchNorm + ReLU "
                      "prologue; per-wave stream: input-row ring in LDS, 2x2 taps x 16 channels per MFMA K-step, 16-byte stores, BatchNorm sums in the pass): "
                      "the largest instance of the step's largest kernel family by GPU time.  `achieved` / `frac`: isolated mmvae_conv2d_fwd launches timed here; "
                      "`in_step`: the same launch in the committed trace, where it shares HBM with its shortcut twin on the side stream",
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "achieved_GBs": ach, "frac_of_hbm_peak": ach / HBM_PEAK_GBS,
            "in_step": in_step, "family_in_step": fam, "family_layers_isolated": rows}


def wgrad_stream_family_roofline(M, device, N, reps=10):
    """The wgrad_stream_kernel family (weight gradients of the big thin layers as per-wave streams: uplayer4's two fused backward passes on the
    caller's stream, encoder.layer1's conv2 and conv1 + shortcut pair on the side stream): its largest instance, the pair -- both weight gradients
    of encoder.layer1's 3x3 stride-2 conv1 and 1x1 stride-2 shortcut from ONE pass over the 32x32x32 block input -- timed in isolation through the
    C ABI (mmvae_conv2d_wgrad_pair: the kernel + its two ordered reduces).  Algorithmic bytes = x + dy + dy_shortcut (bf16), once."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    wsc = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    x = torch.randn(N, 32, 32, 32, device=device).to(torch.bfloat16)
    dy = torch.randn(N, 16, 16, 32, device=device).to(torch.bfloat16)
    dys = torch.randn(N, 16, 16, 32, device=device).to(torch.bfloat16)
    dw, dws = torch.zeros(32, 32, 3, 3, device=device), torch.zeros(32, 32, 1, 1, device=device)
    sc, sh = torch.rand(32, device=device) + 0.5, torch.randn(32, device=device) * 0.1

    def fn():
        L.check(lib.mmvae_conv2d_wgrad_pair(1, L.ptr(x), L.ptr(dy), L.ptr(dys), L.ptr(dw), L.ptr(dws), N, 32, 32, 32, 32, L.ptr(sc), L.ptr(sh), 1,
                                            L.ptr(wsc), st), "conv2d_wgrad_pair")
    ms = _time_calls(fn, reps, warm=2)
    alg = (x.numel() + dy.numel() + dys.numel()) * 2
    ach = alg / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic("wgrad_stream.encoder.layer1.pair", N)
    top = committed_top_kernels(N)
    fam = None if top is None else next((f for f in top["families"] if f["family"] == "wgrad_stream_kernel"), None)
    in_step = None
    if top is not None:
        me = next((l for l in top.get("largest", []) if "wgrad_stream_kernel<3, 2" in l["kernel"]), None)
        if me is not None:
            in_step = {"ms": me["ms"], "GBs": round(alg / (me["ms"] * 1e-3) / 1e9, 1), "frac": round(alg / (me["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                       "measured_hbm_MB": me.get("hbm_MB"), "stream": me.get("stream"), "source": top["source"],
                       "note": "the last launch of the step's side stream: it runs beside stem_bwd_kernel, which reads the same block input"}
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
            "kernel": "wgrad_stream_kernel<3x3 s2 + centre-tap companion> (+ 2 wgrad_reduce_kernel) @ encoder.layer1.conv1 + downsample.0: both weight gradients from one "
                      "read of the block input (per-wave strips, G-row ring + P rows in wave-private LDS, 10 tap accumulators x 2x2 channel tiles in registers, "
                      "per-block partial images, ordered reduce): the largest instance of the step's largest kernel family by GPU time.  `achieved` / `frac`: "
                      "isolated mmvae_conv2d_wgrad_pair calls timed here; `in_step`: the same launch in the committed trace",
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "achieved_GBs": ach, "frac_of_hbm_peak": ach / HBM_PEAK_GBS,
            "in_step": in_step, "family_in_step": fam}


# the layers whose forward / data-gradient launches run on pos_conv_kernel (maps up to 4x4, >= 128 output channels or 4x4 inputs) and on
# deep2_conv_kernel (the 64-channel layers on 8x8 / 16x16 maps): name, transposed, Cin, Cout, k, stride, pad, H (input side of the forward op)
CONV_FAMILY_LAYERS = {
    "pos_conv_kernel": [("encoder.layer3.0.conv2", 0, 128, 128, 3, 1, 1, 4), ("encoder.layer4.0.conv1", 0, 128, 256, 3, 2, 1, 4),
                        ("encoder.layer4.0.conv2", 0, 256, 256, 3, 1, 1, 2), ("decoder.uplayer1.0.conv2", 1, 128, 128, 4, 2, 1, 2),
                        ("decoder.uplayer2.0.conv2", 1, 64, 64, 4, 2, 1, 4), ("decoder.uplayer2.0.upsample.0", 1, 128, 64, 4, 2, 1, 4)],
    "deep2_conv_kernel": [("encoder.layer2.0.conv1", 0, 32, 64, 3, 2, 1, 16), ("encoder.layer2.0.conv2", 0, 64, 64, 3, 1, 1, 8),
                          ("encoder.layer3.0.conv1", 0, 64, 128, 3, 2, 1, 8), ("decoder.uplayer3.0.upsample.0", 1, 64, 32, 4, 2, 1, 8)],
}


def conv_family_roofline(M, device, N, family, reps=10):
    """The tile-kernel families of the channel-heavy forward / data-gradient convolutions (pos_conv_kernel, deep2_conv_kernel): every layer of the
    family timed in isolation through the C ABI (mmvae_conv2d_fwd with the BatchNorm + ReLU prologue and statistics, mmvae_conv2d_dgrad), the
    SLOWEST launch is the family's largest instance: algorithmic bytes = input + output activations (bf16) + weights, flops = 2 * pixels * Cin *
    Cout * k^2 (nominal: what the layer is, padded taps included); bound = mfma when flops / bytes exceeds the bf16 ridge (2.5 PF / 8 TB/s)."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for name, tr, Cin, Cout, k, sd, p, H in CONV_FAMILY_LAYERS[family]:
        Ho = (H - 1) * sd - 2 * p + k if tr else (H + 2 * p - k) // sd + 1
        w = torch.randn((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=device) * 0.05
        x = torch.randn(N, H, H, Cin, device=device).to(torch.bfloat16)
        y = torch.empty(N, Ho, Ho, Cout, device=device, dtype=torch.bfloat16)
        dx = torch.empty_like(x)
        scratch = torch.empty(4 * w.numel() + 1024, device=device, dtype=torch.uint8)
        sc, sh = torch.rand(Cin, device=device) + 0.5, torch.randn(Cin, device=device) * 0.1
        stats = torch.empty(4096 * 2 * Cout, device=device)

        def fwd(pack=False):
            L.check(lib.mmvae_conv2d_fwd(1, tr, L.ptr(x), L.ptr(w) if pack else None, L.ptr(y), N, H, H, Cin, Cout, k, sd, p, L.ptr(sc), L.ptr(sh), 1,
                                         L.ptr(stats), L.ptr(scratch), st), name)

        def dgrad():
            L.check(lib.mmvae_conv2d_dgrad(1, tr, L.ptr(y), L.ptr(w), L.ptr(dx), N, H, H, Cin, Cout, k, sd, p, L.ptr(scratch), st), name)
        fwd(True)
        flop = 2.0 * N * (H * H if tr else Ho * Ho) * Cin * Cout * k * k
        alg = (x.numel() + y.numel() + w.numel()) * 2
        rows.append(dict(layer=name, op="forward", ms=_time_calls(fwd, 6, warm=1), flop=flop, bytes=alg))
        rows.append(dict(layer=name, op="data gradient (+ its weight pack launch)", ms=_time_calls(dgrad, 6, warm=1), flop=flop, bytes=alg))
        del x, y, dx
    worst = max(rows, key=lambda r: r["ms"])
    ms, alg, flop = worst["ms"], worst["bytes"], worst["flop"]
    gbs, tfs = alg / (ms * 1e-3) / 1e9, flop / (ms * 1e-3) / 1e12
    mfma_bound = flop / alg > MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    top = committed_top_kernels(N)
    fam = None if top is None else next((f for f in top["families"] if f["family"] == family), None)
    return {"bound": "mfma" if mfma_bound else "hbm", "achieved": tfs if mfma_bound else gbs, "peak": MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if mfma_bound else "GB/s", "frac": (tfs / MFMA_PEAK_TFLOPS) if mfma_bound else (gbs / HBM_PEAK_GBS),
            "traffic": None, "traffic_source": None,
            "kernel": f"{family} @ {worst['layer']} {worst['op']}: the slowest isolated launch of the step's largest kernel family by GPU time in the committed "
                      f"profile ({len(rows)} launches of {len(rows) // 2} layers timed through mmvae_conv2d_fwd / mmvae_conv2d_dgrad)",
            "algorithmic_bytes_per_launch": alg, "algorithmic_flop_per_launch": flop, "avg_launch_ms": ms, "achieved_GBs": gbs, "achieved_TFLOPs": tfs,
            "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "frac_of_mfma_peak": tfs / MFMA_PEAK_TFLOPS, "family_in_step": fam,
            "family_launches_isolated": [{"layer": r["layer"], "op": r["op"], "ms": round(r["ms"], 4),
                                          "frac_of_mfma_peak": round(r["flop"] / (r["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 3)} for r in rows]}


def dominant_kernel_roofline(M, device, N, reps=10):
    """`roofline`: the kernel family with the most GPU time per step in the committed kernel statistics (profiles/*_top_kernels.json, taken on
    THIS build by tools/profile_round.sh) decides which kernel is reported: its largest instance, timed live in isolation.  Round 4: the weight
    gradients of the channel-heavy layers left wgrad2_kernel for wgrad_pos_kernel one by one, so the family that led rounds 2-3 may no longer
    lead; four families are within 8 % of each other now (wgrad_stream, deep2_conv, convT4_stream, pos_conv: 0.61-0.66 ms per step), and which of
    them leads a given profile is decided by tens of microseconds.  The weight-gradient table is kept beside it (`wgrad_families`)."""
    top = committed_top_kernels(N)
    fams = [] if top is None else [f for f in top["families"] if f.get("ms")]
    dominant = fams[0]["family"] if fams else "wgrad2_kernel"
    w2 = wgrad2_family_roofline(M, device, N, reps)
    if "convT4_stream" in dominant:
        out = convt4_family_roofline(M, device, N)
    elif "wgrad_stream" in dominant:
        out = wgrad_stream_family_roofline(M, device, N)
    elif dominant in CONV_FAMILY_LAYERS:
        out = conv_family_roofline(M, device, N, dominant)
    else:
        out = dict(w2)
        if "wgrad2" not in dominant:
            out["note"] = f"the committed profile's largest family is {dominant}; bench.py has no isolated probe for it: the wgrad2 family is reported"
    out["dominant_family_in_profile"] = None if not fams else {"family": dominant, "ms": fams[0]["ms"], "launches": fams[0].get("launches"),
                                                                 "source": top.get("source"), "same_build": top.get("same_build")}
    if out is not w2:
        keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "family_in_step", "family_layers_isolated_ms", "family_isolated_ms_sum",
                "wgrad_pos_layers_isolated")
        out["wgrad_families"] = {k: w2[k] for k in keys if k in w2}
    return out


def largest_launch_roofline(M, device, N, reps=10):
    """Second roofline object: the single longest launch of the step, join_bwd_stream_kernel -- the last up-block's backward in one pass
    (conv_joinbwd.hip): reads the two branch outputs (671 MB each at N = 5120), the one-plane reconstruction gradient and the two
    32x32 operands, writes the two 32x32 data gradients; dy2 / dys never exist.  Timed through mmvae_upblock_bwd_fused (events around the
    call: the kernel plus two weight packs, two partial-image reduces and one row sum, ~25 us of small launches)."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    bf = torch.bfloat16
    y2, ys = torch.randn(N, 64, 64, 16, device=device).to(bf), torch.randn(N, 64, 64, 16, device=device).to(bf)
    y1, xin = torch.randn(N, 32, 32, 16, device=device).to(bf), torch.randn(N, 32, 32, 16, device=device).to(bf)
    da1, gin = torch.empty_like(y1), torch.empty_like(xin)
    d_raw = torch.randn(N, 1, 64, 64, device=device)
    tw = torch.randn(1, 16, 3, 3, device=device) * 0.1
    w2, wu = torch.randn(16, 16, 4, 4, device=device) * 0.1, torch.randn(16, 16, 4, 4, device=device) * 0.1
    dw2, dwu = torch.zeros_like(w2), torch.zeros_like(wu)
    co = [torch.rand(16, device=device) + 0.5 for _ in range(12)]
    sums = torch.zeros(2, 16, device=device)
    wsc = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream().cuda_stream
    P = L.ptr

    def launch():
        L.check(lib.mmvae_upblock_bwd_fused(P(d_raw), P(tw), P(y2), P(co[0]), P(co[1]), P(ys), P(co[2]), P(co[3]), P(co[4]), P(co[5]), P(co[6]), P(co[7]),
                                            P(co[8]), P(co[9]), P(y1), P(co[10]), P(co[11]), P(w2), P(dw2), P(da1), P(sums), P(xin), None, None, P(wu),
                                            P(dwu), P(gin), N, P(wsc), st), "upblock_bwd_fused")
    ms = _time_calls(launch, reps)
    alg = 2 * y2.numel() * 2 + d_raw.numel() * 4 + 4 * y1.numel() * 2
    ach = alg / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic("uplayer5.join_bwd", N)
    top = committed_top_kernels(N)
    inst = None if top is None else next((k for k in top["largest"] if "join_bwd_stream_kernel" in k["kernel"]), None)
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": src,
            "kernel": "join_bwd_stream_kernel @ decoder.uplayer5: join BatchNorm backward + both ConvTranspose2d weight / data gradients + bn1 sums in one "
                      "pass, fused with the decoder.conv2 dgrad (isolated mmvae_upblock_bwd_fused calls, helper launches included)",
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms, "in_step": inst}


def step_traffic(N):
    """Whole-step HBM bytes (sum over every launch of one train step, FETCH_SIZE x 2 + WRITE_SIZE) from the committed PMC table
    of this build, against the ideal-fusion byte count the step roofline uses."""
    total, src = pmc_traffic("__step__", N)
    if total is None:
        return None
    ideal = TRAIN_BYTES_PER_FRAME * N
    return {"hbm_bytes_per_step": total, "ideal_fusion_bytes_per_step": ideal, "ratio": total / ideal, "source": src}


def spawn_ranks(n, argv):
    """Started without torch.distributed.run: launch the N ranks as a child job (this process has not touched the GPU and
    never will) and return the child's exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()              # does not initialise the GPU
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible -- refusing to run fewer ranks than asked for")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=["c2", "c4"],
                    help="c2 (default, the headline): BASELINE configs[1], reference depth, z=128, 256 clips; c4: BASELINE configs[3], the deeper "
                         "build-defined variant (2 residual blocks per stage), z=512, 512 clips")
    ap.add_argument("--clips", type=int, default=None, help="clips (of 20 frames) per GPU per step (default: 256 for c2, 512 for c4)")
    ap.add_argument("--z", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8 = BASELINE configs[4]: bf16 storage, fp8 (e4m3) MFMA in the forward pass of the deep layers (tests/test_fp8_gpu.py)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="BatchNorm statistics over the global batch (default: per rank)")
    ap.add_argument("--comm", default="torch", choices=["torch", "rccl"],
                    help="gradient exchange through torch.distributed's nccl(=RCCL) backend, or through the library's own RCCL "
                         "communicator (mmvae_comm_*)")
    a = ap.parse_args()
    blocks = 2 if a.config == "c4" else 1
    if a.clips is None:
        a.clips = 512 if a.config == "c4" else 256
    if a.z is None:
        a.z = 512 if a.config == "c4" else 128

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)
    pkg = importlib.import_module(PKG)
    M = importlib.import_module(PKG + ".model")

    torch.manual_seed(0)                      # identical initial weights on every rank (and broadcast below)
    model = M.VAE(1, 32, 1, 2, a.z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype=a.dtype, blocks_per_stage=blocks).to(device).train()
    opt = M.FusedAdam(list(model.parameters()))
    if world > 1:
        M.GradSync(model, sync_bn=a.sync_bn, comm=a.comm)
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    batch = synthetic_clips(a.clips, 1234 + rank, device)
    frames = a.clips * 20

    def run(k, b=None):
        return pkg.train(model, [batch if b is None else b] * k, opt, device, args, data_mean=DATA_MEAN, data_std=DATA_STD)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.warmup > 0:
        run(a.warmup)
    fence()
    t0 = time.perf_counter()
    losses = run(a.steps)[0]
    fence()
    dt = time.perf_counter() - t0
    ranks_seen = world
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
        one = torch.ones(1, device=device)
        dist.all_reduce(one)                  # the RCCL rank count actually taking part
        ranks_seen = int(one.item())
    if rank == 0:
        value = world * frames * a.steps / dt
        out = {
            "metric": METRIC, "value": value, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[{1 if a.config == 'c2' else 3}]: {a.clips} clips x 20 frames x 64x64 per GPU per step (={frames} frames), "
                                    f"conv-VAE{'' if blocks == 1 else f' ({blocks} residual blocks per stage: deeper, build-defined)'} z={a.z}, "
                                    f"Gaussian NLL sigma=0.1 + KL (normal_vae_1_kl_0_mmd), Adam, random-init weights, Bernoulli({P_ON}) q=2 labels"),
                       "global_frames_per_step": world * frames, "parallelism": f"dp{world}", "rccl_ranks": ranks_seen,
                       "grad_exchange": ("none" if world == 1 else ("mmvae_comm_allreduce (library RCCL communicator)" if a.comm == "rccl"
                                                                   else "torch.distributed nccl (= RCCL)")) ,
                       "bn": "global batch statistics (SyncBN)" if (a.sync_bn and world > 1) else "per-rank batch statistics"},
            "frames_per_sec_per_gpu": value / world,
            "final_loss": losses[-1],
            "step_hbm_roofline": {"algorithmic_bytes_per_frame": TRAIN_BYTES_PER_FRAME, "achieved_GBs": value / world * TRAIN_BYTES_PER_FRAME / 1e9,
                                  "frac_of_8TBs": value / world * TRAIN_BYTES_PER_FRAME / 1e9 / HBM_PEAK_GBS,
                                  "mfma_frac_of_2.5PF": value / world * TRAIN_FLOP_PER_FRAME / 2.5e15},
            "build": importlib.import_module(PKG + "._lib").build_hash(),
        }
        if a.config == "c4":
            # algorithmic work of the deeper variant per frame (conv / convT MACs x 2, forward + both gradients; counted from the layer
            # table like SURVEY 8d): the MFMA-bound stress configuration is judged against the bf16 MFMA peak
            fl = C4_TRAIN_FLOP_PER_FRAME[a.z] if a.z in C4_TRAIN_FLOP_PER_FRAME else None
            out["step_hbm_roofline"] = None
            out["step_mfma_roofline"] = None if fl is None else {"train_flop_per_frame": fl, "achieved_TFLOPs": value / world * fl / 1e12,
                                                                 "frac_of_2.5PF": value / world * fl / 2.5e15}
        out["step_traffic_ratio"] = step_traffic(frames) if a.config == "c2" else None
    if world == 1 and not a.no_roofline and a.config == "c2":
        # the other reading of "batch 256" (SURVEY 8): 256 FRAMES per step, same model -- a latency-bound point, for reference
        b256 = (torch.rand((256, 64, 64), generator=torch.Generator().manual_seed(99)) < P_ON).long().to(device)
        run(3, b256)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(20, b256)
        torch.cuda.synchronize()
        d256 = (time.perf_counter() - t1) / 20
        out["batch256_frames"] = {"frames_per_sec": 256 / d256, "ms_per_step": 1e3 * d256, "note": "256 frames (not clips) per step"}
    if world == 1 and not a.no_roofline and a.config == "c2" and a.dtype == "bf16":
        # compact records of BASELINE configs[3] (deeper net, z = 512, 512 clips) and configs[4] (fp8 mode of config 2) in the default line,
        # a few steps each (the full runs: --config c4, --dtype fp8)
        del model, opt
        torch.cuda.empty_cache()

        def sub_record(dtype, blocks_, z_, clips_, steps_):
            torch.manual_seed(0)
            m2 = M.VAE(1, 32, 1, 2, z_, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype=dtype, blocks_per_stage=blocks_).to(device).train()
            o2 = M.FusedAdam(list(m2.parameters()))
            b2 = synthetic_clips(clips_, 4321, device)
            pkg.train(m2, [b2] * 3, o2, device, args, data_mean=DATA_MEAN, data_std=DATA_STD)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ls = pkg.train(m2, [b2] * steps_, o2, device, args, data_mean=DATA_MEAN, data_std=DATA_STD)[0]
            torch.cuda.synchronize()
            d2 = (time.perf_counter() - t2) / steps_
            del m2, o2, b2
            torch.cuda.empty_cache()
            return d2, ls[-1]

        d4, l4 = sub_record("bf16", 2, 512, 512, 5)
        fl4 = C4_TRAIN_FLOP_PER_FRAME[512]
        out["config4_deeper"] = {"ms_per_step": 1e3 * d4, "frames_per_sec": 10240 / d4, "frames": 10240, "steps": 5, "final_loss": l4,
                                 "achieved_TFLOPs": 10240 / d4 * fl4 / 1e12, "frac_of_2.5PF": 10240 / d4 * fl4 / 2.5e15,
                                 "note": "BASELINE configs[3]: 2 residual blocks per stage, z=512, 512 clips x 20 frames, bf16"}
        # (like with like: both modes through the same 30-step sub-record -- a 10-step run carries ~5 % of start-up and drain that the 50-step
        # headline run does not)
        d5, l5 = sub_record("fp8", 1, a.z, a.clips, 30)
        d5b, _ = sub_record("bf16", 1, a.z, a.clips, 30)
        out["config5_fp8"] = {"ms_per_step": 1e3 * d5, "frames_per_sec": frames / d5, "steps": 30, "final_loss": l5,
                              "bf16_same_protocol_ms_per_step": 1e3 * d5b, "vs_bf16_same_run": d5b / d5,
                              "note": "BASELINE configs[4]: compute_dtype=fp8 (e4m3 MFMA in the forward convs of the >= 64-channel layers that run on deep2_conv_kernel -- "
                                      "8x8 maps; the 2x2 / 4x4-map layers stay on the bf16 position-major kernel, which is faster than the e4m3 form there; "
                                      "e4m3 STORAGE of the last up-block's two branch outputs; bf16 elsewhere)"}
        # PixelVAE: `categorical_pixelvae_1_kl_0_mmd` through select_model, the same 5120 frames -- a correctness path
        # (masked 7x7 convs on the generic kernels), recorded so that its cost is a number and not an adjective
        try:
            pargs = types.SimpleNamespace(model="categorical_pixelvae_1_kl_0_mmd", input_channels=1, input_image_size=64, intermediate_channels=16,
                                          z_dimension=a.z, sigma_decoder=0.0, require_rsample=True, num_pixelcnn_layers=3, pixelcnn_activation="ReLu",
                                          nll=1, quantization="2", decoder_out_channels=2, data_ratio_of_labels=torch.ones(2, device=device),
                                          dataset="MovingMNIST", quiet=True)
            torch.manual_seed(0)
            mp_, _ = pkg.select_model(pargs)
            mp_ = mp_.to(device).train()
            op_ = M.FusedAdam(list(mp_.parameters()))
            bp_ = synthetic_clips(a.clips, 4321, device)
            pkg.train(mp_, [bp_] * 2, op_, device, pargs, data_mean=DATA_MEAN, data_std=DATA_STD)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            lp = pkg.train(mp_, [bp_] * 3, op_, device, pargs, data_mean=DATA_MEAN, data_std=DATA_STD)[0]
            torch.cuda.synchronize()
            d6 = (time.perf_counter() - t3) / 3
            out["pixelvae"] = {"model": "categorical_pixelvae_1_kl_0_mmd (3 PixelCNN layers, 16 intermediate channels, q = 2)", "frames": frames, "steps": 3,
                               "ms_per_step": 1e3 * d6, "frames_per_sec": frames / d6, "final_loss": lp[-1],
                               "vs_vae_same_run": (1e3 * d6) / (1e3 * dt / a.steps)}
            del mp_, op_, bp_
            torch.cuda.empty_cache()
        except Exception as ex:  # noqa: BLE001 -- a sub-record must never take the headline down
            out["pixelvae"] = {"error": repr(ex)[:300]}
    if rank == 0:
        if not a.no_roofline and a.config == "c2":
            out["roofline"] = dominant_kernel_roofline(M, device, frames)
            if a.dtype == "bf16":
                out["roofline_largest_launch"] = largest_launch_roofline(M, device, frames)
            top = committed_top_kernels(frames)
            out["top_kernels"] = None if top is None else {"source": top["source"], "same_build": top["same_build"], "step_ms_traced": top["step_ms_traced"],
                                                           "hbm_GB": top.get("hbm_GB"), "families": top["families"][:12], "largest": top["largest"][:6]}
            out["elbo_rel_err_vs_cpu_oracle"] = elbo_check(M, device, a.dtype)
        if world == 1 and not a.no_cpu_baseline and a.config == "c2":
            out["cpu_baseline"] = cpu_baseline(a.z)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
