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
Besides the contract's keys the line carries ``roofline`` (the dominant kernel family's largest instance, isolated launches
timed with events on the launch stream), ``roofline_largest_launch`` (the single longest launch of the step, same method),
``step_hbm_roofline`` (whole step against the ideal-fusion byte count) and ``cpu_baseline`` (the CPU oracle on a bounded sample).
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


def dominant_kernel_roofline(M, device, N, reps=20):
    """HBM roofline of the dominant kernel class, measured live with events on the launch stream: the streaming
    ConvTranspose2d forward at the decoder's widest layer (decoder.uplayer5.0.conv2, ConvTranspose2d 16->16 k4 s2, 32^2 -> 64^2).
    Algorithmic bytes per launch = input + output activations (bf16) + weights, each touched once."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    Cin = Cout = 16
    H = 32
    x = torch.randn(N, H, H, Cin, device=device).to(torch.bfloat16)
    w = torch.randn(Cin, Cout, 4, 4, device=device) * 0.1
    y = torch.empty(N, 2 * H, 2 * H, Cout, device=device, dtype=torch.bfloat16)
    scratch = torch.empty(2 * w.numel() * 2 + 256, dtype=torch.uint8, device=device)
    stats = torch.zeros(4096 * 2 * Cout, device=device)
    st = torch.cuda.current_stream().cuda_stream

    def launch(weights=None):
        L.check(lib.mmvae_conv2d_fwd(1, 1, L.ptr(x), L.ptr(weights), L.ptr(y), N, H, H, Cin, Cout, 4, 2, 1, None, None, 0, L.ptr(stats),
                                     L.ptr(scratch), st), "conv2d_fwd")
    launch(w)                  # packs the weights into `scratch`; the timed launches below are the conv kernel alone
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = x.numel() * 2 + y.numel() * 2 + w.numel() * 2
    ach = alg / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic("uplayer5.conv2.fwd", N)
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": src,
            "kernel": "convT4_stream_kernel<32> @ decoder.uplayer5.0.conv2 (ConvTranspose2d 16->16 k4 s2, 32x32 -> 64x64; per-wave stream: input-row ring in LDS, 2x2 taps x 16 channels per MFMA K-step, 16-byte stores; isolated launches, cold input)",
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms}


def largest_launch_roofline(M, device, N, reps=10):
    """Second roofline object: the single longest launch of the step, the last up-block's join-backward apply pass
    (tail_apply_mfma_kernel): reads the two branch outputs, recomputes the incoming gradient from the 1-plane
    d_raw, writes dy2 / dys.  Algorithmic bytes per launch = 4 x (N*64*64*16 bf16) + d_raw once (f32)."""
    L = importlib.import_module(PKG + "._lib")
    lib = L.lib()
    H = 64
    bf = torch.bfloat16
    y2 = torch.randn(N, H, H, 16, device=device).to(bf)
    ys = torch.randn(N, H, H, 16, device=device).to(bf)
    dy2, dys = torch.empty_like(y2), torch.empty_like(ys)
    d_raw = torch.randn(N, 1, H, H, device=device)
    w = torch.randn(1, 16, 3, 3, device=device) * 0.1
    co = [torch.rand(16, device=device) + 0.5 for _ in range(10)]
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        L.check(lib.mmvae_tail_join_bwd_apply(1, L.ptr(d_raw), L.ptr(w), 1, L.ptr(y2), L.ptr(co[0]), L.ptr(co[1]), L.ptr(ys), L.ptr(co[2]),
                                              L.ptr(co[3]), L.ptr(co[4]), L.ptr(co[5]), L.ptr(co[6]), L.ptr(co[7]), L.ptr(co[8]), L.ptr(co[9]),
                                              L.ptr(dy2), L.ptr(dys), N, H, H, st), "tail_join_bwd_apply")
    for _ in range(2):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = 4 * y2.numel() * 2 + d_raw.numel() * 4
    ach = alg / (ms * 1e-3) / 1e9
    traffic, src = pmc_traffic("uplayer5.join_bwd_apply", N)
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": src,
            "kernel": "tail_apply_mfma_kernel @ decoder.uplayer5 join backward (apply pass) fused with the decoder.conv2 dgrad (isolated launches)",
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms}


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
    if rank == 0:
        if not a.no_roofline and a.config == "c2":
            out["roofline"] = dominant_kernel_roofline(M, device, frames)
            if a.dtype == "bf16":
                out["roofline_largest_launch"] = largest_launch_roofline(M, device, frames)
            out["elbo_rel_err_vs_cpu_oracle"] = elbo_check(M, device, a.dtype)
        if world == 1 and not a.no_cpu_baseline and a.config == "c2":
            out["cpu_baseline"] = cpu_baseline(a.z)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
