"""Data-parallel path: one process per device, flat-gradient buckets, sum all-reduce, 1/world folded
into Adam.  CPU part: the exchange logic of GradSync over gloo, world_size 2.  GPU part: two full HIP ranks sharing the
one test GPU (gloo moves CUDA tensors through the host; on the 8-GPU node the backend is "nccl" = RCCL over xGMI)."""
import importlib
import os
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


# ---------------------------------------------------------------------------------------------- CPU: exchange logic
def _cpu_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        M = importlib.import_module("moving-mnist-vae_amd.model")
        n, dec_off = 1000, 600
        fake = types.SimpleNamespace(_flat=torch.full((n,), float(rank + 1)), _bnf=torch.full((7,), float(rank)), _sync=None,
                                     _dec_off=dec_off, _n_params=n, _ensure_flat=lambda: None)
        sync = M.GradSync(fake)
        assert fake._sync is sync and sync.world == world
        assert torch.all(fake._flat == 1.0) and torch.all(fake._bnf == 0.0)           # rank 0's values everywhere
        G = torch.arange(n, dtype=torch.float32) * (rank + 1)
        sync.bucket_ready(G, dec_off, n)          # decoder gradients are complete first
        sync.bucket_ready(G, 0, dec_off)
        scale = sync.finish(G)
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        assert scale == 1.0 / world
        assert torch.equal(G, expect)
        # a bucket that was never announced (e.g. a frozen decoder) is still reduced by finish()
        G2 = torch.ones(n) * (rank + 1)
        sync.bucket_ready(G2, dec_off, n)
        sync.finish(G2)
        assert torch.all(G2 == 3.0)
        # logged step scalars: one all-reduce of the stacked (steps, 4) tensor, mean over ranks
        groups = [M._StepScalars(torch.tensor([1.0, 2.0, 3.0, 4.0]) * (rank + 1) * (k + 1)) for k in range(3)]
        sync.reduce_step_scalars(groups)
        for k, g in enumerate(groups):
            assert g.get(0) == 1.5 * (k + 1) and g.get(3) == 6.0 * (k + 1), g.vals
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_gradsync_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, 29531, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---------------------------------------------------------------------------------------------- GPU: two HIP ranks
def _gpu_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from oracle import vae_oracle as O
        pkg = importlib.import_module("moving-mnist-vae_amd")
        M = importlib.import_module("moving-mnist-vae_amd.model")
        dev = torch.device("cuda:0")
        z, S, N = 32, 64, 8
        spec = O.state_spec(1, z, 1, S, True)
        state = O.filled_state(spec, seed=0)
        args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
        batches = [O.synthetic_labels(N, S, seed=100 + r).view(N, S * S) for r in range(world)]
        noise = []
        for r in range(world):
            g = torch.Generator().manual_seed(500 + r)
            noise.append((torch.randn(N, z, 1, 1, generator=g), torch.randn(N, z, generator=g)))

        def fresh(sync):
            m = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype="f32")
            m.load_state_dict(state if (rank == 0 or not sync) else {k: v * 0 + 1 for k, v in state.items()})  # non-zero ranks start wrong on purpose
            m.to(dev).train()
            return m

        # ---- data-parallel step: every rank its own batch, broadcast from rank 0, averaged gradients
        m = fresh(True)
        M.GradSync(m)
        opt = M.FusedAdam(list(m.parameters()))
        m.injected_eps, m.injected_true_samples = noise[rank][0].to(dev), noise[rank][1].to(dev)
        pkg.train(m, [batches[rank]], opt, dev, args, data_mean=O.DATA_MEAN, data_std=O.DATA_STD)
        torch.cuda.synchronize()
        mine = m._flat.detach().cpu()
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        for r in range(1, world):
            assert torch.equal(gathered[0], gathered[r]), "ranks diverged"
        if rank == 0:
            # ---- expected: gradients of each shard computed one after the other on the same weights, averaged, one Adam step
            grads = []
            for r in range(world):
                mr = fresh(False)
                mr.injected_eps, mr.injected_true_samples = noise[r][0].to(dev), noise[r][1].to(dev)
                img, tgt = pkg.main.prepare_batch(mr, batches[r], dev, args, O.DATA_MEAN, O.DATA_STD)
                out = mr(img)
                loss = mr.loss(tgt, *out, dev, args)[0]
                loss.backward()
                grads.append(torch.cat([p.grad.reshape(-1) for p in mr.parameters()]).clone())
            ref = fresh(False)
            p0 = ref._flat.detach().clone()
            gavg = sum(grads) / world
            # Adam, first step: p - lr * g / (|g| + eps)   (bias-corrected m/sqrt(v) = g/|g|)
            expect = p0 - 1e-3 * gavg / (gavg.abs() + 1e-8)
            err = (mine.to(dev) - expect).abs().max().item()
            assert err < 2e-5, err
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-1500:]))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.gpu
def test_two_hip_ranks_average_gradients():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, 29532, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _syncbn_worker(rank, world, port, q):
    """Two ranks, 4 frames each, BatchNorm statistics over the global batch (GradSync(sync_bn=True)) against ONE process
    running the 8 frames: same loss, same averaged gradients (f32 kernels; rel 2e-4 of each tensor's max, the summation order
    of the statistics differs), same BatchNorm running statistics."""
    try:
        _init(rank, world, port)
        from oracle import vae_oracle as O
        pkg = importlib.import_module("moving-mnist-vae_amd")
        M = importlib.import_module("moving-mnist-vae_amd.model")
        dev = torch.device("cuda:0")
        z, S, N = 32, 64, 8
        n_loc = N // world
        spec = O.state_spec(1, z, 1, S, True)
        state = O.filled_state(spec, seed=0)
        args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
        labels = O.synthetic_labels(N, S, seed=321).view(N, S * S)
        g = torch.Generator().manual_seed(77)
        eps, ts = torch.randn(N, z, 1, 1, generator=g), torch.randn(N, z, generator=g)

        def run(m, lab, e, t):
            m.injected_eps, m.injected_true_samples = e.to(dev), t.to(dev)
            img, tgt = pkg.main.prepare_batch(m, lab, dev, args, O.DATA_MEAN, O.DATA_STD)
            out = m(img)
            loss = m.loss(tgt, *out, dev, args)[0]
            m.zero_grad()
            loss.backward()
            torch.cuda.synchronize()
            return float(loss), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).detach().clone()

        m = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype="f32")
        m.load_state_dict(state)
        m.to(dev).train()
        sync = M.GradSync(m, sync_bn=True)
        sl = slice(rank * n_loc, (rank + 1) * n_loc)
        loss_r, grad_r = run(m, labels[sl], eps[sl], ts[sl])
        G = m._grad_target()
        scale = sync.finish(G)                       # sums the buckets over the ranks; Adam would apply `scale`
        torch.cuda.synchronize()
        gavg = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).detach().cpu() * scale
        lsum = torch.tensor([loss_r])
        dist.all_reduce(lsum)
        bn = m._bnf.detach().cpu().clone()
        if rank == 0:
            ref = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype="f32")
            ref.load_state_dict(state)
            ref.to(dev).train()
            loss_1, grad_1 = run(ref, labels, eps, ts)
            # MMD is a cross-sample term evaluated per rank; its coefficient is 0 here, so the losses agree
            assert abs(lsum.item() / world - loss_1) <= 2e-5 * abs(loss_1), (lsum.item() / world, loss_1)
            names = [n for n, _ in ref.named_parameters()]
            off, bad = 0, {}
            for n_, p in zip(names, ref.parameters()):
                a, b = gavg[off:off + p.numel()], grad_1[off:off + p.numel()].cpu()
                off += p.numel()
                if n_ == "decoder.conv2.bias":
                    continue
                err = float((a - b).abs().max())
                if err > 2e-4 * float(b.abs().max()) + 1e-9:
                    bad[n_] = (err, float(b.abs().max()))
            assert not bad, bad
            assert float((bn - ref._bnf.detach().cpu()).abs().max()) < 1e-4
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()[-2500:]))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.gpu
def test_sync_bn_two_ranks_match_one_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, 29541, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


# ---------------------------------------------------------------------------------------------- CPU: control flow of the rccl path
def test_gradsync_communicator_path_orders_streams_without_a_peer(monkeypatch):
    """GradSync(comm=<Communicator>) issues the decoder bucket from a communication stream that (1) waits for the caller's stream, (2) is
    ordered behind the library's side stream through mmvae_net_join, and only then (3) launches the collective; the encoder bucket follows on
    the same stream and finish() makes the caller's stream wait for it before Adam reads the sum.  No GPU, no peer: streams, the library
    and the communicator are recording fakes -- what is tested is the ORDER of the calls, which no single-GPU run can exercise with a peer."""
    M = importlib.import_module("moving-mnist-vae_amd.model")
    log = []

    class FakeStream:
        def __init__(self, name="comm", device=None):
            self.name, self.cuda_stream = name, 0xC0FFEE if name == "comm" else 0x1
        def wait_stream(self, other):
            log.append(("wait_stream", self.name, other.name))

    caller = FakeStream("caller")
    active = [caller]

    class StreamCtx:
        def __init__(self, st): self.st = st
        def __enter__(self): active.append(self.st); log.append(("enter", self.st.name))
        def __exit__(self, *a): active.pop(); log.append(("exit", self.st.name))

    class FakeLib:
        def mmvae_net_join(self, h, stream):
            log.append(("net_join", h, stream, active[-1].name))
            return 0

    class FakeComm:
        def all_reduce_(self, t, stream=None):
            log.append(("all_reduce", int(t.numel()), active[-1].name))
            t.mul_(2.0)                                     # "sum over two identical ranks"

    monkeypatch.setattr(M.torch.cuda, "Stream", lambda device=None: FakeStream("comm"))
    monkeypatch.setattr(M.torch.cuda, "current_stream", lambda device=None: caller)
    monkeypatch.setattr(M.torch.cuda, "stream", lambda st: StreamCtx(st))
    monkeypatch.setattr(M, "lib", lambda: FakeLib())
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    n, dec_off = 1000, 600
    fake = types.SimpleNamespace(_flat=torch.zeros(n), _bnf=torch.zeros(7), _sync=None, _dec_off=dec_off, _n_params=n, _h=0xAB,
                                 _ensure_flat=lambda: None)
    sync = M.GradSync(fake, broadcast=False, comm=FakeComm())
    G = torch.ones(n)
    sync.bucket_ready(G, dec_off, n, side_of=fake)           # what _DecoderFn.backward does with a deferred join
    sync.bucket_ready(G, 0, dec_off)                         # _EncoderFn.backward
    scale = sync.finish(G)
    assert scale == 0.5 and torch.all(G == 2.0)
    assert log == [
        ("wait_stream", "comm", "caller"), ("enter", "comm"), ("net_join", 0xAB, 0xC0FFEE, "comm"), ("all_reduce", n - dec_off, "comm"), ("exit", "comm"),
        ("wait_stream", "comm", "caller"), ("enter", "comm"), ("all_reduce", dec_off, "comm"), ("exit", "comm"),
        ("wait_stream", "caller", "comm"),
    ], log
    # a bucket nobody announced is reduced by finish() itself, after the wait
    log.clear()
    G2 = torch.ones(n)
    sync.bucket_ready(G2, dec_off, n, side_of=fake)
    sync.finish(G2)
    assert [e[0] for e in log] == ["wait_stream", "enter", "net_join", "all_reduce", "exit", "wait_stream", "all_reduce"] and torch.all(G2 == 2.0), log
