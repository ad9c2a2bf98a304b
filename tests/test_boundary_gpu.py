"""GPU: pieces of the drop-in boundary that are not on the train step's critical path -- VAE.compute_kernel
(model.py:367-376), the RCCL communicator entry points (mmvae_comm_*, world of one on the test box; a real 2-rank RCCL job when
two GPUs are visible), and the graph-capturability the header promises for the stream-ordered entry points."""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _M():
    return importlib.import_module("moving-mnist-vae_amd.model")


def test_compute_kernel_matches_reference_formula(oracle):
    M = _M()
    m = M.VAE(1, 32, 1, 2, 32, False, False).to("cuda")
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(70, 32, generator=g), torch.randn(133, 32, generator=g) * 1.5
    k = m.compute_kernel(x.cuda(), y.cuda()).cpu()
    ref = oracle.compute_kernel(x, y)                      # restates model.py:367-376 (tiled x / y, mean over dim, / dim, exp)
    assert k.shape == (70, 133)
    assert (k - ref).abs().max().item() <= 2e-6
    mmd = m.compute_mmd(x.cuda(), x.cuda() * 0.5).item()
    ref64 = oracle.compute_mmd(x.double(), x.double() * 0.5).item()      # the three sums cancel: judge both against f64
    assert abs(mmd - ref64) <= 2e-5 * 70 * 70


def test_rccl_communicator_world_of_one():
    """mmvae_comm_unique_id / init / allreduce / destroy against the RCCL already loaded by PyTorch-ROCm (dlopen at run time)."""
    M = _M()
    c = M.Communicator(1, 0, M.Communicator.unique_id())
    t = torch.arange(1000, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c.all_reduce_(t)                                   # enqueued on the current (side) stream
    side.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.float32))
    c.destroy()


def _nccl_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        M = _M()
        for comm in ("torch", "rccl"):
            torch.manual_seed(0)
            m = M.VAE(1, 32, 1, 2, 32, False, False, compute_dtype="bf16").to(f"cuda:{rank}").train()
            opt = M.FusedAdam(list(m.parameters()))
            M.GradSync(m, comm=comm)
            x = torch.randn(8, 1, 64, 64, generator=torch.Generator().manual_seed(rank)).to(f"cuda:{rank}")
            mu, lv, enc, rec = m(x)
            loss = m.loss(x, mu, lv, enc, rec, x.device, None)[0]
            opt.zero_grad(); loss.backward(); opt.step()
            flat = m._flat.clone()
            dist.all_reduce(flat)
            assert torch.allclose(flat / world, m._flat, rtol=0, atol=0), comm      # identical parameters on every rank
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_gradsync_nccl_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, 29547, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_bench_refuses_more_ranks_than_gpus():
    import subprocess
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_train_step_is_hipgraph_capturable_and_replays_the_eager_trajectory(oracle):
    """include/mmvae.h promises stream-ordered, graph-capturable entry points (the library's side stream forks from and joins the
    caller's stream with events, nothing allocates or synchronises).  One whole train step -- labels -> forward -> loss -> backward
    (side-stream weight gradients included) -> Adam with its step count on the device -- is captured once with torch.cuda.graph
    and replayed three times on new inputs; losses and parameters must follow the eager run of the same three steps (f32 mode)."""
    import types
    M = _M()
    main = importlib.import_module("moving-mnist-vae_amd.main")
    dev = torch.device("cuda")
    N, z = 40, 32
    batches = [oracle.synthetic_labels(N, 64, seed=60 + i).to(dev) for i in range(3)]
    g = torch.Generator().manual_seed(2)
    noise = [(torch.randn(N, z, 1, 1, generator=g).to(dev), torch.randn(N, z, generator=g).to(dev)) for _ in range(3)]
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)

    def fresh():
        torch.manual_seed(5)
        m = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="f32").to(dev).train()
        return m, M.FusedAdam(list(m.parameters()), capturable=True)

    def one_step(m, opt, labels):
        image, target = main.prepare_batch(m, labels, dev, args, oracle.DATA_MEAN, oracle.DATA_STD)
        mu, lv, enc, rec = m(image)
        loss = m.loss(target, mu, lv, enc, rec, dev, args, deferred=True)[0]
        opt.zero_grad()                       # main.py:397-399
        loss.backward()
        opt.step()
        return loss.detach()

    # eager reference, twice (the step is bit-reproducible: ordered partial-row reductions everywhere, exact f64 loss accumulation)
    def eager_run():
        m, opt = fresh()
        ls = []
        for i in range(3):
            m.injected_eps, m.injected_true_samples = noise[i]
            ls.append(one_step(m, opt, batches[i]).item())
        return ls, {k: v.detach().clone() for k, v in m.named_parameters()}

    eager, p_eager = eager_run()
    eager2, p_eager2 = eager_run()
    # captured: static input buffers, warm-up on a side stream (creates the library's streams / events), state restored, capture
    m, opt = fresh()
    lab, e_s, t_s = batches[0].clone(), noise[0][0].clone(), noise[0][1].clone()
    m.injected_eps, m.injected_true_samples = e_s, t_s
    snap = (m._flat.detach().clone(), m._bnf.detach().clone(), m._bni.detach().clone())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        one_step(m, opt, lab)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.no_grad():
        m._flat.copy_(snap[0]); m._bnf.copy_(snap[1]); m._bni.copy_(snap[2])
        opt._m.zero_(); opt._v.zero_(); opt._step_dev.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = one_step(m, opt, lab)
    replayed = []
    for i in range(3):
        lab.copy_(batches[i]); e_s.copy_(noise[i][0]); t_s.copy_(noise[i][1])
        graph.replay()
        replayed.append(out.item())
    torch.cuda.synchronize()
    assert int(opt._step_dev.item()) == 3
    # No reduction of the step depends on launch timing (no float atomics; the f64 loss accumulation is exact), so a replayed graph must
    # reproduce the eager trajectory to the bit: losses, every parameter, both Adam moments.
    assert eager == eager2 and all(torch.equal(p_eager[k], p_eager2[k]) for k in p_eager), (eager, eager2)
    assert replayed == eager, (replayed, eager)
    assert replayed[2] < replayed[0]
    for k, v in m.named_parameters():
        assert torch.equal(v.detach(), p_eager[k]), (k, (v.detach() - p_eager[k]).abs().max().item())
    print(f"\ngraph replay == eager: losses {replayed}")


@pytest.mark.parametrize("dtype,categorical", [("f32", False), ("bf16", False), ("bf16", True)])
def test_train_step_is_bit_reproducible(oracle, dtype, categorical):
    """include/mmvae.h: no reduction of the train step ends in a float atomic -- partial images / partial rows are summed in a fixed
    order -- so the same step on the same inputs gives the same BITS: loss, every parameter gradient, the BatchNorm running statistics
    and the parameters after Adam.  The second run starts from a NaN-poisoned workspace and gradient buffer (a kernel that read
    memory it had not written, or summed in launch-timing order, would show), and a dummy kernel load on another stream shifts the
    timing of the library's side stream."""
    import types
    M = _M()
    main = importlib.import_module("moving-mnist-vae_amd.main")
    dev = torch.device("cuda")
    N, z = 48, 32
    labels = oracle.synthetic_labels(N, 64, seed=77).to(dev)
    g = torch.Generator().manual_seed(4)
    eps, ts = torch.randn(N, z, 1, 1, generator=g).to(dev), torch.randn(N, z, generator=g).to(dev)
    args = types.SimpleNamespace(data_ratio_of_labels=torch.tensor([0.0521, 0.9479]) if categorical else None, dataset="MovingMNIST", quiet=True)

    def run(poison):
        torch.manual_seed(5)
        m = M.VAE(1, 32, 2 if categorical else 1, 2, z, False, False, 4, "ReLu", 1, 1, 10 if not categorical else 0, True, 0.1, 64,
                  compute_dtype=dtype).to(dev).train()
        opt = M.FusedAdam(list(m.parameters()))
        m.injected_eps, m.injected_true_samples = eps, ts
        if poison:
            # allocate the workspace and both gradient buffers now and fill them with NaN; keep the second stream busy
            m._workspace(N, True).view(torch.float32).fill_(float("nan"))
            for i in (0, 1):
                m._G[i] = torch.full((m._n_params,), float("nan"), device=dev)
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                junk = torch.randn(4096, 4096, device=dev)
                for _ in range(20):
                    junk = junk @ junk * 1e-3
        image, target = main.prepare_batch(m, labels, dev, args, oracle.DATA_MEAN, oracle.DATA_STD)
        mu, lv, enc, rec = m(image)
        loss = m.loss(target, mu, lv, enc, rec, dev, args, deferred=True)[0]
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        return loss.detach().clone(), grads, m._flat.detach().clone(), m._bnf.detach().clone(), rec.detach().clone()

    a = run(False)
    b = run(True)
    assert torch.equal(a[4], b[4]), "reconstruction differs"
    assert torch.equal(a[0], b[0]), (a[0].item(), b[0].item())
    bad = [k for k in a[1] if not torch.equal(a[1][k], b[1][k])]
    assert not bad, f"gradients differ between two runs of the same step: {bad[:8]} ({len(bad)} of {len(a[1])})"
    assert torch.equal(a[3], b[3]), "BatchNorm running statistics differ"
    assert torch.equal(a[2], b[2]), "parameters after Adam differ"


def test_gaussian_loss_gradient_folded_into_the_decoder_backward(oracle):
    """VAE.loss's Gaussian NLL gradient is evaluated inside the output BatchNorm's backward (mmvae_decoder_bwd_gauss: no d_recon tensor)
    when the reconstruction handed to loss() is the decoder's own output; fuse_loss_tail = False, a cropped reconstruction (S = 56) or a
    reconstruction that went through another op take the materialised path.  Same gradients either way."""
    import types
    M = _M()
    dev = torch.device("cuda")
    args = types.SimpleNamespace(data_ratio_of_labels=None)
    for S, N in ((64, 12), (56, 6)):
        labels = oracle.synthetic_labels(N, S, seed=5)
        image = oracle.normalise(labels, S).to(dev)
        g = torch.Generator().manual_seed(9)
        eps, ts = torch.randn(N, 32, 1, 1, generator=g).to(dev), torch.randn(N, 32, generator=g).to(dev)
        grads = {}
        for mode in ("fused", "plain", "through_op"):
            torch.manual_seed(3)
            m = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype="f32").to(dev).train()
            m.fuse_loss_tail = mode != "plain"
            m.injected_eps, m.injected_true_samples = eps, ts
            mu, lv, enc, rec = m(image)
            if mode == "through_op":
                rec = rec * 1.0                       # a different tensor: the loss cannot fold its gradient
            loss = m.loss(image, mu, lv, enc, rec, dev, args)[0]
            loss.backward()
            assert m._pending_tail is None
            grads[mode] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        for k, ref in grads["plain"].items():
            scale = ref.abs().max().item() + 1e-30
            for mode in ("fused", "through_op"):
                err = (grads[mode][k] - ref).abs().max().item()
                assert err <= 2e-5 * scale + 1e-7, (S, mode, k, err, scale)
