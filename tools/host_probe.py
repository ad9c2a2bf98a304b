"""Developer tool (GPU box): where the HOST spends its time in one train step of BASELINE configs[1] -- the loop of main.train with
perf_counter stamps between its phases and no synchronisation, then one synchronize.  host ms/step well under GPU ms/step = the GPU never
waits for the host; close to it = launch-bound stretches.   usage: python tools/host_probe.py [steps]"""
import importlib
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "moving-mnist-vae_amd"
bench = importlib.import_module("bench")


def main(steps=30):
    dev = torch.device("cuda", 0)
    pkg = importlib.import_module(PKG)
    M = importlib.import_module(PKG + ".model")
    main_mod = importlib.import_module(PKG + ".main")
    torch.manual_seed(0)
    model = M.VAE(1, 32, 1, 2, 128, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
    opt = M.FusedAdam(list(model.parameters()))
    args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
    batch = bench.synthetic_clips(256, 1234, dev)
    pkg.train(model, [batch] * 5, opt, dev, args, data_mean=bench.DATA_MEAN, data_std=bench.DATA_STD)
    torch.cuda.synchronize()
    names = ["prepare_batch", "forward", "loss", "zero_grad", "backward", "adam"]
    acc = [0.0] * len(names)
    t0 = time.perf_counter()
    for _ in range(steps):
        s = [time.perf_counter()]
        model.train(True)
        image, target = main_mod.prepare_batch(model, batch, dev, args, bench.DATA_MEAN, bench.DATA_STD)
        s.append(time.perf_counter())
        mu, logvar, enc, rec = model(image)
        s.append(time.perf_counter())
        loss, *_ = model.loss(target, mu, logvar, enc, rec, dev, args, deferred=True)
        s.append(time.perf_counter())
        opt.zero_grad()
        s.append(time.perf_counter())
        loss.backward()
        s.append(time.perf_counter())
        opt.step()
        s.append(time.perf_counter())
        for i in range(len(names)):
            acc[i] += s[i + 1] - s[i]
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    print(f"steps {steps}: host enqueue {1e3 * host / steps:.3f} ms/step, wall (host + drain) {1e3 * total / steps:.3f} ms/step")
    for n, v in zip(names, acc):
        print(f"  {n:14s} {1e3 * v / steps:7.3f} ms/step (host)")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 30)
