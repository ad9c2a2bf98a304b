"""Developer tool: eager train steps vs hipGraph replay of one captured step (config 2 shape).  usage: python tools/graph_bench.py [steps]"""
import importlib
import sys
import time
import types

import torch

sys.path.insert(0, ".")
M = importlib.import_module("moving-mnist-vae_amd.model")
main = importlib.import_module("moving-mnist-vae_amd.main")
dev = torch.device("cuda")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
torch.manual_seed(0)
m = M.VAE(1, 32, 1, 2, 128, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
opt = M.FusedAdam(list(m.parameters()), capturable=True)
args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
lab = (torch.rand((256, 20, 64, 64)) < 0.0521).long().to(dev)


def one_step():
    image, target = main.prepare_batch(m, lab, dev, args, 0.0521, 0.2222)
    mu, lv, enc, rec = m(image)
    loss = m.loss(target, mu, lv, enc, rec, dev, args, deferred=True)[0]
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach()


for _ in range(5):
    one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one_step()
torch.cuda.synchronize()
print(f"eager : {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step")
m.injected_eps = torch.randn(5120, 128, 1, 1, device=dev)
m.injected_true_samples = torch.randn(5120, 128, device=dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    one_step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = one_step()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    g.replay()
torch.cuda.synchronize()
print(f"graph : {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step   loss {out.item():.1f}")
