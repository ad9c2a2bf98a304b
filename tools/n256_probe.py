"""Developer probe: train steps at N = 256 frames (the other reading of "batch 256"), with progress lines."""
import importlib, sys, time, types, torch
sys.path.insert(0, ".")
pkg = importlib.import_module("moving-mnist-vae_amd"); M = importlib.import_module("moving-mnist-vae_amd.model")
dev = torch.device("cuda")
torch.manual_seed(0)
m = M.VAE(1, 32, 1, 2, 128, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, 64, compute_dtype="bf16").to(dev).train()
opt = M.FusedAdam(list(m.parameters()))
args = types.SimpleNamespace(data_ratio_of_labels=None, dataset="MovingMNIST", quiet=True)
for n in (int(a) for a in sys.argv[1:]):
    b = (torch.rand((n, 64, 64)) < 0.05).long().to(dev)
    print("N", n, "start", flush=True)
    out = pkg.train(m, [b] * 3, opt, dev, args, data_mean=0.05, data_std=0.22)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pkg.train(m, [b] * 20, opt, dev, args, data_mean=0.05, data_std=0.22); torch.cuda.synchronize()
    print("N", n, "ok", out[0], f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)
