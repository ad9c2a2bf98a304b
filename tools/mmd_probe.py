"""Developer tool (GPU box): stand-alone timing of mmvae_mmd_fwd at the bench batch (n = 5120, d = 32)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("moving-mnist-vae_amd._lib"); lib = L.lib()
for n, d in ((5120, 32), (5120, 128), (2560, 32)):
    x = torch.randn(n, d, device="cuda"); y = torch.randn(n, d, device="cuda")
    scratch = torch.zeros(2 * n, device="cuda"); acc = torch.zeros(1, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): L.check(lib.mmvae_mmd_fwd(L.ptr(x), L.ptr(y), n, d, L.ptr(scratch), L.ptr(acc), st), "mmd")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.check(lib.mmvae_mmd_fwd(L.ptr(x), L.ptr(y), n, d, L.ptr(scratch), L.ptr(acc), st), "mmd")
    e1.record(); torch.cuda.synchronize()
    print(n, d, "us per call (incl. 2 row-norm launches):", e0.elapsed_time(e1) * 1000 / 20, flush=True)
