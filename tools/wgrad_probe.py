"""Developer tool (GPU box): weight gradients of the big thin layers timed in isolation through the C ABI.
usage: python tools/wgrad_probe.py [N=5120] [reps=10]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("moving-mnist-vae_amd._lib")

# name, transposed, Cin, Cout, k, s, p, H (input side of the forward op), prologue on x
LAYERS = [
    ("dec.up5.conv2", 1, 16, 16, 4, 2, 1, 32, 1),
    ("dec.up5.upsample", 1, 16, 16, 4, 2, 1, 32, 0),
    ("dec.up5.conv1", 0, 16, 16, 1, 1, 0, 32, 0),
    ("dec.up4.conv2", 1, 16, 16, 4, 2, 1, 16, 1),
    ("dec.up4.upsample", 1, 32, 16, 4, 2, 1, 16, 0),
    ("enc.layer1.conv1", 0, 32, 32, 3, 2, 1, 32, 1),
    ("enc.layer1.conv2", 0, 32, 32, 3, 1, 1, 16, 1),
    ("enc.layer1.down", 0, 32, 32, 1, 2, 0, 32, 1),
]


def main(N=5120, reps=10):
    lib = L.lib()
    dev = torch.device("cuda")
    s = torch.cuda.current_stream().cuda_stream
    wsc = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    print(f"{'layer':20s} {'us':>8s} {'MB':>8s} {'TB/s':>6s}")
    for name, tr, Cin, Cout, k, st, p, H, pro in LAYERS:
        Ho = (H - 1) * st - 2 * p + k if tr else (H + 2 * p - k) // st + 1
        x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(N, Ho, Ho, Cout, device=dev).to(torch.bfloat16)
        dw = torch.zeros((Cin, Cout, k, k) if tr else (Cout, Cin, k, k), device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5
        sh = torch.randn(Cin, device=dev) * 0.1
        mb = (x.numel() + dy.numel()) * 2 / 1e6

        def fn():
            L.check(lib.mmvae_conv2d_wgrad(1, tr, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, H, H, Cin, Cout, k, st, p,
                                           sc.data_ptr() if pro else None, sh.data_ptr() if pro else None, 1, wsc.data_ptr(), s), name)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{name:20s} {us:8.1f} {mb:8.1f} {mb / us:6.2f}")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
