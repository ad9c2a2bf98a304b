"""Developer tool (GPU box): a few train steps of `categorical_pixelvae_1_kl_0_mmd` (bench.py's `pixelvae` sub-record) for
rocprofv3 --kernel-trace --stats.   usage: python tools/pixel_probe.py [clips=256] [steps=3]"""
import importlib
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("moving-mnist-vae_amd")
M = importlib.import_module("moving-mnist-vae_amd.model")


def main(clips=256, steps=3):
    dev = torch.device("cuda")
    args = types.SimpleNamespace(model="categorical_pixelvae_1_kl_0_mmd", input_channels=1, input_image_size=64, intermediate_channels=16, z_dimension=128,
                                 sigma_decoder=0.0, require_rsample=True, num_pixelcnn_layers=3, pixelcnn_activation="ReLu", nll=1, quantization="2",
                                 decoder_out_channels=2, data_ratio_of_labels=torch.ones(2, device=dev), dataset="MovingMNIST", quiet=True)
    torch.manual_seed(0)
    m, _ = pkg.select_model(args)
    m = m.to(dev).train()
    opt = M.FusedAdam(list(m.parameters()))
    batch = (torch.rand((clips, 20, 64, 64), generator=torch.Generator().manual_seed(1)) < 0.0521).long().to(dev)
    pkg.train(m, [batch] * 2, opt, dev, args, data_mean=0.0521, data_std=0.2222)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    pkg.train(m, [batch] * steps, opt, dev, args, data_mean=0.0521, data_std=0.2222)
    torch.cuda.synchronize()
    print(f"pixelvae: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step at {clips * 20} frames")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
