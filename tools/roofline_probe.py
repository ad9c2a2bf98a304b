import importlib, sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
M = importlib.import_module("moving-mnist-vae_amd.model")
r = bench.dominant_kernel_roofline(M, torch.device("cuda"), 5120)
print(os.environ.get("MMVAE_DBG"), os.environ.get("MMVAE_GATHER3_MAXK"), "ms", round(r["avg_launch_ms"],4), "GB/s", round(r["achieved"],1))
