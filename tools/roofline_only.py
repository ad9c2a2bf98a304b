import sys, importlib, torch
sys.path.insert(0, '/root/repo')
import bench
M = importlib.import_module('moving-mnist-vae_amd.model')
r = bench.dominant_kernel_roofline(M, torch.device('cuda'), 5120)
print({k: r[k] for k in ('achieved','frac','avg_launch_ms')})
