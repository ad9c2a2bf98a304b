import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vae_oracle as O
M = importlib.import_module("moving-mnist-vae_amd.model")
N, z, S = 8, 32, 64
spec = O.state_spec(1, z, 1, S, True)
state = O.filled_state(spec, seed=0)
m = M.VAE(1, 32, 1, 2, z, False, False, 4, "ReLu", 1, 1, 0, True, 0.1, S, compute_dtype="f32")
m.load_state_dict(state); m.to("cuda").train()
image = O.normalise(O.synthetic_labels(N, S, seed=11), S)
torch.manual_seed(5); m.injected_eps = torch.randn(N, z, 1, 1).cuda()
out = m(image.cuda())
torch.cuda.synchronize()
ws = m._ws[True]
off = N * S * S * 4
y0 = ws[off: off + N * 32 * 32 * 32 * 4].view(torch.float32).view(N, 32, 32, 32).cpu()
ref = torch.nn.functional.conv2d(image, state["encoder.conv1.weight"], None, 2, 2).permute(0, 2, 3, 1)
print("y0 err", (y0 - ref).abs().max().item(), "ref max", ref.abs().max().item(), "sum", y0.double().sum().item())
x_t = ws[:off].view(torch.float32).view(N, S, S).cpu()
print("x_t err", (x_t - image[:, 0]).abs().max().item())
torch.save(dict(y0=y0, mu=out[0].cpu(), rec=out[3].cpu(), bn=m._bnf.cpu()), f"gpurun_out/stem_{os.environ.get('MMVAE_STEM_DIRECT','0')}.pt")
# ---- backward probe
import types
m.injected_true_samples = torch.randn(N, z).cuda()
caps = {}
mu, lv, enc, rec = out
rec.register_hook(lambda g: caps.__setitem__("d_recon", g.detach().cpu().clone()))
enc.register_hook(lambda g: caps.__setitem__("d_enc", g.detach().cpu().clone()))
mu.register_hook(lambda g: caps.__setitem__("d_mu", g.detach().cpu().clone()))
loss = m.loss(image.cuda(), mu, lv, enc, rec, torch.device("cuda"), types.SimpleNamespace())[0]
loss.backward()
torch.cuda.synchronize()
caps["loss"] = loss.detach().cpu()
for k, p in m.named_parameters():
    caps["g." + k] = p.grad.detach().cpu().clone()
torch.save(caps, f"gpurun_out/stemb_{os.environ.get('MMVAE_STEM_DIRECT','0')}.pt")
