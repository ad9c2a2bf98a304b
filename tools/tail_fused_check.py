import importlib, os, sys, subprocess, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
def run(S, dt, n=int(os.environ.get('CHK_N', '8'))):
    M = importlib.import_module("moving-mnist-vae_amd.model")
    torch.manual_seed(0)
    m = M.VAE(1, 32, 1, 2, 32, False, False, 4, "ReLu", 1, 1.0, 0.0, True, 1.0, S, compute_dtype=dt).to("cuda").train()
    x = (torch.rand(n, 1, S, S, device="cuda") > 0.7).float()
    m.injected_eps = torch.randn(n, 32, 1, 1, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    mu, lv, enc, rec = m(x)
    return rec.detach().float().cpu().numpy()
if __name__ == "__main__":
    if len(sys.argv) > 1:
        S, dt, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
        np.save(out, run(S, dt)); sys.exit(0)
    for S in (32, 64):
        for dt in ("f32", "bf16"):
            outs = []
            for fused in ("1", "0"):
                env = dict(os.environ, MMVAE_TAIL_FWD_FUSED=fused)
                f = f"/tmp/rec_{S}_{dt}_{fused}.npy"
                subprocess.run([sys.executable, __file__, str(S), dt, f], env=env, check=True)
                outs.append(np.load(f))
            d = np.abs(outs[0] - outs[1])
            idx = np.unravel_index(d.argmax(), d.shape)
            print(S, dt, "max diff", d.max(), "at", idx, "rows with diff>1e-2:", sorted(set(np.argwhere(d > 1e-2)[:, 2].tolist()))[:40], flush=True)
