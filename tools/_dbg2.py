import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, _lib
os.environ["DVQ_VQ_DBG"] = "1"
dev = "cuda:0"; M = 65536
lib = _lib.load()
nws = lib.dvq_vq_fast_workspace_bytes(M, 512, 256)
for seed in range(4):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    z = torch.randn(M, 256, device=dev, generator=g); E = torch.randn(512, 256, device=dev, generator=g)
    pk = ops.vq_pack(E)
    for _ in range(3): idx = ops.vq_argmin(z, E, packed=pk)
    torch.cuda.synchronize()
    ws = ops.workspace(nws, torch.device(dev))
    n_wg = M // 128
    full = np.frombuffer(ws[:n_wg * 64].cpu().numpy().tobytes(), dtype=np.uint64).reshape(n_wg, 8).astype(np.int64)
    rel = (full[:, :4] - full[:, 0].min()) * 0.01
    ep = rel[:, 3] - rel[:, 2]; nov = full[:, 5]
    ex = ops.vq_argmin(z, E, fast=False)
    print(f"seed {seed}: end med/max {np.median(rel[:,3]):.1f} {rel[:,3].max():.1f}; overflow rows {nov.sum()}; overflow-WG tail max {ep[nov>0].max() if (nov>0).any() else 0:.1f}; other tail max {ep[nov==0].max():.1f}; match {(idx==ex).float().mean().item()}")
