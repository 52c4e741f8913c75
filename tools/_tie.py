import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops
dev = "cuda:0"; M = 65536
z = torch.randn(M, 256, device=dev)
for name, E in [("uniform 1/512", (torch.rand(512, 256, device=dev) * 2 - 1) / 512), ("uniform 1/16", (torch.rand(512, 256, device=dev) * 2 - 1) / 16),
                ("uniform 1", (torch.rand(512, 256, device=dev) * 2 - 1))]:
    pk = ops.vq_pack(E)
    for fast in (True, False):
        idx = ops.vq_argmin(z, E, fast=fast, packed=pk if fast else None); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): idx = ops.vq_argmin(z, E, fast=fast, packed=pk if fast else None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        if fast: fi = idx
        print(name, "fast" if fast else "exact", f"{dt*1e6:.0f} us", "match" if fast else bool((fi == idx).all()))
