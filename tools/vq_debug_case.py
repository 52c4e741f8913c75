"""Diagnostic (GPU box): fast vs exact kernel on a few regimes, with and without the DVQ_VQ_DBG instantiation."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for name, M, Efn in (("normal", 512, lambda: torch.randn(512, 256, device=dev)),
                     ("tie-prone", 512, lambda: (torch.rand(512, 256, device=dev) * 2 - 1) / 512),
                     ("tie-prone", 33, lambda: (torch.rand(512, 256, device=dev) * 2 - 1) / 512),
                     ("tie-prone", 4096, lambda: (torch.rand(512, 256, device=dev) * 2 - 1) / 512),
                     ("normal", 65536, lambda: torch.randn(512, 256, device=dev))):
    E = Efn()
    z = torch.randn(M, 256, device=dev)
    ex = ops.vq_argmin(z, E, fast=False)
    for dbg in (False, True):
        if dbg: os.environ["DVQ_VQ_DBG"] = "1"
        else: os.environ.pop("DVQ_VQ_DBG", None)
        bad = []
        for rep in range(5):
            slow = torch.zeros(1, dtype=torch.int64, device=dev)
            f = ops.vq_argmin(z, E, fast=True, slow_rows=slow)
            torch.cuda.synchronize()
            bad.append(int((f != ex).sum()))
        wrong = (f != ex).nonzero().flatten()[:8].tolist()
        print(f"{name} M={M} dbg={dbg}: mismatches per rep {bad} slow rows {int(slow)} first wrong rows {wrong}")
