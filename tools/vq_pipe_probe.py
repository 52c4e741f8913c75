"""GPU box: one problem size per process (a fault kills the process): vq_pipe.hip against the exact kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvqvae_amd
from dvqvae_amd import ops, _lib
lib = _lib.load()
M = int(sys.argv[1])
os.environ["DVQ_VQ_KERNEL"] = "17"; lib.dvq_reload_env()
torch.manual_seed(3)
E = torch.randn(512, 256, device="cuda:0")
z = torch.randn(M, 256, device="cuda:0")
pk = ops.vq_pack(E)
a = ops.vq_argmin(z, E, packed=pk, fast=True)
torch.cuda.synchronize()
b = ops.vq_argmin(z, E, fast=False)
print(f"M={M} mismatches={(a != b).sum().item()}", flush=True)
