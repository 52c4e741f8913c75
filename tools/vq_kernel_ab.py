"""A/B of the two streaming VQ kernels in ONE process (GPU box): interleaved rounds, six rotating 64 MiB inputs, one HIP-event
pair around a train of 30 calls; both must return the same indices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvqvae_amd
from dvqvae_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()
M, D, K = 65536, 256, 512
zs = [torch.randn(M, D, device=dev) for _ in range(6)]
E = torch.randn(K, D, device=dev)
pk = ops.vq_pack(E)
ref = None
KERNELS = tuple(os.environ.get("VQ_AB_KERNELS", "32,16,8").split(","))
res = {k: [] for k in KERNELS}
for rnd in range(5):
    for kern in KERNELS:
        os.environ["DVQ_VQ_KERNEL"] = kern; lib.dvq_reload_env()
        for i in range(6): idx = ops.vq_argmin(zs[i], E, packed=pk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): idx = ops.vq_argmin(zs[i % 6], E, packed=pk)
        e1.record(); torch.cuda.synchronize()
        res[kern].append(e0.elapsed_time(e1) * 1e3 / 30)
        idx0 = ops.vq_argmin(zs[0], E, packed=pk)
        if ref is None: ref = idx0
        assert torch.equal(idx0, ref), f"kernel {kern} returns different indices"
for k, v in res.items():
    v = sorted(v)
    print(f"DVQ_VQ_KERNEL={k:2s}: median {v[len(v) // 2]:.2f} us, min {v[0]:.2f} us per call  ({[round(x, 2) for x in v]})")
