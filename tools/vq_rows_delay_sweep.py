"""Sweep of vq_rows.hip's stagger delay (GPU box): DVQ_VQ_KERNEL=32 with DVQ_VQ_ROWS_DELAY = 0 .. 900 (10 ns ticks) against the default
kernel, six rotating 64 MiB inputs, one HIP-event pair around a train of 30 calls; every setting must return the same indices."""
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
settings = [("16", "0")] + [("32", d) for d in sys.argv[1:] or ("0", "200", "350", "500", "650", "800")]
res = {s: [] for s in settings}
for rnd in range(4):
    for s in settings:
        os.environ["DVQ_VQ_KERNEL"], os.environ["DVQ_VQ_ROWS_DELAY"] = s; lib.dvq_reload_env()
        for i in range(6): idx = ops.vq_argmin(zs[i], E, packed=pk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): idx = ops.vq_argmin(zs[i % 6], E, packed=pk)
        e1.record(); torch.cuda.synchronize()
        res[s].append(e0.elapsed_time(e1) * 1e3 / 30)
        idx0 = ops.vq_argmin(zs[0], E, packed=pk)
        if ref is None: ref = idx0
        assert torch.equal(idx0, ref), f"setting {s} returns different indices"
for s, v in res.items():
    v = sorted(v)
    print(f"kernel {s[0]:2s} delay {s[1]:>4s}: median {v[len(v) // 2]:.2f} us, min {v[0]:.2f} us per call")
