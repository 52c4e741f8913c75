"""Diagnostic (GPU box, diagnostics build: make -C d-vqvae_amd/csrc diag): per-workgroup phase durations of the sixteen-wave VQ
streaming kernel from its s_memrealtime stamps (100 MHz): prologue, tile loop, last merges, refine tail."""
import os, sys
os.environ["DVQ_DIAG_LIB"] = "1"; os.environ["DVQ_VQ16_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dvqvae_amd
from dvqvae_amd import ops
dev = "cuda:0"
if len(sys.argv) > 1: os.environ["DVQ_VQ16_ABL"] = sys.argv[1]          # usage: vq16_phase_stamps.py [abl]
M, D, K = 65536, 256, 512
zs = [torch.randn(M, D, device=dev) for _ in range(6)]
E = torch.randn(K, D, device=dev)
pk = ops.vq_pack(E)
for i in range(12): ops.vq_argmin(zs[i % 6], E, packed=pk)
torch.cuda.synchronize()
ws = ops.workspace(0, torch.device(dev))
st = ws[: 256 * 64].view(torch.int64).view(256, 8).cpu().numpy().astype(np.float64)
t0 = st[:, 0].min()
ph = {"start skew": st[:, 0] - t0, "prologue": st[:, 1] - st[:, 0], "tile loop": st[:, 2] - st[:, 1], "last merges": st[:, 3] - st[:, 2],
      "refine + store": st[:, 4] - st[:, 3], "whole workgroup": st[:, 4] - st[:, 0]}
for k, v in ph.items():
    v = v * 0.01
    print(f"{k:16s} median {np.median(v):6.2f} us   p10 {np.percentile(v, 10):6.2f}   p90 {np.percentile(v, 90):6.2f}   max {v.max():6.2f}")
print("kernel span (first start .. last end): %.2f us" % ((st[:, 4].max() - t0) * 0.01))
print("pairs per workgroup: mean %.1f max %d; all-entries rows: %d" % (st[:, 5].mean(), st[:, 5].max(), st[:, 6].sum()))
