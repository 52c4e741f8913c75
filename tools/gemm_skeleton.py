"""Timing-only ablations of gemm_f16x2_pp_kernel (diagnostics build, DVQ_DIAG_LIB=1 DVQ_GEMM_ABL=<bits>) on one weight source:
time per launch against K -> cost per 32-wide K-tile (slope) and per tile outside the loop (intercept).  Bits: 2 no MFMA, 8 no weight
DMA, 16 no split + plane store, 32 no activation loads, 64 half the fragment reads, 128 no fragment reads, 256 no epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, packing
dev = "cuda:0"
M, N = 16384, 1024
res = []
for K in (512, 1024, 2048, 4096):
    x = (torch.randn(M, K, device=dev) * 0.1).contiguous(); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    pl = packing.split_planes(w); out = torch.empty(M, N, device=dev)
    for _ in range(5): ops.linear(x, w, b, out=out, planes=pl)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.linear(x, w, b, out=out, planes=pl)
    e1.record(); torch.cuda.synchronize()
    res.append((K, e0.elapsed_time(e1) * 1e3 / 30))
(k0, t0), (k1, t1) = res[1], res[3]
slope = (t1 - t0) / ((k1 - k0) / 32)           # us per K-tile per launch (two rounds of workgroups)
print("abl", os.environ.get("DVQ_GEMM_ABL", "0"), " ".join(f"K={k}: {t:.1f} us" for k, t in res),
      f"| per K-tile and round {slope / 2 * 1e3:.0f} ns, outside the loop {(t0 - slope * k0 / 32) / 2:.1f} us per round")
