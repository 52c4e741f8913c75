"""GEMM microbench (GPU box): TFLOP/s of the default GEMM arithmetic (fp16 three-product split; DVQ_GEMM=bf16x3: the six-product
bf16 split) on the PixelCNN's shapes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, packing
dev = "cuda:0"
for M, N, K in ((16384, 1024, 1536), (16384, 512, 2560), (16384, 512, 1024), (16384, 256, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
    pl = packing.split_planes(w); out = torch.empty(M, N, device=dev)
    for _ in range(5): ops.linear(x, w, b, out=out, planes=pl)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.linear(x, w, b, out=out, planes=pl)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"M={M} N={N} K={K}: {us:.1f} us, {2.0 * M * N * K / us * 1e-6:.1f} TFLOP/s", flush=True)
ref = (x.double() @ w.double().t() + b.double()).float()
print("max err vs fp64", float((out - ref).abs().max()))
