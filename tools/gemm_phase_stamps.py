"""Diagnostic (GPU box): DVQ_GEMM_CLK=1 makes the bf16x3 DMA GEMM stamp every block (start, loop end, block end) and
print the in-kernel clock (needs the diagnostics build `make -C d-vqvae_amd/csrc diag`, loaded with DVQ_DIAG_LIB=1); DVQ_GEMM_ABL=4/5/6 ablate the activation split / all DMA / the weight DMA pieces (timing only)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DVQ_GEMM_CLK"] = "1"
os.environ["DVQ_DIAG_LIB"] = "1"
import dvqvae_amd
from dvqvae_amd import ops, packing
dev = "cuda:0"
M, N, K = 16384, 1024, 1536
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
pl = packing.split_bf16x3(w); out = torch.empty(M, N, device=dev)
for _ in range(4): ops.linear(x, w, b, out=out, planes=pl)
torch.cuda.synchronize()
