import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, packing
dev = "cuda:0"
M, N, K = 16384, 1024, 1536
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; b = torch.randn(N, device=dev)
pl = packing.split_bf16x3(w)
out = torch.empty(M, N, device=dev)
def run(planes, n=20):
    for _ in range(3): ops.linear(x, w, b, out=out, planes=planes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): ops.linear(x, w, b, out=out, planes=planes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return ms, 2.0 * M * N * K / ms / 1e9
print(os.environ.get("DVQ_GEMM", "bf16x3"), os.environ.get("DVQ_GEMM_ABL", "0"), "on-the-fly W: %.3f ms %.1f TF" % run(None), "| planes+DMA: %.3f ms %.1f TF" % run(pl))
