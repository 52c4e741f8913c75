"""Small-M GEMM microbench (GPU box): one launch on warm weights (same panel every call: L2 / Infinity Cache resident) and on
rotating panels (each call streams 48 distinct panels: HBM), skinny kernel vs the tiled kernels (DVQ_GEMM_SKINNY=0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, packing, _lib
dev = "cuda:0"
lib = _lib.load()

def timed(fn, n):
    for _ in range(3): fn(0)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

for M in (1, 8, 100, 256):
    for N, K in ((1024, 2048), (512, 512), (1024, 512)):
        x = torch.randn(M, K, device=dev); b = torch.randn(N, device=dev)
        ws = [torch.randn(N, K, device=dev) * 0.03 for _ in range(48)]
        pls = [packing.split_bf16x3(w) for w in ws]
        out = torch.empty(M, N, device=dev)
        res = {}
        for mode in ("1", "2", "0"):
            os.environ["DVQ_GEMM_SKINNY"] = mode; lib.dvq_reload_env()
            res[mode] = (timed(lambda i: ops.linear(x, ws[0], b, out=out, planes=pls[0]), 96),
                         timed(lambda i: ops.linear(x, ws[i % 48], b, out=out, planes=pls[i % 48]), 96))
        print(f"M={M:4d} N={N} K={K}: skinny (LDS-staged) warm {res['1'][0]:6.1f} us, rotating {res['1'][1]:6.1f} us | register-staged warm {res['2'][0]:6.1f} us, rotating {res['2'][1]:6.1f} us | tiled warm {res['0'][0]:6.1f} us, rotating {res['0'][1]:6.1f} us", flush=True)
