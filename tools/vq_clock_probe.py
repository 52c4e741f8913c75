"""GPU box: does the VQ kernel's time depend on how long the chip has been busy?  The microbench times a 30-call train (1 ms of work)
after an idle gap; inside the 65 536-grasp step the same kernel runs between seconds of PointNet and GEMM kernels.  Times one call
(HIP-event pair around a train) for trains of 30 ... 10 000 calls, back to back and after a matrix-core burn, for kernels 16 and 17."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvqvae_amd
from dvqvae_amd import ops, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
M, D, K = 65536, 256, 512
zs = [torch.randn(M, D, device=dev) for _ in range(6)]
E = torch.randn(K, D, device=dev)
pk = ops.vq_pack(E)
A = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def train(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ops.vq_argmin(zs[i % 6], E, packed=pk)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def burn(ms):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            torch.matmul(A, A)
        torch.cuda.synchronize()


for kern in ("16", "17"):
    os.environ["DVQ_VQ_KERNEL"] = kern
    lib.dvq_reload_env()
    for i in range(6):
        ops.vq_argmin(zs[i], E, packed=pk)
    torch.cuda.synchronize()
    time.sleep(0.5)
    print(f"kernel {kern}: after 0.5 s idle, trains of n calls (us per call): " + ", ".join(f"n={n}: {train(n):.2f}" for n in (30, 30, 100, 300, 1000, 3000, 10000, 30, 300)), flush=True)
    time.sleep(0.5)
    burn(500)
    print(f"kernel {kern}: right after a 500 ms bf16 GEMM burn: " + ", ".join(f"n={n}: {train(n):.2f}" for n in (30, 30, 300, 3000)), flush=True)
    # calls spaced out (a gap of idle time between single calls): what a lone call inside an otherwise idle process costs
    ts = []
    for i in range(20):
        time.sleep(0.01)
        ts.append(train(1))
    print(f"kernel {kern}: single calls 10 ms apart: median {sorted(ts)[10]:.2f} us (event pair around ONE call: includes its launch)", flush=True)
