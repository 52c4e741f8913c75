"""GPU box, diagnostics build: per-wave timeline of periods 3 and 4 of vq_pipe.hip's tile loop (shader-clock cycles after the
first wave left the period's barrier; median over the workgroups)."""
import os, sys
os.environ["DVQ_DIAG_LIB"] = "1"; os.environ["DVQ_VQ_KERNEL"] = "17"; os.environ["DVQ_VQP_DBG"] = "1"; os.environ["DVQ_VQP_VAR"] = str(8 | int(sys.argv[1]) if len(sys.argv) > 1 else 8)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dvqvae_amd
from dvqvae_amd import ops
dev = "cuda:0"
M, D, K = 65536, 256, 512
zs = [torch.randn(M, D, device=dev) for _ in range(6)]
E = torch.randn(K, D, device=dev)
pk = ops.vq_pack(E)
for i in range(12): idx = ops.vq_argmin(zs[i % 6], E, packed=pk)
torch.cuda.synchronize()
ws = ops.workspace(0, torch.device(dev))
raw = ws[256 * 64: 256 * 64 + 256 * 1024].view(torch.int32).view(256, 16, 2, 8).cpu().numpy().astype(np.int64) & 0xffffffff
assert torch.equal(idx, ops.vq_argmin(zs[5], E, fast=False)), "stamped kernel must still be exact"   # (overwrites the workspace)
print("structure variant", int(os.environ["DVQ_VQP_VAR"]) & 7)
for tl in range(2):
    st = raw[:, :, tl, :]                                     # [wg, wave, 8]: 0 barrier left, 2 matrix phase (+ conversion, merge) done,
    base = st[:, :, 0].min(axis=1, keepdims=True)[:, :, None]  #                3 scores stored, 4 record done (= barrier reached)
    rel = (st - base) & 0xffffffff
    med = np.median(rel, axis=0)
    nxt = np.median(((raw[:, :, 1, 0].min(axis=1) - raw[:, :, 0, 0].min(axis=1)) & 0xffffffff)) if tl == 0 else float("nan")
    print(f"period {3 + tl}: cycles after the first wave left the barrier (next period's barrier: {nxt:.0f})")
    print("wave    bar  deferred S  P+C+M done  scores done  record done")
    for w in range(16):
        m = med[w]
        print(f"{w:4d} {m[0]:6.0f} {m[1]:11.0f} {m[2]:11.0f} {m[3]:12.0f} {m[4]:12.0f}")
