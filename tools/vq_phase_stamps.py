"""Diagnostic (GPU box): per-workgroup phase durations of the VQ fast kernel from its DVQ_VQ_DBG stamps (100 MHz
s_memrealtime): start spread, rows -> registers, chunk loop, epilogue + refine, pair / slow-row counts."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, _lib
os.environ["DVQ_VQ_DBG"] = "1"
dev = "cuda:0"
M = 65536
z = torch.randn(M, 256, device=dev); E = torch.randn(512, 256, device=dev)
pk = ops.vq_pack(E)
for _ in range(3):
    idx = ops.vq_argmin(z, E, packed=pk)
torch.cuda.synchronize()
lib = _lib.load()
nws = lib.dvq_vq_fast_workspace_bytes(M, 512, 256)
ws = ops.workspace(nws, torch.device(dev))
raw = ws[:nws].cpu().numpy()
n_wg = M // 128
full = np.frombuffer(raw.tobytes()[: n_wg * 64], dtype=np.uint64).reshape(n_wg, 8).astype(np.int64)
dbg = full[:, :4]
t0 = dbg[:, 0].min()
rel = (dbg - t0) * 0.01   # us (100 MHz)
print("start  min/med/max us:", rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max())
print("prologue (z in regs) dur med/max:", np.median(rel[:, 1] - rel[:, 0]), (rel[:, 1] - rel[:, 0]).max())
print("chunk loop dur med/max:", np.median(rel[:, 2] - rel[:, 1]), (rel[:, 2] - rel[:, 1]).max())
print("epilogue+refine dur med/max:", np.median(rel[:, 3] - rel[:, 2]), (rel[:, 3] - rel[:, 2]).max())
print("end  min/med/max us:", rel[:, 3].min(), np.median(rel[:, 3]), rel[:, 3].max())

ep = rel[:, 3] - rel[:, 2]
tot, nov = full[:, 4], full[:, 5]
print("pairs per WG mean/max:", tot.mean(), tot.max(), " overflow rows total:", nov.sum(), " WGs with overflow:", (nov > 0).sum())
for lo, hi in [(0, 17), (17, 33), (33, 65), (65, 1000)]:
    m = (tot >= lo) & (tot < hi) & (nov == 0)
    if m.any(): print(f"pairs in [{lo},{hi}) no overflow: n={m.sum()} refine med/max = {np.median(ep[m]):.2f} {ep[m].max():.2f}")
m = nov > 0
if m.any(): print(f"overflow WGs: n={m.sum()} refine med/max = {np.median(ep[m]):.2f} {ep[m].max():.2f}")
