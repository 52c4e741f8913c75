"""Diagnostic (GPU box): per-workgroup phase durations of the VQ streaming kernel from its DVQ_VQ_DBG stamps (100 MHz
s_memrealtime): start spread, prologue (codebook to registers, first tile converted), tile loop, refine tail;
pair / slow-row counts; plus a back-to-back timing of the call and the bit match against the exact kernel."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, _lib
dev = "cuda:0"
M = int(os.environ.get("VQ_M", 65536))
torch.manual_seed(0)
zs = [torch.randn(M, 256, device=dev) for _ in range(6)]
z = zs[0]
E = torch.randn(512, 256, device=dev)
pk = ops.vq_pack(E)
idx = ops.vq_argmin(z, E, packed=pk)
ex = ops.vq_argmin(z, E, fast=False)
torch.cuda.synchronize()
print("bit match vs exact kernel:", float((idx == ex).float().mean()), " mismatches:", int((idx != ex).sum()))
for i in range(6):
    ops.vq_argmin(zs[i], E, packed=pk)
torch.cuda.synchronize()
n = 60
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    ops.vq_argmin(zs[i % 6], E, packed=pk)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
alg = M * 256 * 4 + 512 * 256 * 4 + M * 8
print(f"train of {n} calls: {us:.2f} us per call = {alg / us / 1e3:.1f} GB/s = {alg / us / 1e3 / 8000:.3f} of 8 TB/s")

os.environ["DVQ_VQ_DBG"] = "1"
os.environ["DVQ_DIAG_LIB"] = "1"          # diagnostics build: make -C d-vqvae_amd/csrc diag
for _ in range(3):
    idx = ops.vq_argmin(z, E, packed=pk)
torch.cuda.synchronize()
lib = _lib.load()
nws = lib.dvq_vq_fast_workspace_bytes(M, 512, 256)
ws = ops.workspace(nws, torch.device(dev))
n_wg = min(256, (M + 31) // 32)
REC = 64 + 8 * 32 * 4
raw = ws[: n_wg * REC].cpu().numpy().tobytes()
full = np.stack([np.frombuffer(raw[w * REC: w * REC + 64], dtype=np.uint64) for w in range(n_wg)]).astype(np.int64)
st = np.stack([np.frombuffer(raw[w * REC + 64: (w + 1) * REC], dtype=np.uint32) for w in range(n_wg)]).astype(np.int64).reshape(n_wg, 8, 2, 16)
dbg = full[:, :4]
t0 = dbg[:, 0].min()
rel = (dbg - t0) * 0.01   # us (100 MHz)
print("start  min/med/max us:", rel[:, 0].min(), np.median(rel[:, 0]), rel[:, 0].max())
print("prologue dur med/max:", np.median(rel[:, 1] - rel[:, 0]), (rel[:, 1] - rel[:, 0]).max())
pa, pb = (full[:, 7] >> 32) * 0.01, (full[:, 7] & 0xffffffff) * 0.01
print("   of which: until the rows of tile 0 have arrived med/max:", np.median(pa), pa.max(), " until tile 0 is converted:", np.median(pb), pb.max())
print("tile loop dur med/max:", np.median(rel[:, 2] - rel[:, 1]), (rel[:, 2] - rel[:, 1]).max())
print("refine tail dur med/max:", np.median(rel[:, 3] - rel[:, 2]), (rel[:, 3] - rel[:, 2]).max())
t2b = (full[:, 6] - t0) * 0.01
print("   of which pair chains med/max:", np.median(t2b - rel[:, 2]), (t2b - rel[:, 2]).max(), " slow rows + index write med/max:", np.median(rel[:, 3] - t2b), (rel[:, 3] - t2b).max())
for lo, hi in ((0, 33), (33, 65), (65, 129), (129, 10000)):
    m = (full[:, 4] >= lo) & (full[:, 4] < hi)
    if m.any(): print(f"   WGs with {lo}..{hi - 1} pairs: {m.sum():4d}  chains med/max {np.median((t2b - rel[:, 2])[m]):.2f} / {(t2b - rel[:, 2])[m].max():.2f}")
print("end  min/med/max us:", rel[:, 3].min(), np.median(rel[:, 3]), rel[:, 3].max())
tot, nov = full[:, 4], full[:, 5]
print("pairs per WG mean/max:", tot.mean(), tot.max(), " slow rows total:", nov.sum(), " WGs with slow rows:", (nov > 0).sum())
# in-loop stamps (shader clock, 32-bit) of one tile: 0 start, 1 behind the barrier, 2 after gap 7, 3 after gap 15, 4 after gap 23, 5 end
def d(a, b, tile):
    v = ((st[:, :, tile, b] - st[:, :, tile, a]) & 0xffffffff).reshape(-1)
    return f"{np.median(v):7.0f} /{np.percentile(v, 90):7.0f}"
for tile, name in ((0, "t=2"), (1, "t=5")):
    print(f"-- tile {name}: median / p90 shader cycles:  barrier", d(0, 1, tile), " gaps 0-7", d(1, 2, tile), " gaps 8-15", d(2, 3, tile),
          " gaps 16-23", d(3, 4, tile), " gaps 24-31", d(4, 5, tile), " total", d(0, 5, tile))
