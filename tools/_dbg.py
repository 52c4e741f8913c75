import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, _lib
os.environ["DVQ_VQ_ABL"] = "9"
dev = "cuda:0"
z = torch.randn(65536, 256, device=dev); E = torch.randn(512, 256, device=dev)
pk = ops.vq_pack(E)
for _ in range(3):
    idx = ops.vq_argmin(z, E, packed=pk)
torch.cuda.synchronize()
lib = _lib.load()
nws = lib.dvq_vq_fast_workspace_bytes(65536, 512, 256)
ws = ops.workspace(nws, torch.device(dev))
raw = ws[:nws].cpu().numpy()
dbg = np.frombuffer(raw.tobytes()[-1536:], dtype=np.uint64).reshape(2, 8, 12)
names = ["prologue", "top wait+barrier", "issue glds", "prepass", "barrier", "MFMA loop", "score+min", "barrier", "thr+scan", "barrier", "finalize", "-"]
for b in range(2):
    print("block", b)
    for w in (0, 3, 7):
        print(" wave", w, [int(v) for v in dbg[b, w]], "sum", int(dbg[b, w].sum()))

print(names)
import time
os.environ["DVQ_VQ_ABL"] = "0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (1, 10, 50):
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        idx = ops.vq_argmin(z, E, packed=pk)
    e1.record(); torch.cuda.synchronize()
    print(n, "calls:", e0.elapsed_time(e1) * 1e3 / n, "us per call")
def r256(x): return (x + 255) // 256 * 256
M = 65536
off_cnt = r256(M * 16)
off_ambc = off_cnt + r256(M)
raw = ws[:nws].cpu().numpy()
ambc = np.frombuffer(raw[off_ambc:off_ambc + 4096].tobytes(), dtype=np.int32)[:256]
cntb = raw[off_cnt:off_cnt + M]
print("ambiguous rows:", int(ambc.sum()), "of", M, "=", ambc.sum() / M, " per-WG min/max", ambc.min(), ambc.max())

