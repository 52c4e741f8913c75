import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd
from dvqvae_amd import ops, _lib
os.environ["DVQ_VQ_DBG"] = "1"
dev = "cuda:0"; M = 65536
lib = _lib.load()
nws = lib.dvq_vq_fast_workspace_bytes(M, 512, 256)
z = torch.randn(M, 256, device=dev); E = torch.randn(512, 256, device=dev)
pk = ops.vq_pack(E)
for it in range(3):
    idx = ops.vq_argmin(z, E, packed=pk)
    torch.cuda.synchronize()
    ws = ops.workspace(nws, torch.device(dev))
    n_wg = M // 128
    full = np.frombuffer(ws[:n_wg * 64].cpu().numpy().tobytes(), dtype=np.uint64).reshape(n_wg, 8).astype(np.int64)
    hw, xcc = full[:, 6], full[:, 7] & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    print("distinct CUs:", len(set(key.tolist())), " xcc of first 16 WGs:", xcc[:16].tolist())
    print("key of WG 0..7:", key[:8].tolist(), " WG 256..263:", key[256:264].tolist())
    from collections import defaultdict
    d = defaultdict(list)
    for i, k in enumerate(key.tolist()): d[k].append(i)
    diffs = [v[1] - v[0] for v in d.values() if len(v) == 2]
    print("co-resident pairs:", len(diffs), " index difference histogram:", np.unique(diffs, return_counts=True))
    print("sizes:", np.unique([len(v) for v in d.values()], return_counts=True))
