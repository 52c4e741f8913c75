"""PointNet trunks alone: per-kernel time of one encoder pass at the benchmark's launch shape (prof events), object and hand clouds.
PN_B samples (default 7282 = one launch of the 65 536 batch), repeated PN_REP times."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dvqvae_amd
from dvqvae_amd import synth, _lib
from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
from util import load_synth
dev = torch.device("cuda:0")
B = int(os.environ.get("PN_B", "7282")); rep = int(os.environ.get("PN_REP", "3"))
lib = _lib.load()
for C, N in ((4, 1024), (3, 778)):
    net = PointNetEncoder(channel=C); load_synth(net, 3); net = net.eval().to(dev)
    x = synth.synthetic_clouds(B, N, seed=1, channels=C).to(dev)
    for _ in range(2): net(x)
    torch.cuda.synchronize()
    lib.dvq_prof_reset(); lib.dvq_prof_enable(1)
    t = time.time()
    for _ in range(rep): net(x)
    torch.cuda.synchronize(); dt = (time.time() - t) / rep
    lib.dvq_prof_enable(0)
    buf = (_lib.ProfEntry * 64)(); n = lib.dvq_prof_read(buf, 64)
    k = {buf[i].name.decode(): buf[i].ms / rep for i in range(min(n, 64))}
    lib.dvq_prof_reset()
    pn = sum(v for kk, v in k.items() if kk.startswith("pn_"))
    print(f"C={C} N={N} B={B}: encode {dt*1e3:.2f} ms; " + ", ".join(f"{kk} {v:.3f}" for kk, v in sorted(k.items(), key=lambda kv: -kv[1])[:6]) + f"; pn total {pn:.3f} ms", flush=True)
