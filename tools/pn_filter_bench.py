"""Filtered vs six-product PointNet trunk: time per encode, candidate statistics (DVQ_PN_STATS), worst deviation."""
import sys, time, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import dvqvae_amd
from dvqvae_amd import synth, ops, _lib
from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
from util import load_synth
dev = torch.device("cuda:0")
C = int(os.environ.get("PN_C", "4")); N = int(os.environ.get("PN_N", "1024")); B = int(os.environ.get("PN_B", "8192"))
net = PointNetEncoder(channel=C); load_synth(net, 3); net = net.to(dev)
x = synth.synthetic_clouds(B, N, seed=1, channels=C).to(dev)
res = {}
for mode in ("1", "0"):
    os.environ["DVQ_PN_FILTER"] = mode
    dvqvae_amd._lib.load().dvq_reload_env()          # the library reads its knobs once
    for _ in range(2): net(x)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): f, tr, _ = net(x)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    res[mode] = (f, tr)
    print("filter" if mode == "1" else "six-product", "encode C=%d B=%d N=%d: %.2f ms" % (C, B, N, dt * 1e3), flush=True)
print("max |feat diff|", float((res["1"][0] - res["0"][0]).abs().max()), "max |trans diff|", float((res["1"][1] - res["0"][1]).abs().max()))
os.environ["DVQ_PN_FILTER"] = "1"; os.environ["DVQ_PN_STATS"] = "1"
dvqvae_amd._lib.load().dvq_reload_env()
net(x[: min(B, 1024)])
