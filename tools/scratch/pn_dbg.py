import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import dvqvae_amd
from dvqvae_amd import synth
from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
from util import load_synth
dev = torch.device("cuda:0")
def run(C, N, B, seed):
    net = PointNetEncoder(channel=C); load_synth(net, seed); net = net.to(dev)
    x = synth.synthetic_clouds(B, N, seed=300 + N, channels=C).to(dev)
    os.environ["DVQ_PN_FILTER"] = "1"; f1, t1, _ = net(x)
    os.environ["DVQ_PN_FILTER"] = "0"; f0, t0, _ = net(x)
    d = (f1 - f0).abs().amax(dim=1); dt = (t1 - t0).abs().amax(dim=(1, 2))
    print(f"C={C} N={N} B={B} seed={seed}: feat diff per sample", [f"{v:.1e}" for v in d.tolist()[:8]], "trans", [f"{v:.1e}" for v in dt.tolist()[:8]], flush=True)
for cfg in [(3, 778, 4, 20261003 + 3), (3, 778, 5, 20261003 + 30), (3, 778, 4, 20261003 + 30), (3, 1024, 5, 33), (4, 778, 5, 33), (3, 768, 5, 33), (3, 778, 5, 33), (3, 100, 5, 33), (4, 100, 5, 33)]:
    run(*cfg)
