"""GPU box: time of one PointNet encoder call (STN trunk + encoder trunk) at N = 778 and N = 1024 with the tail tile on / off
(DVQ_PN_TAIL): `python tools/tail_time.py`; under rocprofv3 --kernel-trace --stats it gives the per-kernel split."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import dvqvae_amd
from dvqvae_amd import _lib
from test_gpu_parity import _pointnet, synth, gpu
net, _ = _pointnet(3, 7)
for N in (778, 1024):
    x = gpu(synth.synthetic_clouds(7282, N, seed=1, channels=3))
    for tail in ("1", "0"):
        os.environ["DVQ_PN_TAIL"] = tail; _lib.load().dvq_reload_env()
        for _ in range(2): net(x)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(5): net(x)
        torch.cuda.synchronize(); print("N", N, "tail", tail, "ms per call", (time.time() - t0) / 5 * 1e3, flush=True)
