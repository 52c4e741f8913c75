"""Where one small-batch GenNet.gen call spends its host time (GPU box): wall time per call, cProfile of the host side,
launch count from the library's own per-launch profiling."""
import cProfile, pstats, sys, time, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvqvae_amd
from dvqvae_amd import _lib, mano as dmano, synth
from dvqvae_amd.network.gen_net import GenNet
dev = torch.device("cuda:0")
K = 512
net = GenNet(n_embeddings=K, prior_tokens=K, prior_classes=K)
sd = synth.synthetic_state_dict(net.state_dict(), 1234)
net.load_state_dict(sd); net.eval().to(dev)
net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(dev))
lib = _lib.load()
for B in (1, 8, 100):
    obj = synth.synthetic_clouds(B, 1024, seed=1).to(dev)
    for _ in range(3): net.gen(obj)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); net.gen(obj); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    ts.sort(key=lambda x: x[1])
    print(f"B={B}: host-side call {ts[3][0]:.2f} ms, with sync {ts[3][1]:.2f} ms")
    if B == 1:
        lib.dvq_prof_reset(); lib.dvq_prof_enable(1); net.gen(obj); torch.cuda.synchronize(); lib.dvq_prof_enable(0)
        buf = (_lib.ProfEntry * 64)(); n = lib.dvq_prof_read(buf, 64)
        tot = sum(buf[i].count for i in range(n)); ms = sum(buf[i].ms for i in range(n))
        print(f"  launches bracketed: {tot}, summed kernel time {ms:.2f} ms")
        for i in sorted(range(n), key=lambda i: -buf[i].ms)[:12]:
            print(f"    {buf[i].name.decode():20s} {buf[i].count:5d} launches {buf[i].ms:8.3f} ms")
        pr = cProfile.Profile(); pr.enable(); net.gen(obj); torch.cuda.synchronize(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
