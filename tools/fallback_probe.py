"""Where the per-row range fallback's time goes: one B = 1 gen() on the six-product images, flips of the image kind."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dvqvae_amd
from dvqvae_amd import synth, _lib, packing, ops, mano as dmano
from dvqvae_amd.network.gen_net import GenNet
dev = torch.device("cuda:0")
net = GenNet(n_embeddings=512, prior_tokens=512, prior_classes=512)
net.load_state_dict(synth.synthetic_state_dict(net.state_dict(), 1234)); net.eval().to(dev)
net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(dev))
obj = synth.synthetic_clouds(64, 1024, seed=3).to(dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("B=1 f16x2 gen: %.2f ms" % t(lambda: net.gen(obj[:1], seed=1)))
def flip():
    with packing.gemm_kind_as(_lib.PLANES_BF16X3):
        net.gen(obj[:1], seed=1)
print("B=1 bf16x3 gen incl. both flips: %.2f ms" % t(flip))
with packing.gemm_kind_as(_lib.PLANES_BF16X3):
    print("B=1 bf16x3 gen, no flip: %.2f ms" % t(lambda: net.gen(obj[:1], seed=1)))
    with ops.no_range_check():
        print("B=1 bf16x3 _gen_impl only: %.2f ms" % t(lambda: net._gen_impl(obj[:1], None, net._noise_key(1, 0, 0))))
print("B=64 f16x2 gen after flips: %.2f ms" % t(lambda: net.gen(obj, seed=1)))
# one bad row in a big batch: where do the extra milliseconds go?
B = 16384
big = obj[torch.arange(B, device=dev) % 64].contiguous()
print("B=%d clean: %.2f ms" % (B, t(lambda: net.gen(big, seed=2), 3)))
bad = big.clone(); bad[5000] *= 1e6
print("B=%d one bad row: %.2f ms (fallback rows so far %d)" % (B, t(lambda: net.gen(bad, seed=2), 3), net.range_fallback_rows))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); net.gen(bad, seed=2); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
