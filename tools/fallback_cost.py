"""Where one out-of-range row of 65 536 costs time (the scenario of test_gen_range_fallback_regenerates_only_the_rows_that_need_it):
per-kernel event times of a clean step and of a step with the row, and the wall time of each."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dvqvae_amd
from dvqvae_amd import synth, _lib
from test_gpu_parity import _gennet
DEV = torch.device("cuda:0")
net, _ = _gennet()
lib = _lib.load()
B = 40
clean = synth.synthetic_clouds(B, 300, seed=91).to(DEV)
Bb, row, j = 65536, 40000, 77
big = clean[torch.arange(Bb, device=DEV) % B].contiguous()
q = torch.empty(Bb, 9, 512, device=DEV).exponential_()
q[:, 1, j] = 1.0e30
E0 = net.vqvae0.vector_quantization.embedding.weight
with torch.no_grad():
    E0[j] = 3.0e6
def run(prof):
    torch.cuda.synchronize()
    if prof:
        lib.dvq_prof_reset(); lib.dvq_prof_enable(1)
    t0 = time.perf_counter()
    net.gen(big, noise=q)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = {}
    if prof:
        lib.dvq_prof_enable(0)
        buf = (_lib.ProfEntry * 64)(); n = lib.dvq_prof_read(buf, 64)
        k = {buf[i].name.decode(): (buf[i].ms, buf[i].count) for i in range(min(n, 64))}
        lib.dvq_prof_reset()
    return dt, k
run(False); run(False)
tc = min(run(False)[0] for _ in range(3)); _, kc = run(True)
q[row, 1, j] = 1.0e-30
run(False); run(False)
tb = min(run(False)[0] for _ in range(3)); _, kb = run(True)
print(f"clean {tc*1e3:.1f} ms, one bad row {tb*1e3:.1f} ms, fallback rows {net.range_fallback_rows}")
for name in sorted(set(kc) | set(kb), key=lambda n: -(kb.get(n, (0, 0))[0] - kc.get(n, (0, 0))[0])):
    a, b = kc.get(name, (0.0, 0)), kb.get(name, (0.0, 0))
    if abs(b[0] - a[0]) > 0.05 or a[1] != b[1]:
        print(f"  {name:28s} clean {a[0]:8.2f} ms / {a[1]:4d} calls   bad {b[0]:8.2f} ms / {b[1]:4d} calls   diff {b[0]-a[0]:+.2f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); net.gen(big, noise=q); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
