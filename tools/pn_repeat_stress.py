"""Repeatability stress of the PointNet trunks and of GenNet.gen at the benchmark size (how the round-3 packed-fp32 fault was
found: csrc/Makefile).  Every call of 65 536 clouds (2 048 distinct ones, 32 copies each) is compared with the
exhaustive evaluation (DVQ_PN_EXHAUSTIVE=1) bit for bit; gen() calls with each other.
    gpurun -- python tools/pn_repeat_stress.py [encoder calls per encoder, default 150] [gen calls, default 40]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T                                  # noqa: E402  (the GenNet fixture of the parity tests)
from dvqvae_amd import _lib, synth                           # noqa: E402

net, _ = T._gennet()
blk, reps, N = 2048, 32, 1024
obj = synth.synthetic_clouds(blk, N, seed=91).to("cuda:0")
q = synth.exp1_noise(blk, 9, 512, seed=92).to("cuda:0")
big_obj, big_q = obj.repeat(reps, 1, 1), q.repeat(reps, 1, 1)
lib = _lib.load()
mods = {"pos": net.obj_encoder_pos, "type": net.obj_encoder_type}
os.environ["DVQ_PN_EXHAUSTIVE"] = "1"
lib.dvq_reload_env()
ref = {k: m(big_obj)[0] for k, m in mods.items()}
del os.environ["DVQ_PN_EXHAUSTIVE"]
lib.dvq_reload_env()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad_enc, t0 = 0, time.time()
for it in range(iters):
    for k, m in mods.items():
        f = m(big_obj)[0]
        d = f != ref[k]
        if bool(d.any()):
            bad_enc += 1
            for r in d.any(1).nonzero().flatten().tolist()[:3]:
                ch = d[r].nonzero().flatten().tolist()
                print(f"call {it} {k}: cloud {r} (= block cloud {r % blk}) channels {ch[:8]} got {f[r, ch[:3]].tolist()} want {ref[k][r, ch[:3]].tolist()}", flush=True)
print(f"encoders: {bad_enc} bad calls of {2 * iters} ({time.time() - t0:.1f} s)")
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 40
r0, p0, a0 = net.gen(big_obj, noise=big_q, return_aux=True)
keep = {k: a0[k].clone() for k in ("feat_pos", "feat_type", "hand_feat", "codes", "idx6")}
r0, p0 = r0.clone(), p0.clone()
bad_gen = 0
for it in range(gens):
    r, p, a = net.gen(big_obj, noise=big_q, return_aux=True)
    diff = [k for k in keep if not torch.equal(a[k], keep[k])] + ([] if torch.equal(r, r0) and torch.equal(p, p0) else ["recon/recon_pos"])
    if diff:
        bad_gen += 1
        print(f"gen call {it}: {diff} differ from the first call", flush=True)
print(f"gen: {bad_gen} bad calls of {gens}")
from dvqvae_amd import ops                                   # noqa: E402
faults = ops.pointnet_fault_counters()
print(f"run-time consistency counters of the filtered trunk (suspect records, channels outside their interval): {faults}")
sys.exit(1 if (bad_enc or bad_gen or faults != (0, 0)) else 0)
