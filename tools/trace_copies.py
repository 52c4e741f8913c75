#!/usr/bin/env python3
"""Where do the copyBuffer / fillBuffer launches of one 65 536-grasp step come from?

rocprofv3's kernel stats of the default bench count ~425 __amd_rocclr_copyBuffer and ~100 __amd_rocclr_fillBufferAligned launches
per step (profiles/r05_v6_bench_stats_rocprofv3_kernel_stats.csv).  The library itself issues one hipMemsetAsync per trunk launch;
the copies are torch's (Tensor.copy_ of a contiguous same-dtype tensor is a device-to-device hipMemcpyAsync).  This runs bench.py's
step under torch.profiler with Python stacks and prints, per calling line, the number of device memcpys / memsets and their time.

    python tools/trace_copies.py [batch] > gpurun_out/trace_copies.txt
"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import dvqvae_amd
from dvqvae_amd import _lib, dist, mano as dmano, ops, synth
from dvqvae_amd.network.gen_net import GenNet
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N, K = 1024, 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
lib = _lib.load()
net = GenNet(n_embeddings=K, prior_tokens=K, prior_classes=K)
sd = synth.synthetic_state_dict(net.state_dict(), 1234)
net.load_state_dict(sd)
net.eval().to(dev)
sd = synth.diversify_object_codebook(net, sd, N)
net.set_noise_seed(20261003)
net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(dev))
pool = synth.synthetic_clouds(min(B, 4096), N, seed=1000).to(dev)
rows = torch.arange(0, B, device=dev)
obj = pool[rows % pool.shape[0]].contiguous()


def step(i):
    recon, pos = net.gen(obj, seed=20261003, row0=0, stream_id=i)
    return dist.all_gather_rows(ops.assemble61(recon, pos), total_rows=B, verify=False)


step(0); step(1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2)
    torch.cuda.synchronize()

ev = prof.events()
# device-side activity names of the runtime's copies and fills
dev_ev = [e for e in ev if e.device_type == torch.autograd.DeviceType.CUDA]
names = collections.Counter(e.name[:60] for e in dev_ev)
print("device activities of one step (name, count):")
for n, c in names.most_common(25):
    print(f"  {c:6d}  {n}")

# CPU-side ops that own a memcpy / memset: walk the op events, attribute by innermost repo frame
def frame_of(e):
    for fr in (e.stack or []):
        if "/d-vqvae_amd/" in fr or "/dvqvae_amd/" in fr or "bench.py" in fr or "trace_copies" in fr:
            return fr.strip()
    return (e.stack[0].strip() if e.stack else "?")


by = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    ks = [k for k in (e.kernels or []) if "Memcpy" in k.name or "Memset" in k.name or "copyBuffer" in k.name or "fillBuffer" in k.name]
    if not ks or e.cpu_children and any(c.kernels for c in e.cpu_children):
        continue
    rec = by[frame_of(e)]
    rec[0] += len(ks); rec[1] += sum(k.duration for k in ks); rec[2][e.name] += len(ks)
print("\nmemcpy / memset launches by calling line (count, device us, ops):")
for fr, (n, us, ops_) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n:5d} {us:9.1f}  {fr}   {dict(ops_)}")
