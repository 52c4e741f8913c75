"""GPU box: the round-6 VQ kernel (DVQ_VQ_KERNEL=17, vq_pipe.hip) against the exact kernel on growing problems, then timing against
the sixteen-wave kernel.  Prints, never asserts (a first-contact tool); every stage in its own process would be safer against hangs:
run under `timeout`.   usage: vq_pipe_check.py [stage ...]   stages: small mid full adv time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvqvae_amd
from dvqvae_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()
stages = sys.argv[1:] or ["small", "mid", "full", "adv", "time"]


def use(k):
    os.environ["DVQ_VQ_KERNEL"] = str(k); lib.dvq_reload_env()


def cmp(z, E, what, pk=None):
    use(17)
    a = ops.vq_argmin(z, E, packed=pk, fast=True)
    torch.cuda.synchronize()
    b = ops.vq_argmin(z, E, fast=False)
    bad = (a != b).nonzero().flatten()
    print(f"{what}: M={z.shape[0]} mismatches={bad.numel()}" + (f" first rows {bad[:8].tolist()} got {a[bad[:8]].tolist()} want {b[bad[:8]].tolist()}" if bad.numel() else ""), flush=True)
    return bad.numel()


torch.manual_seed(3)
E = torch.randn(512, 256, device=dev)
pk = ops.vq_pack(E)
if "small" in stages:
    for M in (32, 1, 31, 33, 64, 100):
        cmp(torch.randn(M, 256, device=dev), E, "small", pk)
if "mid" in stages:
    for M in (255, 256, 257, 1000, 4096, 8192, 8192 + 5):
        cmp(torch.randn(M, 256, device=dev), E, "mid", pk)
if "full" in stages:
    for M in (65536, 70001, 200000):
        z = torch.randn(M, 256, device=dev)
        cmp(z, E, "full", pk)
        use(17)
        a = ops.vq_argmin(z, E, packed=pk)
        for _ in range(3):
            assert torch.equal(ops.vq_argmin(z, E, packed=pk), a), "not repeatable"
if "adv" in stages:
    z = torch.randn(300, 256, device=dev)
    z[0] = E[7]; z[1] = 0.5 * (E[3] + E[9]); z[2, 5] = float("nan"); z[3, 9] = float("inf"); z[4] = 0.0
    z[5] = 7.0e4; z[6] = 1e-6 * z[6]; z[10:40] = E[100:130] + 1e-4 * torch.randn(30, 256, device=dev)
    cmp(z, E, "adversarial", pk)
    Et = (torch.rand(512, 256, device=dev) * 2 - 1) / 512
    cmp(torch.randn(1000, 256, device=dev), Et, "tie-prone codebook")
    E2 = E.clone(); E2[100] = E2[7]; E2[300] = E2[7]
    for k in range(400, 412): E2[k] = E2[399]
    z2 = torch.randn(256, 256, device=dev); z2[0] = E2[7]; z2[2] = E2[405]
    cmp(z2, E2, "duplicated entries")
    for sz, se in ((100.0, 0.01), (1e-3, 1e3), (30.0, 30.0)):
        cmp(torch.randn(2048, 256, device=dev) * sz, (torch.rand(512, 256, device=dev) * 2 - 1) * se, f"scales {sz},{se}")
if "time" in stages:
    M = 65536
    zs = [torch.randn(M, 256, device=dev) for _ in range(6)]
    res = {}
    for rnd in range(5):
        for kern in ("17", "16"):
            use(kern)
            for i in range(6): idx = ops.vq_argmin(zs[i], E, packed=pk)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30): idx = ops.vq_argmin(zs[i % 6], E, packed=pk)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(kern, []).append(e0.elapsed_time(e1) * 1e3 / 30)
    for k, v in res.items():
        v = sorted(v)
        print(f"DVQ_VQ_KERNEL={k}: median {v[len(v) // 2]:.2f} us, min {v[0]:.2f} us per call ({[round(x, 2) for x in v]})", flush=True)
