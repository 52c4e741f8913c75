"""GPU box, diagnostics build (make -C d-vqvae_amd/csrc diag): vq_pipe.hip's per-workgroup phase stamps (100 MHz) and the
timing-only ablations of its loop (results INVALID under an ablation: only the time is read)."""
import os, sys
os.environ["DVQ_DIAG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dvqvae_amd
from dvqvae_amd import ops, _lib
dev = "cuda:0"
lib = _lib.load()
M, D, K = 65536, 256, 512
zs = [torch.randn(M, D, device=dev) for _ in range(6)]
E = torch.randn(K, D, device=dev)
pk = ops.vq_pack(E)


def setenv(**kw):
    for k in ("DVQ_VQP_ABL", "DVQ_VQP_VAR", "DVQ_VQP_DBG"):
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in kw.items()})
    lib.dvq_reload_env()


def timeit(n=30):
    for i in range(6): ops.vq_argmin(zs[i], E, packed=pk)
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): ops.vq_argmin(zs[i % 6], E, packed=pk)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(out)[1]


def stamps():
    """median per-workgroup phase durations (us) of the last launch: tables + prologue, tile loop, last merge + expand, refine + store"""
    for i in range(12): ops.vq_argmin(zs[i % 6], E, packed=pk)
    torch.cuda.synchronize()
    ws = ops.workspace(0, torch.device(dev))
    st = ws[: 256 * 64].view(torch.int64).view(256, 8).cpu().numpy().astype(np.float64)
    d = [np.median(st[:, i + 1] - st[:, i]) * 0.01 for i in range(4)]
    span = (st[:, 4].max() - st[:, 0].min()) * 0.01
    return d, span, st


os.environ["DVQ_VQ_KERNEL"] = "17"
ONLY = os.environ.get("VQP_DIAG_ONLY")            # "var": the structure variants only
if ONLY == "var":
    VARIANTS_FILTER = lambda env: "DVQ_VQP_VAR" in env
else:
    VARIANTS_FILTER = lambda env: True
VARIANTS = ([({"DVQ_VQP_VAR": v}, f"VAR {v}: " + w) for v, w in (
    (0, "as generated"), (1, "deferred scores of waves 8-15"), (5, "1 + static priority, younger first"),
    (16, "priority by progress"), (17, "16 + 1"), (18, "16 + vector work in the first ten gaps"), (19, "16 + 1 + 2"))]
    + [({"DVQ_VQP_ABL": a}, f"ABL {a:3d} {w}") for a, w in (
        (32, "no MFMA"), (96, "no MFMA, no fragment reads"), (4, "no conversion"), (8, "no scoring"), (2, "no merge"), (1, "no row loads"),
        (15, "no loads/merge/conversion/scoring"), (256, "row loads without nt"), (512, "row loads from tile 0 only (L2)"))])
def phase_table():
    setenv(DVQ_VQP_DBG=1)
    d, span, st = stamps()
    t0 = st[:, 0].min()
    ph = {"start skew": st[:, 0] - t0, "tables + prologue": st[:, 1] - st[:, 0], "tile loop": st[:, 2] - st[:, 1],
          "last merge + expand": st[:, 3] - st[:, 2], "refine + store": st[:, 4] - st[:, 3], "whole workgroup": st[:, 4] - st[:, 0]}
    for k, v in ph.items():
        v = v * 0.01
        print(f"{k:20s} median {np.median(v):6.2f} us   p10 {np.percentile(v, 10):6.2f}   p90 {np.percentile(v, 90):6.2f}   max {v.max():6.2f}")
    print("kernel span (first start .. last end): %.2f us" % span)
    print("pairs per workgroup: mean %.1f max %d; all-entries rows: %d" % (st[:, 5].mean(), st[:, 5].max(), st[:, 6].sum()))
    print("%-46s %8s | %8s %8s %8s %8s %8s" % ("variant", "call us", "prologue", "loop", "merge", "refine", "span"), flush=True)


ONE = os.environ.get("VQP_DIAG_ONE")              # child: one row of the table ("index")
if ONE is not None:
    env, what = VARIANTS[int(ONE)]
    setenv(**env)
    t = timeit()
    setenv(DVQ_VQP_DBG=1, **env)
    d, span, _ = stamps()
    print("%-46s %8.2f | %8.2f %8.2f %8.2f %8.2f %8.2f" % (what, t, d[0], d[1], d[2], d[3], span), flush=True)
    sys.exit(0)
import subprocess
phase_table()
for i, (env, what) in enumerate(VARIANTS):
    if not VARIANTS_FILTER(env): continue
    # every row in a process of its own: an ablation computes with garbage (its results are invalid by design) and may fault
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, VQP_DIAG_ONE=str(i)), capture_output=True, text=True, timeout=300)
    rows = [ln for ln in r.stdout.splitlines() if " | " in ln and not ln.startswith("variant")]
    print(rows[-1] if r.returncode == 0 and rows else "%-46s FAILED (rc %d): %s" % (what, r.returncode, (r.stderr.strip().splitlines() or ["?"])[-1][:120]), flush=True)
os.environ["DVQ_VQ_KERNEL"] = "16"
setenv()
print("sixteen-wave kernel: %.2f us" % timeit())
