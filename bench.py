#!/usr/bin/env python3
"""Headline benchmark: grasps/s of the batched grasp-generation path (GenNet.gen) on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W`` (N>1 is launched through
``python -m torch.distributed.run``); one JSON line on rank 0.

Workload: ``--batch`` (default 65536 = BASELINE.json's batch) grasps per step from synthetic N=1024-point clouds with
K=512 codebooks; a step is one full pass of the hot path over the batch (device Philox noise for the prior's draws ->
PointNet x2 -> VQ argmin -> cached PixelCNN sampling -> decoder -> MANO -> PointNet -> pos decoder -> 61-parameter
assembly) followed by the all-gather of the [B,61] parameters over RCCL.  ``--scaling strong`` (default, what the metric
string describes): the batch is the GLOBAL batch, rank r generates rows shard_range(B, r, R) -- B/R grasps, 2.0 MB per
rank into the all-gather at B=65536, R=8; ``--scaling weak``: ``--batch`` grasps per rank.  Inputs and weights are
resident in HBM before the timed region; the noise is generated inside the step, keyed by the global row.
The headline is timed with the library's per-launch profiling OFF; the kernel breakdown comes from a second, identical
pass with every launch bracketed by HIP events on the launch stream.

Extra objects in the JSON line:
  roofline            the matrix-core kernel with the largest share of the step (GEMMs AND the PointNet trunk are candidates),
                      algorithmic FLOPs / HIP-event time against the roof of its arithmetic; ``traffic`` from the committed PMC passes
                      next to ``algorithmic_bytes``; the dominant GEMM and the PointNet group beside it
  (--config 3 / 4)    BASELINE configs 3 and 4 as lines of their own, each against SURVEY 8(d)'s bound
  roofline_vq_argmin  BASELINE config 2 (VectorQuantizer K=512 D=256 argmin-only, M=65536) against HBM peak
  cpu_baseline        the CPU oracle (a port of the reference) timed on this box's host cores, bounded sample
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak
BF16_MFMA_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA
# The GEMMs run "f16x2" (default): every fp32 product costs THREE fp16 MFMA products (two fp16 pieces per operand, the second
# scaled by 2^11; two accumulators), so the roof of ALGORITHMIC fp32 FLOP/s is the dense fp16 peak / 3.  DVQ_GEMM=bf16x3: the
# exact three-piece bf16 split, six products (peak / 6).  DVQ_GEMM=fp32: v_mfma_f32_32x32x2_f32.
_gm = os.environ.get("DVQ_GEMM", "").strip().lower()
GEMM_MODE = "fp32" if _gm == "fp32" else ("bf16x3" if _gm == "bf16x3" else "f16x2")
GEMM_PEAK_TF = {"fp32": FP32_MFMA_PEAK_TF, "bf16x3": BF16_MFMA_PEAK_TF / 6.0, "f16x2": BF16_MFMA_PEAK_TF / 3.0}[GEMM_MODE]
GEMM_PEAK_NOTE = {"fp32": "fp32 MFMA (= fp32 vector) peak",
                  "bf16x3": "dense bf16 MFMA 2500 TF / 6 partial products per fp32 product",
                  "f16x2": "dense fp16 MFMA 2500 TF / 3 partial products per fp32 product"}[GEMM_MODE]
DTYPE_NOTE = {
    "fp32": "f32",
    "bf16x3": "f32 (every fp32 operand split exactly into 3 bf16 pieces on the bf16 matrix cores, fp32 accumulate; the 6 "
              "partial products of weight >= 2^-24 are kept, the 3 dropped ones are <= 3*2^-24 |a||b| per product: "
              "fp32-GEMM-class accuracy, <= 4e-6*scale against fp64 in tests/test_gpu_parity.py::test_linear_fuzz..., "
              "not IEEE-fp32 bitwise; PointNet conv3 + max over the points: fp16 matrix-core filter that only SELECTS "
              "candidate points, every emitted value is a plain fp32 FMA dot product)",
    "f16x2": "f32 (GEMMs: every fp32 operand as two fp16 pieces, a1 = fp16(a), a2 = fp16((a - a1) * 2^11), weights pre-scaled by "
             "one power of two per output row; the 3 partial products a1 b1, a1 b2, a2 b1 are exact in fp32 and accumulate in "
             "TWO fp32 accumulators (large / cross terms) on v_mfma_f32_16x16x32_f16; dropped: a2 b2 and the two residuals, "
             "<= 3*2^-22 |a||b| per product: fp32-GEMM-class accuracy, <= 4e-6*scale against fp64 in "
             "tests/test_gpu_parity.py::test_linear_fuzz... (measured ~1e-7*scale, a third of the six-product bf16 split kept "
             "behind DVQ_GEMM=bf16x3), not IEEE-fp32 bitwise; PointNet conv1: fp32 vector ALU; conv2: the same fp16 three-product "
             "split with a per-point power-of-two activation scale (the six-product bf16 split only with DVQ_PN_FILTER=0 and for "
             "shapes the filter does not cover); conv3 + max over the points: fp16 matrix-core filter that only SELECTS "
             "candidate points, every emitted value is a plain fp32 FMA dot product; VQ argmin: fp16 matrix-core filter + "
             "canonical fp32 refine, indices bit-exact)"}[GEMM_MODE]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=65536, help="grasps per step: global (strong scaling) or per rank (weak)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--prof-steps", type=int, default=1, help="steps of the second (profiled) pass; 0 = no kernel breakdown")
    ap.add_argument("--points", type=int, default=1024)
    ap.add_argument("--codebook", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-grasps", type=int, default=256, help="bounded CPU-baseline sample of the batched port (grasps)")
    ap.add_argument("--vq-iters", type=int, default=20)
    ap.add_argument("--vq-train", type=int, default=1000, help="calls of the VQ microbench's timed train (after 300 untimed ones: the chip's clock settles over the first few hundred calls)")
    ap.add_argument("--no-prof", action="store_true", help="skip the second (profiled) pass")
    ap.add_argument("--vq-only", action="store_true", help="only the VQ argmin microbench (BASELINE config 2)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4],
                    help="0 (default): the headline (BASELINE.json's metric); 2: the VQ argmin microbench (= --vq-only); 3: PointNet -> VQ -> "
                         "decoder (+ MANO -> hand PointNet -> wrist decoder) without the prior, batch 16384, N=1024; 4: gated-PixelCNN "
                         "sampling + decode, batch 8192, and the full gen() on HO3D-sized clouds (N=3000) beside it.  Each prints its own line")
    ap.add_argument("--vq-tie-prone", action="store_true",
                    help="also time the reference-init codebook U(+-1/K) (SURVEY 8d second run); off by default so that the "
                         "fast kernel's rocprofv3 average reflects the headline workload only")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl = RCCL on GPUs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank on cuda:0 with gloo: exercises the N > 1 path on a one-GPU box (numbers meaningless)")
    ap.add_argument("--force-pg", action="store_true",
                    help="create the process group and run every collective even at world size 1 (first contact with RCCL on a "
                         "one-GPU box: init, all_gather_into_tensor, barrier, all_reduce)")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch latency leg (B = 1, 8, 100)")
    ap.add_argument("--sources-digest", action="store_true", help="print the digest of the kernel sources (what a PMC file must carry to be quoted) and exit")
    return ap.parse_args()


def relaunch_if_needed(args):
    """`python bench.py --gpus N` without a launcher: start torchrun as a child (before any GPU use)."""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))


_RANK_LOG = None


def log(msg):
    line = f"[bench {time.strftime('%H:%M:%S')}] {msg}"
    print(line, file=sys.stderr, flush=True)
    if _RANK_LOG is not None:
        print(line, file=_RANK_LOG, flush=True)


def usable_cores():
    """Cores this process may actually run on: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, n)


def kernel_sources_sha256():
    """One digest over what the library's kernels are built from (d-vqvae_amd/csrc: *.hip, *.h but not the generated header, the
    Makefile; the generator of the generated header): tools/collect_profiles.sh stores it beside the counters it collects, and
    pmc_traffic() quotes a committed PMC file only while the digest still matches."""
    import glob, hashlib
    src = os.path.join(ROOT, "d-vqvae_amd", "csrc")
    files = sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + [os.path.join(src, "Makefile"),
                   os.path.join(ROOT, "tools", "gen_vq_pipe.py")])
    h = hashlib.sha256()
    for f in files:
        if os.path.basename(f) == "vq_pipe_loop.h":
            continue
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read() + b"\0")
    return h.hexdigest()


def pmc_file_is_current(doc):
    """(True, "") when the PMC file `doc` was collected on THIS tree's kernels, else (False, reason): the digest of the kernel
    sources stored with the counters must equal this tree's, and where a git history is at hand (not on the GPU box) the commit
    it names must be HEAD or an ancestor of it."""
    want = doc.get("kernel_sources_sha256")
    if not want:
        return False, "the file carries no digest of the kernel sources it was collected on"
    if want != kernel_sources_sha256():
        return False, "the kernel sources changed since it was collected (digest differs)"
    commit = doc.get("commit")
    if commit and os.path.isdir(os.path.join(ROOT, ".git")):
        import subprocess
        try:
            r = subprocess.run(["git", "-C", ROOT, "merge-base", "--is-ancestor", commit, "HEAD"], capture_output=True, timeout=20)
            if r.returncode != 0:
                return False, f"its commit {commit} is not HEAD or an ancestor of HEAD"
        except Exception:
            pass
    return True, ""


def pmc_traffic(kind, batch):
    """(HBM bytes per launch, source) of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE, profiles/*_pmc_hbm_traffic.json: an EARLIER run of the same command, collected at the default batch only);
    (None, None) if there is none, (None, "refused: ...") if the newest file was collected on other kernels than this tree's
    (pmc_file_is_current).  Not measured in this run: PMC collection needs rocprofv3 around the process.
    A profiler key covers every template variant the library launches under it (gemm_gate: plain launches <2, 4, false> and
    launches that continue a class-table accumulator state <2, 4, true>): the figure is the launch-weighted mean."""
    if batch != 65536:
        return None, None
    # kernel names as rocprofv3 prints them (truncated in the committed files): a PREFIX of the template-argument list, so that
    # "<2, 4>" (before the accumulator-state argument), "<2, 4, false>" and "<2, 4, true>" all match
    names = {"vq_fast": ("::vq_stream", "::vq_pipe"),      # vq_stream16_kernel (default) / vq_stream_kernel (DVQ_VQ_KERNEL=8) / vq_pipe_kernel (17)
             "pn_trunk": ("pn_trunk_filter_kernel<4, false", "pn_trunk_filter_kernel<3, false", "pn_trunk_filter_kernel<4>", "pn_trunk_filter_kernel<3>"),
             "pn_exact": ("pn_exact_kernel",),
             "gemm_gate": ("gemm_f16x2_pp_kernel<2,", "gemm_f16x2_pp_kernel<2>"),
             "gemm_bias": ("gemm_f16x2_pp_kernel<0,", "gemm_f16x2_pp_kernel<0>"),
             "gemm_resid": ("gemm_f16x2_pp_kernel<1,", "gemm_f16x2_pp_kernel<1>")}
    if GEMM_MODE == "bf16x3":
        names.update({"gemm_gate": ("gemm_bf16x3_wide_kernel<2,",), "gemm_bias": ("gemm_bf16x3_wide_kernel<0,",), "gemm_resid": ("gemm_bf16x3_wide_kernel<1,",)})
    try:
        import glob
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))[-1]
        doc = json.load(open(path))
        ok, why = pmc_file_is_current(doc)
        if not ok:
            return None, f"refused: profiles/{os.path.basename(path)} is stale -- {why}"
        want_scope = "vq microbench" if kind == "vq_fast" else None
        tot_b = tot_n = 0.0
        for row in doc["per_launch_bytes"]:
            if row.get("scope") != want_scope and not (kind == "vq_fast" and tot_n == 0 and row.get("scope") is None):
                continue
            if any(nm in row["kernel"] for nm in names.get(kind, ())):
                if kind == "vq_fast" and row.get("scope") == "vq microbench":
                    tot_b, tot_n = 0.0, 0.0                      # the microbench's own pass wins over the step's
                tot_b += row["hbm_bytes_corrected"] * row["launches"]
                tot_n += row["launches"]
        if tot_n:
            return tot_b / tot_n, f"profiles/{os.path.basename(path)} (rocprofv3 PMC passes of {doc.get('collected', 'an earlier run')})"
    except Exception:
        pass
    return None, None


def prof_read(lib, _lib):
    buf = (_lib.ProfEntry * 64)()
    n = lib.dvq_prof_read(buf, 64)
    return {buf[i].name.decode(): dict(count=int(buf[i].count), ms=float(buf[i].ms), flops=float(buf[i].flops),
                                        bytes=float(buf[i].bytes)) for i in range(min(n, 64))}


def vq_microbench(args, lib, _lib, ops, dev, K):
    """BASELINE config 2: VectorQuantizer K=512 D=256 argmin-only, M=65536, against the HBM roofline.
    Algorithmic bytes per call: M*D*4 (z) + K*D*4 (codebook) + M*8 (int64 indices) = 68 157 440."""
    import torch
    log("vq_argmin microbench")
    M, D = 65536, 256
    # six distinct 64 MiB inputs, used round-robin: 384 MiB > the 256 MiB Infinity Cache, so every call streams z from HBM
    zs = [torch.randn(M, D, device=dev) for _ in range(6)]
    z = zs[0]
    E = torch.randn(K, D, device=dev)
    packed = ops.vq_pack(E) if ops.vq_fast_supported(K, D) else None      # once per codebook (a parameter)
    for i in range(6):
        idx = ops.vq_argmin(zs[i], E, packed=packed)
    torch.cuda.synchronize(dev)
    # (a) per-launch HIP events inside the library (kernel-by-kernel)
    lib.dvq_prof_reset(); lib.dvq_prof_enable(1)
    for i in range(args.vq_iters):
        idx = ops.vq_argmin(zs[i % 6], E, packed=packed)
    torch.cuda.synchronize(dev)
    lib.dvq_prof_enable(0)
    pk = prof_read(lib, _lib); lib.dvq_prof_reset()
    # (b) one event pair around a back-to-back train of calls on the launch stream: launch gaps included, event cost amortised.
    # The train is long enough for the steady state (tools/vq_clock_probe.py: a 30-call train after an idle gap gives 37-40 us per
    # call, trains of 1 000 and more 32 us -- the clock settles over the first few hundred calls), after an untimed warm-up train.
    n_train = max(args.vq_train, 30)
    for i in range(300):
        idx = ops.vq_argmin(zs[i % 6], E, packed=packed)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n_train):
        idx = ops.vq_argmin(zs[i % 6], E, packed=packed)
    e1.record()
    torch.cuda.synchronize(dev)
    dur = e0.elapsed_time(e1) * 1e-3 / n_train
    idx = ops.vq_argmin(z, E, packed=packed)
    alg_bytes = M * D * 4 + K * D * 4 + M * 8
    gbs = alg_bytes / dur / 1e9
    exact = ops.vq_argmin(z, E, fast=False)
    ref_idx = torch.argmin((z ** 2).sum(1, keepdim=True) + (E ** 2).sum(1) - 2 * z @ E.t(), dim=1)
    res = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
           "traffic": pmc_traffic("vq_fast", 65536)[0], "traffic_source": pmc_traffic("vq_fast", 65536)[1], "us_per_call": dur * 1e6, "algorithmic_bytes": alg_bytes, "rows_per_s": M / dur,
           "bit_match_vs_exact_fp32_kernel": float((idx == exact).float().mean()),
           "index_match_rate_vs_torch_gpu_expr": float((idx == ref_idx).float().mean()),
           "workload": "VectorQuantizer K=512 D=256 argmin-only, M=65536; duration = every kernel of one call "
                       "(one persistent kernel: z streamed once under an fp16-MFMA filter with the codebook in registers, exact fp32 "
                       "refine of the ambiguous rows in the workgroup), codebook packed once",
           "timing": f"one HIP-event pair around a train of {n_train} back-to-back calls on the launch stream after 300 untimed ones, 6 rotating 64 MiB inputs",
           "kernels_us": {k: v["ms"] / v["count"] * 1e3 for k, v in pk.items()}}
    # the exact fp32-MFMA kernel, for comparison
    lib.dvq_prof_reset(); lib.dvq_prof_enable(1)
    for _ in range(5):
        ops.vq_argmin(z, E, fast=False)
    torch.cuda.synchronize(dev)
    lib.dvq_prof_enable(0)
    pk = prof_read(lib, _lib); lib.dvq_prof_reset()
    res["exact_fp32_kernel_us"] = pk["vq_argmin_total"]["ms"] / pk["vq_argmin_total"]["count"] * 1e3
    # SURVEY 8(d) second run: the reference's initial codebook U(+-1/K) against the same O(1) rows -- ill-conditioned
    # (fp32 itself cancels |z|^2 against a 1e-2 spread).  Forced fast kernel vs exact kernel; the VectorQuantizer module
    # watches the kernel's slow-row counter and uses the exact kernel there (tests/test_gpu_parity.py).
    if not args.vq_tie_prone:
        return res
    Et = (torch.rand(K, D, device=dev) * 2 - 1) / K
    pt = ops.vq_pack(Et)
    slow = torch.zeros(1, dtype=torch.int64, device=dev)
    it = ops.vq_argmin(z, Et, packed=pt, slow_rows=slow)
    et = ops.vq_argmin(z, Et, fast=False)
    torch.cuda.synchronize(dev)

    def timed(fn, n=3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / n
    res["tie_prone_regime"] = {"codebook": "U(-1/512, 1/512) (reference init, quantizer.py:27)", "rows": "N(0,1)",
                               "fast_kernel_forced_us": timed(lambda: ops.vq_argmin(z, Et, packed=pt)),
                               "exact_kernel_us": timed(lambda: ops.vq_argmin(z, Et, fast=False)),
                               "slow_row_fraction": float(slow.item()) / M,
                               "bit_match_fast_vs_exact": float((it == et).float().mean()),
                               "module_choice": "exact kernel (VectorQuantizer switches when > 1/16 of the rows are slow)"}
    return res


def latency_leg(net, synth, dev, K):
    """The reference's own call pattern (gen_diverse_grasp_ho3d.py:212-236: B = 1 per call, 1 / 20 / 49 / 100 grasps per object;
    BASELINE config 1 is batch 8): wall time of ONE GenNet.gen call, host call -> results synchronised, median of 7 after
    two warm-up calls, at N = 1024 and the datasets' N = 3000."""
    import torch
    res = {"what": "one GenNet.gen call + 61-parameter assembly, host wall clock incl. the final synchronisation, median of 7"}
    for n_pts in (1024, 3000):
        row = {}
        for b in (1, 8, 100):
            obj = synth.synthetic_clouds(b, n_pts, seed=900 + b).to(dev)
            ts = []
            for it in range(9):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                net.gen(obj, seed=5, row0=0, stream_id=it)
                torch.cuda.synchronize(dev)
                ts.append((time.perf_counter() - t0) * 1e3)
            row[f"B={b}"] = sorted(ts[2:])[3]
        res[f"N={n_pts}"] = row
    # the reference's loop of B = 1 calls (gen_diverse_grasp_ho3d.py:212-236), here without a host synchronisation per call
    # (gen(check=False): the error flags stay on the device and are read once after the loop): ms per grasp over 32 calls
    obj = synth.synthetic_clouds(1, 1024, seed=901).to(dev)
    flags = []
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for it in range(32):
        _, _, aux = net.gen(obj, seed=5, row0=0, stream_id=100 + it, check=False, return_aux=True)
        flags.append(aux["err"])
    bad = int(torch.stack(flags).max().item())
    torch.cuda.synchronize(dev)
    res["B=1 loop, no per-call host sync (N=1024)"] = {"ms_per_grasp": (time.perf_counter() - t0) * 1e3 / 32, "error_flags": bad}
    log(f"latency: {res}")
    return res


def cpu_info():
    """CPU model, logical cores, torch version and BLAS backend (BASELINE.md 3 asks for them next to the baseline)."""
    import torch
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    cfg = torch.__config__.show()
    blas = "mkl" if "BLAS_INFO=mkl" in cfg else ("open" if "BLAS_INFO=open" in cfg else "other")
    return {"cpu_model": model, "logical_cores": os.cpu_count(), "usable_cores": usable_cores(), "torch": torch.__version__, "blas": blas}


def cpu_baseline(sd, arrays, n_grasps, points, codebook, net=None, dev=None):
    """BASELINE.md 3 on a bounded sample (~30 s in all).  The CPU oracle (a port of the reference's algorithm, naive
    9-forward prior as the reference runs it):
      A  reference-faithful: a loop of B=1 gen calls (the only batch size the reference supports), N=1024 and N=3000,
         up to 64 grasps or 8 s each;
      B  batched port at B=256/call (per-sample label, row-gather lookup; NOT reference behaviour, a stronger baseline):
         ``value`` of the returned object;
      micro: the VectorQuantizer argmin expression at M=65536, K=512, D=256.
    With ``net``: the HIP path is run on the first 64 grasps of B's sample and compared with the oracle (the checker, never the
    thing measured): code match rates and the largest parameter difference go into the ``parity`` object."""
    import torch
    from dvqvae_amd import synth
    from oracle import dvq_oracle, mano_oracle
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} cores (os.cpu_count()={os.cpu_count()})")
    mano = mano_oracle.ManoOracle(arrays)
    cpu_sd = {k: v.cpu() for k, v in sd.items()}
    bsz = min(256, n_grasps)
    obj = synth.synthetic_clouds(bsz, points, seed=4242)
    q = synth.exp1_noise(bsz, 9, codebook, seed=4243)
    loops = {}
    with torch.no_grad():
        dvq_oracle.gen(cpu_sd, obj[:2], q[:2], mano)                      # warm-up
        t0 = time.perf_counter()
        o_out = dvq_oracle.gen(cpu_sd, obj, q, mano, return_aux=True)     # B: one batched call
        dt = time.perf_counter() - t0
        log(f"cpu baseline B: {bsz} grasps batched in {dt:.1f} s")
        for n_pts in (points, 3000):                                       # A: the reference's own call pattern
            o1 = obj if n_pts == points else synth.synthetic_clouds(64, n_pts, seed=4244)
            t1, done = time.perf_counter(), 0
            while done < 64 and (done == 0 or time.perf_counter() - t1 < 8.0):
                dvq_oracle.gen(cpu_sd, o1[done:done + 1], q[done:done + 1], mano)
                done += 1
            loops[f"N={n_pts}"] = {"grasps": done, "grasps_per_s": done / (time.perf_counter() - t1)}
            log(f"cpu baseline A (B=1 loop, N={n_pts}): {done} grasps, {loops[f'N={n_pts}']['grasps_per_s']:.2f} grasps/s")
        zq, Eq = torch.randn(65536, 256), torch.randn(512, 256)
        t2 = time.perf_counter()
        for _ in range(3):
            torch.argmin((zq ** 2).sum(1, keepdim=True) + (Eq ** 2).sum(1) - 2 * zq @ Eq.t(), dim=1)
        vq_ms = (time.perf_counter() - t2) / 3 * 1e3
    parity = None
    if net is not None:
        n = bsz                                            # the whole sample of the batched leg
        with torch.no_grad():
            recon, pos, aux = net.gen(obj[:n].to(dev), noise=q[:n].to(dev), return_aux=True)
        o_recon, o_pos, o_aux = o_out
        # grasps whose decision an fp32 rounding difference could flip are set aside by stated margins and counted
        idx_margin = dvq_oracle.object_code_margin(o_aux, aux["feat_type"])    # from the measured feature difference, per grasp
        safe_idx = (o_aux["idx6_gap"] > idx_margin)[:n]
        safe_race = (o_aux["race_gap"] > 1e-4)[:n]          # fp32 logits differ by ~1e-6 relative between the two paths
        safe = safe_idx & safe_race
        idx_ok = (aux["idx6"].cpu() == o_aux["idx6"][:n]).reshape(n, -1).all(dim=1)
        code_ok = (aux["codes"].cpu() == o_aux["codes"][:n]).reshape(n, -1).all(dim=1)
        both = idx_ok & code_ok
        d = torch.cat([(recon.cpu() - o_recon[:n]).abs(), (pos.cpu() - o_pos[:n]).abs()], dim=1)
        parity = {"grasps": n, "checker": "oracle/dvq_oracle.py (CPU fp32 port, pinned to the reference's goldens)",
                  "distinct_object_codes": len(set(o_aux["idx6"][:n].reshape(-1).tolist())),
                  "excluded_by_gap": int((~safe_idx).sum()), "excluded_by_race": int((safe_idx & ~safe_race).sum()),
                  "checked": int(safe.sum()),
                  "idx6_match_rate": float(idx_ok[safe_idx].float().mean()) if bool(safe_idx.any()) else None,
                  "sampled_codes_match_rate": float(code_ok[safe].float().mean()) if bool(safe.any()) else None,
                  "idx6_match_rate_all": float(idx_ok.float().mean()), "sampled_codes_match_rate_all": float(code_ok.float().mean()),
                  "max_abs_param_diff_on_matched_codes": float(d[both].max()) if bool(both.any()) else None,
                  "tolerance": 1e-5,
                  "max_abs_feature_diff": float((aux["feat_type"].cpu() - o_aux["feat_type"][:n]).abs().max()),
                  "margins": "object code: fp64 top-2 distance gap > 2 |delta z| |e_1 - e_2| + 32 ulp (|z|^2 + |e|^2), delta z = the measured "
                             "GPU-vs-oracle PointNet feature difference of the grasp (oracle/dvq_oracle.py:object_code_margin); sampled "
                             "codes: exponential-race margin > 1e-4 (relative); match rates are over the grasps inside the margins, "
                             "*_all over every grasp"}
        # the object-code margin is derived from the MEASURED feature difference: a larger GPU error widens its own tolerance, so the
        # share of grasps it may set aside is capped (the same 2 % tests/test_gpu_parity.py::test_gen_bench_config_vs_oracle asserts)
        parity["excluded_by_gap_cap"] = 0.02
        parity["ok"] = bool(parity["excluded_by_gap"] <= 0.02 * n and parity["checked"] > 0 and parity["idx6_match_rate"] == 1.0
                            and parity["sampled_codes_match_rate"] == 1.0
                            and (parity["max_abs_param_diff_on_matched_codes"] or 0.0) <= 1e-5)
        if not parity["ok"]:
            log(f"PARITY CHECK FAILED: {parity}")
    return {"value": bsz / dt, "unit": "grasps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"B: {bsz} grasps in one batched call (N={points}, K={codebook}, naive 9-forward prior), {dt:.1f} s; "
                      f"A (reference-faithful B=1 loop): {loops}; VQ argmin expression M=65536: {vq_ms:.0f} ms",
            "reference_faithful_b1_loop": loops, "vq_argmin_expr_ms": vq_ms, **cpu_info()}, parity


def mfma_entry(kernels, name, pe_ms, B):
    """Roofline entry of ONE matrix-core kernel of the step: algorithmic FLOPs per launch over its average launch time (per-launch
    HIP events on the launch stream) against the roof of the arithmetic it runs.
      gemm_*    every fp32 product as THREE fp16 products (f16x2; six bf16 with DVQ_GEMM=bf16x3): dense peak / 3 (/ 6)
      pn_trunk  the filtered PointNet trunk: conv3 (94 % of the FLOPs) as ONE fp16 product -> the dense fp16 peak; conv2's six
                bf16 products count once (an over-statement of the roof for 6 % of the FLOPs, i.e. the fraction is a lower bound);
                with DVQ_PN_FILTER=0 the six-product trunk: peak / 6."""
    v = kernels[name]
    filtered = "pn_exact" in kernels
    if name == "pn_trunk":
        peak = BF16_MFMA_PEAK_TF if filtered else BF16_MFMA_PEAK_TF / 6.0
        note = ("dense fp16 MFMA 2500 TF: conv3 is ONE fp16 product per term (a filter; the exact fp32 re-evaluation of its candidates is "
                "pn_exact)") if filtered else "dense bf16 MFMA 2500 TF / 6 partial products"
        label = ("pn_trunk_filter_kernel (conv1 + conv2 + conv3 filter); one launch here = the kernel over <= 4096 clouds, plus its one-block "
                 "tail kernel for the 778-vertex hand clouds: per step 64 launches on N=1024 object clouds (rocprofv3: pn_trunk_filter_kernel<4, false>) "
                 "and 32 on hand clouds (<3, false> + <3, true>)") if filtered else "pn_trunk_kernel (fused PointNet trunk)"
    else:
        peak, note, label = GEMM_PEAK_TF, GEMM_PEAK_NOTE, f"gemm_{GEMM_MODE} {name}"
    achieved = v["flops"] / (v["ms"] * 1e-3) / 1e12
    tr, tr_src = pmc_traffic(name, B)
    alg_bytes = v["bytes"] / v["count"] if v["count"] else None
    if name == "pn_trunk":
        # the library's byte count for this kernel includes its own scratch (the h2 rows it spills: 512 B per point); ALGORITHMIC
        # bytes are the cloud in (16 B per point) and one 4 KB feature row per cloud out
        pts = v["flops"] / v["count"] / (2.0 * (4 * 64 + 64 * 128 + 128 * 1024))
        alg_bytes = pts * 16 + pts / 1024.0 * 4096
    return {"bound": "mfma", "kernel": label, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "peak_note": note, "launches": v["count"], "avg_launch_ms": v["ms"] / v["count"], "flops_per_launch": v["flops"] / v["count"],
            "share_of_step": v["ms"] / pe_ms, "traffic": tr, "traffic_source": tr_src, "algorithmic_bytes": alg_bytes,
            "traffic_over_algorithmic": (tr / alg_bytes) if (tr and alg_bytes) else None}


def step_roofline(kernels, pe_ms, prof_steps, ms_per_step, B):
    """The ``roofline`` object of the headline line: the matrix-core kernel with the LARGEST share of the step (over every MFMA
    kernel: GEMMs and the PointNet trunk; the exact VQ argmin runs the fp32 chain and is excluded), the dominant GEMM beside it
    when that is another kernel, and the PointNet group."""
    mf = [k for k in kernels if (k.startswith("gemm_") and k != "gemm_argmin") or k == "pn_trunk"]
    dom = max(mf, key=lambda k: kernels[k]["ms"])
    r = mfma_entry(kernels, dom, pe_ms, B)
    r["measured_in"] = (f"second pass of {prof_steps} step(s) with per-launch HIP events, the PointNet on ONE stream so that every "
                        f"launch is timed alone ({pe_ms / prof_steps:.1f} ms per step against {ms_per_step:.1f} ms in the timed region, "
                        f"where its exact stage overlaps the next launch's trunk kernel)")
    gm = [k for k in mf if k.startswith("gemm_")]
    if gm:
        gdom = max(gm, key=lambda k: kernels[k]["ms"])
        if gdom != dom:
            r["dominant_gemm"] = mfma_entry(kernels, gdom, pe_ms, B)
        g_ms = sum(kernels[k]["ms"] for k in gm)
        g_fl = sum(kernels[k]["flops"] for k in gm)
        r["all_gemm_kernels"] = {"achieved": g_fl / (g_ms * 1e-3) / 1e12, "peak": GEMM_PEAK_TF, "frac": g_fl / (g_ms * 1e-3) / 1e12 / GEMM_PEAK_TF,
                                 "share_of_step": g_ms / pe_ms}
    if "pn_exact" in kernels:
        pn_ms = sum(kernels[k]["ms"] for k in ("pn_center", "pn_trunk", "pn_exact") if k in kernels)
        ex_tr, ex_src = pmc_traffic("pn_exact", B)
        r["pointnet_trunks"] = {
            "kernels": "pn_center_kernel + pn_trunk_filter_kernel + pn_exact_kernel", "ms_per_step": pn_ms / prof_steps,
            "share_of_step": pn_ms / pe_ms, "algorithmic_tflops": kernels["pn_trunk"]["flops"] / (pn_ms * 1e-3) / 1e12,
            "frac_of_dense_fp16_peak": kernels["pn_trunk"]["flops"] / (pn_ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF,
            "pn_exact_traffic": ex_tr, "pn_exact_traffic_source": ex_src,
            "note": "conv1 (vector ALU) / conv2 (split matrix-core products, fp32-accurate), conv3 + max as an fp16 matrix-core filter "
                    "(1 product) + exact fp32 re-evaluation of the candidate points; result bit-identical to the exhaustive exact "
                    "maximum (tests/test_gpu_parity.py::test_pointnet_filter_equals_exhaustive_exact_evaluation); algorithmic FLOPs = "
                    "2*points*(4*64+64*128+128*1024)"}
        if dom != "pn_trunk":
            r["pointnet_trunks"]["trunk_kernel"] = mfma_entry(kernels, "pn_trunk", pe_ms, B)
    return r


def config_leg(args, lib, _lib, dev):
    """BASELINE configs 3 and 4 as timed lines of their own (SURVEY 8d gives each its own bound):
      3  PointNet -> VQ -> decoder forward at batch 16 384, N = 1024: gen() WITHOUT the prior -- two object PointNets, the
         object-code argmin, six codebook lookups of GIVEN codes, decoder, MANO, hand PointNet, wrist decoder, 61-parameter
         assembly: 1 592 MFLOP per grasp, fp32-peak bound 157.3 T / 1 592 M = 98.8 k grasps/s;
      4  gated-PixelCNN sampling + decode at batch 8 192: device noise, the cached sampler, six lookups, decoder: 1.618 GFLOP
         per grasp + the weights once per pass, fp32-peak bound 97 k grasps/s; the whole gen() on HO3D-sized clouds
         (N = 3000) is timed beside it.
    Same timing rule as the headline: inputs resident, W untimed passes, K timed passes between two synchronisations."""
    import torch
    from dvqvae_amd import mano as dmano, ops, synth
    from dvqvae_amd.network.gen_net import GenNet, CODE_SLOTS
    K = args.codebook
    cfg = args.config
    B = {3: 16384, 4: 8192}[cfg] if args.batch == 65536 else args.batch
    N = args.points
    net = GenNet(n_embeddings=K, prior_tokens=K, prior_classes=K)
    sd = synth.synthetic_state_dict(net.state_dict(), 1234)
    net.load_state_dict(sd)
    net.eval().to(dev)
    synth.diversify_object_codebook(net, sd, N)
    net.set_noise_seed(20261003)
    net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(dev))
    pool = synth.synthetic_clouds(min(B, 4096), N, seed=1000).to(dev)
    rows = torch.arange(B, device=dev)
    obj = pool[rows % pool.shape[0]].contiguous()
    obj[:, :3] += ((rows // pool.shape[0]).float() * 1e-3)[:, None, None]

    def timed(fn):
        for _ in range(max(args.warmup, 1)):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / args.steps, out

    def profiled(fn):
        lib.dvq_prof_reset(); lib.dvq_prof_enable(1)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        pe = (time.perf_counter() - t0) * 1e3
        lib.dvq_prof_enable(0)
        k = prof_read(lib, _lib); lib.dvq_prof_reset()
        return k, pe

    step_no = [0]
    if cfg == 3:
        codes = torch.randint(0, K, (B, 3, 3), device=dev, dtype=torch.int64, generator=torch.Generator(device=dev).manual_seed(7))

        @torch.no_grad()
        @ops.no_range_check()        # as GenNet.gen: no per-op finiteness check + host sync inside the timed region (one check below)
        def step():
            z_out = torch.empty(B, 2560, device=dev)
            z_pos = torch.empty(B, 2048, device=dev)
            net.obj_encoder_type(obj, out=z_out[:, 1536:])
            net.obj_encoder_pos(obj, out=z_pos[:, 1024:])
            idx6, _ = net.vqvae6.inference(z_out[:, 1536:])
            err = ops.new_err_flag(dev)
            recon = net._decode(codes, {"z_out": z_out}, None, err)
            verts = net._hand_vertices(recon)
            net.recon_encoder(verts, out=z_pos[:, :1024])
            pos = net.pos_decoder(z_pos).view(B, 6)
            return ops.assemble61(recon, pos), idx6
        dt, (p61, idx6) = timed(step)
        assert bool(torch.isfinite(p61).all())
        flop_per_grasp, bound = 1592e6, FP32_MFMA_PEAK_TF * 1e12 / 1592e6
        workload = (f"config 3: two object PointNets (N={N}) -> object-code argmin -> six codebook lookups of given codes -> decoder -> "
                    f"MANO -> hand PointNet (778 vertices) -> wrist decoder -> 61-parameter assembly, batch {B}, no prior")
        extra = {"distinct_object_codes": int(torch.unique(idx6).numel())}
        kernels, pe_ms = profiled(step)
    else:
        label = torch.randint(0, K, (B,), device=dev, dtype=torch.int64, generator=torch.Generator(device=dev).manual_seed(8))
        order = torch.argsort(label, stable=True)
        label_s = label[order].contiguous()
        pk = net.GatedPixelCNN.packed()
        feat = torch.randn(B, 1024, device=dev, generator=torch.Generator(device=dev).manual_seed(9))

        @torch.no_grad()
        @ops.no_range_check()
        def step():
            step_no[0] += 1
            noise = ops.exp1_noise(B, 9 * pk.n_in, 20261003, 0, step_no[0], device=dev, perm=order).view(B, 9, pk.n_in)
            err = ops.new_err_flag(dev)
            codes_s = ops.pixelcnn_sample(pk, label_s, noise, err=err)
            codes = torch.empty_like(codes_s)
            codes[order] = codes_s
            z_out = torch.empty(B, 2560, device=dev)
            z_out[:, 1536:] = feat
            return net._decode(codes, {"z_out": z_out}, None, err), err
        dt, (recon, err) = timed(step)
        assert int(err.item()) == 0 and bool(torch.isfinite(recon).all())
        flop_per_grasp, bound = 1618e6, 97.0e3
        workload = (f"config 4: device Philox noise -> cached gated-PixelCNN sampler (15 layers, 3x3 grid, {K} classes, label-ordered) -> "
                    f"six codebook lookups -> decoder, batch {B}")
        kernels, pe_ms = profiled(step)
        # the whole path on HO3D-sized clouds beside it
        n_ho3d = 3000
        pool3 = synth.synthetic_clouds(min(B, 1024), n_ho3d, seed=1001).to(dev)
        obj3 = pool3[rows % pool3.shape[0]].contiguous()
        obj3[:, :3] += ((rows // pool3.shape[0]).float() * 1e-3)[:, None, None]
        dt3, _ = timed(lambda: net.gen(obj3, seed=20261003, row0=0, stream_id=1000 + step_no[0]))
        extra = {"full_gen_ho3d": {"workload": f"GenNet.gen, N={n_ho3d} points per cloud, batch {B}", "ms_per_step": dt3 * 1e3,
                                   "grasps_per_s": B / dt3}}
    value = B / dt
    out = {"metric": f"grasps/sec, BASELINE config {cfg}", "value": value, "unit": "grasps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": max(args.warmup, 1), "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": DTYPE_NOTE, "data": "synthetic",
           "config": {"workload": workload, "batch": B, "points": N, "codebook": K},
           "bound": {"what": "SURVEY 8(d): algorithmic FLOPs per grasp at the fp32 peak (157.3 TFLOP/s): the bound an fp32 implementation "
                             "of the reference's arithmetic would have; the split-fp16 / filtered kernels here may exceed it",
                     "flop_per_grasp": flop_per_grasp, "grasps_per_s": bound, "frac": value / bound,
                     "algorithmic_tflops": value * flop_per_grasp / 1e12}}
    out.update(extra)
    if kernels:
        out["roofline"] = step_roofline(kernels, pe_ms, 1, dt * 1e3, B)       # PMC traffic exists for the headline's launches only
        out["kernels"] = {k: {"count": v["count"], "ms": round(v["ms"], 3),
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0}
                          for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}
    return out


def main():
    args = parse()
    if args.sources_digest:
        print(kernel_sources_sha256())
        return
    relaunch_if_needed(args)
    import torch
    import dvqvae_amd
    from dvqvae_amd import _lib, dist, mano as dmano, ops, synth
    from dvqvae_amd.network.gen_net import GenNet

    rank, local_rank, world = dist.init(args.backend, args.share_gpu, args.force_pg)
    if world > 1:
        # every rank's stderr (its own log lines, tracebacks, RCCL's native messages) also lands in bench_rank<r>.log: ranks other
        # than 0 write there only, rank 0 keeps the console as well
        path = os.path.join(os.environ.get("DVQ_BENCH_LOG_DIR", "."), f"bench_rank{rank}.log")
        try:
            if rank == 0:
                global _RANK_LOG
                _RANK_LOG = open(path, "w")
            else:
                fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
                os.dup2(fd, 2)
        except OSError as e:
            log(f"no per-rank log file ({e})")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    lib = _lib.load()

    N, K = args.points, args.codebook
    if args.config in (3, 4):
        assert world == 1, "--config 3 / 4 are single-GPU legs"
        print(json.dumps(config_leg(args, lib, _lib, dev)), flush=True)
        dist.shutdown()
        return
    if args.vq_only or args.config == 2:
        print(json.dumps({"roofline_vq_argmin": vq_microbench(args, lib, _lib, ops, dev, K)}), flush=True)
        return
    # rows of the global batch this rank generates
    if args.scaling == "strong":
        B_global = args.batch
        lo, hi = dist.shard_range(B_global, rank, world)
    else:
        B_global = args.batch * world
        lo, hi = rank * args.batch, (rank + 1) * args.batch
    B = hi - lo
    net = GenNet(n_embeddings=K, prior_tokens=K, prior_classes=K)
    sd = synth.synthetic_state_dict(net.state_dict(), 1234)
    net.load_state_dict(sd)
    net.eval().to(dev)
    # object codebook = the net's own object-type features of K seed clouds: the object code differs from grasp to grasp
    # (random codebook rows are all equally far from every synthetic cloud's feature: one code for the whole batch)
    sd = synth.diversify_object_codebook(net, sd, N)
    net.set_noise_seed(20261003)
    arrays = dmano.synthetic_mano_arrays()
    net.set_rh_mano(dmano.ManoLayer(arrays).to(dev))

    # this rank's clouds: a pool of 4096 synthetic objects tiled over the shard (host RAM / time bound), de-duplicated by a
    # per-row offset that depends on the GLOBAL row (so a row's input does not depend on the sharding)
    pool = synth.synthetic_clouds(min(max(B_global, 1), 4096), N, seed=1000).to(dev)   # keyed by the GLOBAL batch: same pool on every rank
    rows = torch.arange(lo, hi, device=dev)
    obj = pool[rows % pool.shape[0]].contiguous()
    obj[:, :3] += ((rows // pool.shape[0]).float() * 1e-3)[:, None, None]
    gathered = None
    step_no = [0]
    strong = args.scaling == "strong"

    def step():
        nonlocal gathered
        recon, pos = net.gen(obj, seed=20261003, row0=lo, stream_id=step_no[0])     # the prior's noise: device Philox, inside the step
        step_no[0] += 1
        p61 = ops.assemble61(recon, pos)
        # shard sizes were verified collectively before the loop; strong scaling = shard_range rows per rank by construction
        gathered = dist.all_gather_rows(p61, total_rows=B_global, verify=False)   # shard sizes: verified once before the warm-up (below)
        return gathered

    # one verified exchange outside every timed region: every rank holds the rows shard_range assigns to it
    dist.all_gather_rows(torch.zeros(B, 1, device=dev), total_rows=B_global, verify=True)

    log(f"rank {rank}/{world}: model and inputs resident ({args.scaling} scaling: rows [{lo}, {hi}) of {B_global}, N={N}, K={K}); warm-up")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    log("timed region (per-launch profiling off)")
    lib.dvq_prof_enable(0)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    dist.barrier()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0, dev)
    log(f"timed region done: {elapsed:.3f} s for {args.steps} steps")
    assert gathered.shape == (B_global, 61) and bool(torch.isfinite(gathered).all())
    import hashlib
    # the last timed step's gathered parameters: identical for every world size (strong scaling + noise keyed by the global row)
    gathered_sha = hashlib.sha256(gathered.cpu().numpy().tobytes()).hexdigest()
    # second pass, NOT part of the headline: every launch bracketed by two HIP events on its stream -> kernel breakdown
    kernels, prof_elapsed = {}, None
    if not args.no_prof and args.prof_steps > 0:
        # The timed steps run the PointNet's exact stage and STN FCs on a second stream beside the next launch's trunk kernel; two
        # kernels sharing the chip each take longer than alone, so a per-launch duration measured that way is not the kernel's own.
        # The breakdown pass therefore runs ONE stream (DVQ_PN_STREAMS=0: same kernels, same launches, same results): its durations
        # are those rocprofv3 reports for the same setting (profiles/*_bench_stats_serial_*), and its step is a few ms longer.
        pn_streams_was = os.environ.get("DVQ_PN_STREAMS")
        os.environ["DVQ_PN_STREAMS"] = "0"
        lib.dvq_reload_env()
        lib.dvq_prof_reset()
        lib.dvq_prof_enable(1)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(args.prof_steps):
            step()
        torch.cuda.synchronize(dev)
        prof_elapsed = time.perf_counter() - t1
        lib.dvq_prof_enable(0)
        kernels = prof_read(lib, _lib)
        lib.dvq_prof_reset()
        if pn_streams_was is None:
            del os.environ["DVQ_PN_STREAMS"]
        else:
            os.environ["DVQ_PN_STREAMS"] = pn_streams_was
        lib.dvq_reload_env()

    rows_per_rank = dist.gather_objects([lo, hi])            # every rank's [lo, hi) of the global batch, as each rank computed it
    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B_global * args.steps / elapsed
        out = {"metric": f"grasps/sec at batch={B_global}, N={N} pts, K={K}", "value": value, "unit": "grasps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": DTYPE_NOTE,
               "data": "synthetic", "gathered_sha256": gathered_sha,
               "config": {"workload": f"GenNet.gen full path, global batch {B_global} ({B} grasps on rank 0), N={N} pts, K={K} "
                                      f"codebooks, 15-layer gated PixelCNN prior (cached sampler), device Philox noise inside "
                                      f"the step, synthetic weights",
                          "global_batch": B_global, "points": N, "codebook": K, "parallelism": f"batch-shard x{world}",
                          "allgather_bytes_per_rank": B * 61 * 4,
                          "collective": dist.collective_used,
                          # what the communication layer itself says it spans: ncclCommCount of the C ABI's communicator (None: the
                          # collectives went through torch.distributed), the process group's size, and every rank's row range
                          "rccl_ranks_seen": dist.comm_ranks_seen(dev), "process_group_ranks": world,
                          "rows_per_rank": rows_per_rank}}
        if kernels:
            pe_ms = prof_elapsed * 1e3
            out["roofline"] = step_roofline(kernels, pe_ms, args.prof_steps, ms_per_step, B)
            out["kernels"] = {k: {"count": v["count"], "ms": round(v["ms"], 3),
                                  "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else 0.0}
                              for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}

        out["roofline_vq_argmin"] = vq_microbench(args, lib, _lib, ops, dev, K)
        if not args.no_latency and world == 1:
            out["latency_ms"] = latency_leg(net, synth, dev, K)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity"] = cpu_baseline(sd, arrays, args.cpu_grasps, N, K, net, dev)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            b1 = out["cpu_baseline"]["reference_faithful_b1_loop"].get(f"N={N}", {}).get("grasps_per_s")
            if b1 and "latency_ms" in out:
                out["latency_ms"]["b1_speedup_over_cpu_b1_loop"] = (1e3 / out["latency_ms"][f"N={N}"]["B=1"]) / b1
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.shutdown()


if __name__ == "__main__":
    main()
