"""Tensor-level wrappers over the C ABI (include/dvq.h).

PyTorch is used for device memory and streams only: every op validates its tensors (device, dtype,
contiguity -> RuntimeError, the reference's error convention), allocates outputs/scratch through the
torch caching allocator and enqueues the HIP work on torch's *current* stream.  No host sync inside.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check

Tensor = torch.Tensor


def _require_gpu(*ts: Tensor) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("dvq ops need tensors on a HIP device (there is no CPU fallback); got " + str(t.device))
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {t.device} vs {dev}")
    return dev


def _f32(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    return t


def _i64(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.int64:
        raise RuntimeError(f"{name}: expected int64, got {t.dtype}")
    return t


def _rows(t: Tensor, name: str) -> Tuple[int, int]:
    """(pointer, row stride) of a 2-D fp32 view whose rows are contiguous."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"{name}: expected a 2-D tensor with contiguous rows, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1]))


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


_ws: Dict[Tuple[int, int, str], Tensor] = {}
_WS_MAX = 16        # (device, stream) scratch buffers kept; the least recently used one goes when a 17th stream shows up


def workspace(nbytes: int, dev: torch.device, tag: str = "main") -> Tensor:
    """Grow-only scratch per (device, current stream): reuse is ordered by the stream the ops are enqueued on, so two
    streams driving the library concurrently never share a buffer.  At most ``_WS_MAX`` buffers are kept (LRU; an evicted
    buffer stays alive until the work already enqueued on it has run: the caching allocator frees by stream order)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, int(torch.cuda.current_stream(idx).cuda_stream), tag)
    buf = _ws.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = None
        buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=dev)
    _ws[key] = buf                                           # (re)inserted last = most recently used
    while len(_ws) > _WS_MAX:
        _ws.pop(next(iter(_ws)))
    return buf


def release_workspaces() -> None:
    _ws.clear()


def new_err_flag(dev: torch.device) -> Tensor:
    return torch.zeros(1, dtype=torch.int32, device=dev)


# ------------------------------------------------------------------------------------------ fp16 range
# The default GEMM arithmetic carries activations as two fp16 pieces: a value beyond fp16's range (|x| >= 65 520) turns its output
# ROW into NaN -- never a silently wrong number (csrc/gemm_f16x2.hip).  Every entry point that multiplies on those images looks at
# its result once (one host synchronisation) and runs the call again on the six-product bf16 images, which have fp32's range, when
# it holds a non-finite value; inputs that are themselves NaN / Inf come out non-finite there too, as in the reference.
# GenNet.gen makes ONE such check for the whole path and switches the per-op checks off around its inner calls.
import contextlib
import threading

_range_state = threading.local()                            # per host thread: one thread's gen() must not switch another's checks off


class no_range_check(contextlib.ContextDecorator):
    """``with ops.no_range_check(): ...`` (or ``@ops.no_range_check()`` on a function) -- the caller checks the final result itself
    (GenNet.gen: one synchronisation per call)."""

    def __enter__(self):
        _range_state.off = getattr(_range_state, "off", 0) + 1
        return self

    def __exit__(self, *exc):
        _range_state.off -= 1
        return False


def _out_of_range(kind, *outs: Tensor) -> bool:
    """True when a result of the fp16-image arithmetic holds a non-finite value and the per-op check is on (synchronises)."""
    if getattr(_range_state, "off", 0) or kind != _lib.PLANES_F16X2:
        return False
    return not all(bool(torch.isfinite(o).all()) for o in outs if o is not None)


# ------------------------------------------------------------------------------------------ dense
def _planes_args(pl, w: Tensor, what: str):
    """(kind, planes pointer, plane stride, scale pointer, scale tensor) of a weight image handed to an op: a packing.Planes
    or, as before, the bare int16 [3,N,K] tensor of packing.split_bf16x3."""
    from . import packing
    if isinstance(pl, Tensor):
        pl = packing.Planes(_lib.PLANES_BF16X3, pl)
    n_pl = 2 if pl.kind == _lib.PLANES_F16X2 else 3
    t = pl.planes
    if t.dtype != torch.int16 or tuple(t.shape) != (n_pl,) + tuple(w.shape) or not t.is_contiguous() or not w.is_contiguous():
        raise RuntimeError(f"{what}: planes must be the contiguous int16 [{n_pl},N,K] image of a contiguous weight")
    if pl.kind == _lib.PLANES_F16X2 and (pl.scale is None or pl.scale.numel() != w.shape[0] or pl.scale.dtype != torch.float32):
        raise RuntimeError(f"{what}: fp16 planes need their [N] row scales")
    return pl.kind, t.data_ptr(), w.numel(), (pl.scale.data_ptr() if pl.scale is not None else None), pl.scale


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, relu: bool = False,
           out: Optional[Tensor] = None, planes=None) -> Tensor:
    """y = act(x @ weight^T + bias) (dvq_linear).  ``planes`` = packing.split_planes(weight): the pre-split weight image;
    without it the image is built on the spot (two small launches), so both forms give the same bits."""
    return linear_multi([(x, weight)], bias, relu, out, planes=[planes] if planes is not None else None)


def linear_multi(pairs: Sequence[Tuple[Tensor, Tensor]], bias: Optional[Tensor] = None, relu: bool = False,
                 out: Optional[Tensor] = None, planes: Optional[Sequence] = None) -> Tensor:
    """y = act(sum_s x_s @ w_s^T + bias).  fp16 images of one call must come from ONE packing.split_f16x2 group (shared row
    scales); ``planes=None`` builds that group here when the default (fp16) arithmetic is selected."""
    from . import packing
    lib = _lib.load()
    dev = _require_gpu(*[t for p in pairs for t in p], bias, out)
    M = pairs[0][0].shape[0]
    N = pairs[0][1].shape[0]
    if planes is None and packing.gemm_kind() == _lib.PLANES_F16X2 and all(w.is_contiguous() and w.dtype == torch.float32 for _, w in pairs):
        planes = packing.split_f16x2([w for _, w in pairs])
    srcs = (_lib.GemmSrc * len(pairs))()
    scale0 = None
    for i, (x, w) in enumerate(pairs):
        _f32(x, "x"), _f32(w, "weight")
        if x.shape[0] != M or w.shape[0] != N or x.shape[1] != w.shape[1]:
            raise RuntimeError(f"linear: shape mismatch x{tuple(x.shape)} w{tuple(w.shape)}")
        px, ldx = _rows(x, "x")
        pw, ldw = _rows(w, "weight")
        kind, wp, wps, sp = 0, None, 0, None
        if planes is not None:
            kind, wp, wps, sp, st = _planes_args(planes[i], w, "linear")
            if i == 0:
                scale0 = st
            elif kind == _lib.PLANES_F16X2 and st is not scale0:
                raise RuntimeError("linear_multi: the fp16 images of one call must share their row scales (packing.split_f16x2 of the group)")
        srcs[i] = _lib.GemmSrc(px, pw, ldx, ldw, x.shape[1], kind, wp, wps, sp)
    if bias is not None:
        _f32(bias, "bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise RuntimeError("linear: bias must be a contiguous [N] tensor")
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    po, ldo = _rows(_f32(out, "out"), "out")
    if out.shape[0] != M or out.shape[1] != N:
        raise RuntimeError("linear: bad output shape")
    with torch.cuda.device(dev):
        check(lib.dvq_linear(srcs, len(pairs), M, N, bias.data_ptr() if bias is not None else None,
                             1 if relu else 0, po, ldo, _stream(dev)), "dvq_linear")
    if planes is not None and _out_of_range(srcs[0].wp_kind, out):
        with no_range_check():                               # once more on the six-product images (fp32's range)
            return linear_multi(pairs, bias, relu, out, planes=[packing.split_planes(w.contiguous(), _lib.PLANES_BF16X3) for _, w in pairs])
    return out


def mlp3(x: Tensor, layers, out: Optional[Tensor] = None) -> Tensor:
    """Linear + ReLU, Linear + ReLU, Linear (dvq_mlp3: Decoder / Encoder of network/DVQVAE.py).
    ``layers`` = three (weight [n_out, k_in], bias or None, planes (packing.split_planes) or None) tuples."""
    lib = _lib.load()
    dev = _require_gpu(x, out, *[t for l in layers for t in l[:2]])
    if len(layers) != 3:
        raise RuntimeError("mlp3: exactly three layers")
    px, ldx = _rows(_f32(x, "x"), "x")
    M = x.shape[0]
    arr = (_lib.MlpLayer * 3)()
    k = x.shape[1]
    for i, (w, b, pl) in enumerate(layers):
        _f32(w, "weight")
        if not w.is_contiguous() or w.shape[1] != k or (b is not None and (b.numel() != w.shape[0] or not b.is_contiguous())):
            raise RuntimeError(f"mlp3: layer {i} has weight {tuple(w.shape)} for {k} inputs")
        kind, wp, sp = 0, None, None
        if pl is not None:
            kind, wp, _, sp, _ = _planes_args(pl, w, "mlp3")
        arr[i] = _lib.MlpLayer(w.data_ptr(), b.data_ptr() if b is not None else None, wp, w.shape[0], w.shape[1], sp, kind, 0)
        k = w.shape[0]
    if out is None:
        out = torch.empty(M, k, dtype=torch.float32, device=dev)
    po, ldo = _rows(_f32(out, "out"), "out")
    if tuple(out.shape) != (M, k):
        raise RuntimeError("mlp3: bad output shape")
    nws = lib.dvq_mlp3_workspace_bytes(M, layers[0][0].shape[0], layers[1][0].shape[0])
    ws = workspace(nws, dev, tag="mlp3")
    with torch.cuda.device(dev):
        check(lib.dvq_mlp3(px, ldx, M, arr, po, ldo, ws.data_ptr(), ws.numel(), _stream(dev)), "dvq_mlp3")
    if any(l[2] is not None and not isinstance(l[2], Tensor) and l[2].kind == _lib.PLANES_F16X2 for l in layers) and _out_of_range(_lib.PLANES_F16X2, out):
        from . import packing
        with no_range_check():                               # once more on the six-product images (fp32's range)
            return mlp3(x, [(w, b, packing.split_planes(w, _lib.PLANES_BF16X3)) for w, b, _ in layers], out=out)
    return out


# ------------------------------------------------------------------------------------------ VQ
def vq_fast_supported(K: int, D: int) -> bool:
    return bool(_lib.load().dvq_vq_fast_supported(K, D))


def vq_pack(E: Tensor) -> Tensor:
    """Packed image of a codebook for the fast path (fp16 image of -2 sE E, canonical |e_k|^2, max |e_k|, measured rounding error).
    Build it once per codebook state and pass it to ``vq_argmin(..., packed=)``; it is NOT cached here because a
    raw pointer cannot tell a recycled allocation from the same codebook."""
    lib = _lib.load()
    _require_gpu(E)
    K, D = E.shape
    nbytes = lib.dvq_vq_pack_bytes(K, D)
    if nbytes == 0:
        raise RuntimeError(f"vq_pack: no fast path for K={K}, D={D}")
    if not E.is_contiguous():
        raise RuntimeError("vq_pack: codebook must be contiguous")
    packed = torch.empty(nbytes, dtype=torch.uint8, device=E.device)
    with torch.cuda.device(E.device):
        check(lib.dvq_vq_pack(_f32(E, "E").data_ptr(), K, D, packed.data_ptr(), nbytes, _stream(E.device)), "dvq_vq_pack")
    return packed


def vq_argmin(z: Tensor, E: Tensor, return_dist: bool = False, fast: Optional[bool] = None,
              packed: Optional[Tensor] = None, slow_rows: Optional[Tensor] = None):
    """idx[m] = argmin_k (|z_m|^2 + |E_k|^2) - 2 z_m.E_k in the canonical fp32 order -> int64 [M].
    K<=512 (a multiple of 32)/D=256 on dense rows takes the fast kernel (fp16-MFMA filter + exact refine in one launch; same indices, bit for bit); everything else
    (and ``return_dist``) the exact fp32-MFMA kernel.  ``fast`` forces the choice; ``packed`` = vq_pack(E) skips the
    per-call codebook packing (three tiny kernels); ``slow_rows`` (int64 [1] on the device, accumulated, never reset)
    counts the rows the fast kernel could not decide from their candidate lists (see dvq.h)."""
    lib = _lib.load()
    dev = _require_gpu(z, E)
    _f32(z, "z"), _f32(E, "E")
    if not E.is_contiguous():
        raise RuntimeError("vq_argmin: codebook must be contiguous")
    pz, ldz = _rows(z, "z")
    M, D = z.shape
    K = E.shape[0]
    if E.shape[1] != D:
        raise RuntimeError(f"vq_argmin: z has D={D}, codebook has D={E.shape[1]}")
    idx = torch.empty(M, dtype=torch.int64, device=dev)
    can_fast = bool(lib.dvq_vq_fast_supported(K, D)) and not return_dist and (M <= 1 or ldz == D) and pz % 16 == 0
    if fast is None:
        fast = can_fast
    elif fast and not can_fast:
        raise RuntimeError(f"vq_argmin: fast path unavailable for K={K}, D={D}, dense={ldz == D}, return_dist={return_dist}")
    if fast and M > 0:
        if packed is None:
            packed = vq_pack(E)
        nws = lib.dvq_vq_fast_workspace_bytes(M, K, D)
        ws = workspace(nws, dev)
        with torch.cuda.device(dev):
            check(lib.dvq_vq_argmin_fast(pz, E.data_ptr(), packed.data_ptr(), M, K, D, idx.data_ptr(),
                                         _i64(slow_rows, "slow_rows").data_ptr() if slow_rows is not None else None,
                                         ws.data_ptr(), ws.numel(), _stream(dev)), "dvq_vq_argmin_fast")
        return idx
    dmin = torch.empty(M, dtype=torch.float32, device=dev) if return_dist else None
    nws = lib.dvq_vq_argmin_workspace_bytes(M, K)
    ws = workspace(nws, dev)
    with torch.cuda.device(dev):
        check(lib.dvq_vq_argmin(pz, ldz, E.data_ptr(), M, K, D, idx.data_ptr(), dmin.data_ptr() if return_dist else None,
                                ws.data_ptr(), ws.numel(), _stream(dev)), "dvq_vq_argmin")
    return (idx, dmin) if return_dist else idx


def vq_lookup(E: Tensor, idx: Tensor, out: Optional[Tensor] = None, err: Optional[Tensor] = None) -> Tensor:
    """out[m] = E[idx[m]] (row gather == one-hot @ E).  ``idx`` may be a strided 1-D view.
    Out-of-range indices raise RuntimeError unless an ``err`` flag tensor is supplied (then the caller checks it)."""
    lib = _lib.load()
    dev = _require_gpu(E, idx, out, err)
    _f32(E, "E"), _i64(idx, "idx")
    if idx.dim() != 1 or not E.is_contiguous():
        raise RuntimeError("vq_lookup: idx must be 1-D and the codebook contiguous")
    M, (K, D) = idx.shape[0], E.shape
    if out is None:
        out = torch.empty(M, D, dtype=torch.float32, device=dev)
    po, ldo = _rows(_f32(out, "out"), "out")
    own_err = err is None
    if own_err:
        err = new_err_flag(dev)
    with torch.cuda.device(dev):
        check(lib.dvq_vq_lookup(E.data_ptr(), idx.data_ptr(), idx.stride(0) if M > 1 else 1, M, K, D, po, ldo,
                                err.data_ptr(), _stream(dev)), "dvq_vq_lookup")
    if own_err and int(err.item()) != 0:
        raise RuntimeError(f"index out of bounds for codebook with {K} rows")
    return out


# ------------------------------------------------------------------------------------------ sampling noise
def exp1_noise(rows: int, cols: int, seed: int, row0: int = 0, stream_id: int = 0, device=None, out: Optional[Tensor] = None,
               perm: Optional[Tensor] = None) -> Tensor:
    """[rows, cols] Exp(1) variates from the device Philox generator, keyed by (seed, stream_id, GLOBAL row row0 + r, column):
    a shard of a batch (row0 = its first row) draws exactly the rows the whole batch would draw (dvq_exp1_noise).
    ``perm`` (int64 [rows] on the device): row r of the result holds the draws of global row row0 + perm[r]."""
    lib = _lib.load()
    if cols % 4 != 0:                                     # the generator writes 16-byte quads: draw a padded row, keep the columns asked for
        if device is None and out is not None:
            device = out.device
        wide = exp1_noise(rows, (cols + 3) // 4 * 4, seed, row0, stream_id, device=device, perm=perm)
        if out is None:
            return wide[:, :cols].contiguous()
        out.copy_(wide[:, :cols])
        return out
    if out is None:
        if device is None:
            raise RuntimeError("exp1_noise: pass `device` or `out`")
        out = torch.empty(rows, cols, dtype=torch.float32, device=device)
    dev = _require_gpu(out)
    if tuple(out.shape) != (rows, cols) or not out.is_contiguous() or out.dtype != torch.float32:
        raise RuntimeError("exp1_noise: `out` must be a contiguous float32 [rows, cols] tensor")
    if perm is not None and (perm.dtype != torch.int64 or perm.numel() != rows or not perm.is_contiguous() or perm.device != out.device):
        raise RuntimeError("exp1_noise: `perm` must be a contiguous int64 [rows] tensor on the output's device")
    with torch.cuda.device(dev):
        check(lib.dvq_exp1_noise_rows(int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF, int(row0),
                                      perm.data_ptr() if perm is not None else None, rows, cols, out.data_ptr(), _stream(dev)), "dvq_exp1_noise")
    return out


def default_noise_key():
    """(seed, first global row) of noise drawn without an explicit key: the seed follows torch.manual_seed (initial_seed of
    the default generator), and every rank of a process group draws rows of its own (rank * 2^40 + b), so that ranks calling
    gen(obj) on different objects without naming seed / row0 do not all draw the same noise."""
    import torch.distributed as td
    rank = td.get_rank() if td.is_available() and td.is_initialized() else 0
    return int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF, rank << 40


def probe_f16_subnormal(device) -> Tuple[float, float]:
    """(fp16 MFMA product with a subnormal input, its exact value): equal when the matrix core keeps fp16 subnormals."""
    lib = _lib.load()
    out = torch.zeros(2, dtype=torch.float32, device=device)
    dev = _require_gpu(out)
    with torch.cuda.device(dev):
        check(lib.dvq_probe_f16_subnormal(out.data_ptr(), _stream(dev)), "dvq_probe_f16_subnormal")
    a, b = out.cpu().tolist()
    return a, b


# ------------------------------------------------------------------------------------------ PointNet
def pointnet_encode(packed, pc: Tensor, out: Optional[Tensor] = None, want_trans: bool = True):
    """packed: packing.PackedPointNet.  pc [B,C,N] -> (feat [B,1024] (or written into ``out``), trans [B,3,3])."""
    lib = _lib.load()
    dev = _require_gpu(pc, out)
    _f32(pc, "pc")
    if pc.dim() != 3 or not pc.is_contiguous() or pc.shape[1] != packed.C:
        raise RuntimeError(f"pointnet_encode: expected a contiguous [B,{packed.C},N] tensor, got {tuple(pc.shape)}")
    packed.to(dev)
    B, _, N = pc.shape
    if out is None:
        out = torch.empty(B, 1024, dtype=torch.float32, device=dev)
    po, ldo = _rows(_f32(out, "out"), "out")
    if out.shape != (B, 1024):
        raise RuntimeError("pointnet_encode: bad output shape")
    trans = torch.empty(B, 3, 3, dtype=torch.float32, device=dev) if want_trans else None
    nws = lib.dvq_pointnet_workspace_bytes(B, N)
    ws = workspace(nws, dev)
    with torch.cuda.device(dev):
        check(lib.dvq_pointnet_encode(C.byref(packed.cstruct), pc.data_ptr(), B, N, po, ldo,
                                      trans.data_ptr() if want_trans else None, ws.data_ptr(), ws.numel(), _stream(dev)),
              "dvq_pointnet_encode")
    return out, trans


def pointnet_fault_counters(reset: bool = False) -> Tuple[int, int]:
    """(suspect tile records, channels outside their records' interval) the filtered PointNet trunk has counted on the current device
    since the last reset -- its run-time consistency checks (dvq_pointnet_fault_counters).  Both are re-evaluated in full when they
    happen, so the features stay right; anything but (0, 0) means the filter's bookkeeping failed and should be reported."""
    lib = _lib.load()
    out = (C.c_uint64 * 2)()
    check(lib.dvq_pointnet_fault_counters(out, 1 if reset else 0), "dvq_pointnet_fault_counters")
    return int(out[0]), int(out[1])


# ------------------------------------------------------------------------------------------ PixelCNN
def pixelcnn_sample(packed, label: Tensor, noise: Tensor, return_logits: bool = False, err: Optional[Tensor] = None, _retry: bool = False):
    """label [B] int64, noise [B,9,n_in] Exp(1) -> codes [B,3,3] int64 (+ logits [B,9,n_in]).
    With a caller-supplied ``err`` flag nothing synchronises here: a draw from all-NaN logits (fp16 range, bit 2 of the flag) leaves
    -1 at its position of ``codes`` and the caller deals with those rows (GenNet.gen regenerates them on the bf16x3 images)."""
    lib = _lib.load()
    dev = _require_gpu(label, noise, err)
    _i64(label, "label"), _f32(noise, "noise")
    packed.to(dev)
    B = label.shape[0]
    if label.dim() != 1 or not label.is_contiguous():
        raise RuntimeError("pixelcnn_sample: label must be a contiguous [B] tensor")
    if tuple(noise.shape) != (B, 9, packed.n_in) or not noise.is_contiguous():
        raise RuntimeError(f"pixelcnn_sample: noise must be contiguous [B,9,{packed.n_in}], got {tuple(noise.shape)}")
    codes = torch.empty(B, 3, 3, dtype=torch.int64, device=dev)
    logits = torch.empty(B, 9, packed.n_in, dtype=torch.float32, device=dev) if return_logits else None
    own_err = err is None
    if own_err:
        err = new_err_flag(dev)
    nws = lib.dvq_pixelcnn_workspace_bytes(C.byref(packed.cstruct), B)
    ws = workspace(nws, dev)
    with torch.cuda.device(dev):
        check(lib.dvq_pixelcnn_sample(C.byref(packed.cstruct), label.data_ptr(), noise.data_ptr(), B, codes.data_ptr(),
                                      logits.data_ptr() if return_logits else None, err.data_ptr(), ws.data_ptr(),
                                      ws.numel(), _stream(dev)), "dvq_pixelcnn_sample")
    if own_err:
        e = int(err.item())
        if e & 1:
            raise RuntimeError(f"label out of range for the prior's {packed.n_classes} classes")
        if (e & 4) and packed.kind == _lib.PLANES_F16X2 and not _retry:
            # non-finite logits under the fp16 weight images: an activation left fp16's range (or the input is not finite): once
            # more on the six-product bf16 split, which has fp32's range; the images return to the default kind at the next use
            from . import packing
            with packing.gemm_kind_as(_lib.PLANES_BF16X3):
                return pixelcnn_sample(packed, label, noise, return_logits=return_logits, _retry=True)
    return (codes, logits) if return_logits else codes


def pixelcnn_forward(packed, x: Tensor, label: Tensor) -> Tensor:
    """x [B,3,3] int64 tokens, label [B] -> logits [B,n_in,3,3] (GatedPixelCNN.forward)."""
    lib = _lib.load()
    dev = _require_gpu(x, label)
    _i64(x, "x"), _i64(label, "label")
    packed.to(dev)
    B = x.shape[0]
    if tuple(x.shape[1:]) != (3, 3):
        raise RuntimeError("pixelcnn_forward: only the 3x3 latent grid of the grasp path is supported")
    x = x.contiguous()
    label = label.contiguous()
    logits = torch.empty(B, 9, packed.n_in, dtype=torch.float32, device=dev)
    err = new_err_flag(dev)
    nws = lib.dvq_pixelcnn_workspace_bytes(C.byref(packed.cstruct), B)
    ws = workspace(nws, dev)
    with torch.cuda.device(dev):
        check(lib.dvq_pixelcnn_forward(C.byref(packed.cstruct), x.data_ptr(), label.data_ptr(), B, logits.data_ptr(),
                                       err.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)), "dvq_pixelcnn_forward")
    if int(err.item()) & 1:
        raise RuntimeError("index out of range in self (token or class label)")
    if _out_of_range(packed.kind, logits):
        from . import packing
        with no_range_check(), packing.gemm_kind_as(_lib.PLANES_BF16X3):     # once more on the six-product images (fp32's range)
            return pixelcnn_forward(packed, x, label)
    return logits.view(B, 3, 3, packed.n_in).permute(0, 3, 1, 2)


# ------------------------------------------------------------------------------------------ MANO
def mano_forward(packed, betas: Tensor, hand_pose: Tensor, global_orient: Optional[Tensor] = None,
                 transl: Optional[Tensor] = None, channel_major: bool = False, want_joints: bool = False):
    """-> verts [B,778,3] (or [B,3,778] when channel_major), optional joints [B,16,3]."""
    lib = _lib.load()
    dev = _require_gpu(betas, hand_pose, global_orient, transl)
    packed.to(dev)
    B = betas.shape[0]
    pb, ldb = _rows(_f32(betas, "betas"), "betas")
    pp, ldp = _rows(_f32(hand_pose, "hand_pose"), "hand_pose")
    if betas.shape[1] != 10 or hand_pose.shape[1] != 45 or hand_pose.shape[0] != B:
        raise RuntimeError("mano_forward: expected betas [B,10] and hand_pose [B,45]")
    pg = ldg = pt = ldt = 0
    if global_orient is not None:
        pg, ldg = _rows(_f32(global_orient, "global_orient"), "global_orient")
    if transl is not None:
        pt, ldt = _rows(_f32(transl, "transl"), "transl")
    verts = torch.empty((B, 3, 778) if channel_major else (B, 778, 3), dtype=torch.float32, device=dev)
    joints = torch.empty(B, 16, 3, dtype=torch.float32, device=dev) if want_joints else None
    nws = lib.dvq_mano_workspace_bytes(B)
    ws = workspace(nws, dev)
    with torch.cuda.device(dev):
        check(lib.dvq_mano_forward(C.byref(packed.cstruct), pb, ldb, pp, ldp, pg or None, ldg, pt or None, ldt, B,
                                   verts.data_ptr(), 1 if channel_major else 0,
                                   joints.data_ptr() if want_joints else None, ws.data_ptr(), ws.numel(), _stream(dev)),
              "dvq_mano_forward")
    if _out_of_range(getattr(packed, "kind", None), verts, joints):
        from . import packing
        with no_range_check(), packing.gemm_kind_as(_lib.PLANES_BF16X3):     # the blendshape GEMM once more on the six-product images
            return mano_forward(packed, betas, hand_pose, global_orient, transl, channel_major, want_joints)
    return (verts, joints) if want_joints else verts


# ------------------------------------------------------------------------------------------ data movement
def copy_cols(src: Tensor, out: Tensor) -> Tensor:
    lib = _lib.load()
    dev = _require_gpu(src, out)
    ps, lds = _rows(_f32(src, "src"), "src")
    po, ldo = _rows(_f32(out, "out"), "out")
    if src.shape != out.shape:
        raise RuntimeError("copy_cols: shape mismatch")
    with torch.cuda.device(dev):
        check(lib.dvq_copy_cols(ps, lds, src.shape[0], src.shape[1], po, ldo, _stream(dev)), "dvq_copy_cols")
    return out


def assemble61(recon: Tensor, recon_pos: Tensor) -> Tensor:
    lib = _lib.load()
    dev = _require_gpu(recon, recon_pos)
    B = recon.shape[0]
    if tuple(recon.shape) != (B, 55) or tuple(recon_pos.shape) != (B, 6) or not recon.is_contiguous() or not recon_pos.is_contiguous():
        raise RuntimeError("assemble61: expected contiguous recon [B,55] and recon_pos [B,6]")
    out = torch.empty(B, 61, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.dvq_assemble61(_f32(recon, "recon").data_ptr(), _f32(recon_pos, "recon_pos").data_ptr(), B,
                                 out.data_ptr(), _stream(dev)), "dvq_assemble61")
    return out


def transform_cloud(pc: Tensor, R: Tensor, t: Optional[Tensor] = None) -> Tensor:
    """pc [C,N] (one object, broadcast) or [B,C,N]; R [B,3,3]; t [3] -> [B,C,N] with xyz' = R xyz + t."""
    lib = _lib.load()
    dev = _require_gpu(pc, R, t)
    _f32(pc, "pc"), _f32(R, "R")
    B = R.shape[0]
    if not pc.is_contiguous() or not R.is_contiguous() or tuple(R.shape[1:]) != (3, 3):
        raise RuntimeError("transform_cloud: expected contiguous pc and R [B,3,3]")
    if pc.dim() == 2:
        Cc, N = pc.shape
        bstride = 0
    else:
        if pc.shape[0] != B:
            raise RuntimeError("transform_cloud: batch mismatch")
        _, Cc, N = pc.shape
        bstride = Cc * N
    out = torch.empty(B, Cc, N, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.dvq_transform_cloud(pc.data_ptr(), bstride, R.data_ptr(), t.data_ptr() if t is not None else None, B, Cc,
                                      N, out.data_ptr(), _stream(dev)), "dvq_transform_cloud")
    return out


# ---------------------------------------------------------------------------------------- contact / penetration proxies
def _points(x: Tensor, name: str):
    """[B,N,3]-shaped view (any strides: a permuted [B,C,N] cloud is read in place) -> (ptr, batch, point, coord strides)."""
    if x.dim() != 3 or x.shape[2] != 3:
        raise RuntimeError(f"{name}: expected [B,N,3] (got {tuple(x.shape)})")
    _f32(x, name)
    return x.data_ptr(), x.stride(0), x.stride(1), x.stride(2)


def nn_points(src: Tensor, trg: Tensor):
    """Nearest target point per source point (utils_loss.get_NN): src [B,N1,3], trg [B,N2,3] -> (dist2 [B,N1] f32,
    idx [B,N1] int64).  d = fma(dz,dz, fma(dy,dy, dx*dx)); first minimum; NaN first."""
    lib = _lib.load()
    dev = _require_gpu(src, trg)
    ps, sb, sp, sc = _points(src, "src")
    pt, tb, tp, tc = _points(trg, "trg")
    B, N1, N2 = src.shape[0], src.shape[1], trg.shape[1]
    if trg.shape[0] != B:
        raise RuntimeError("nn_points: batch mismatch")
    dist = torch.empty(B, N1, dtype=torch.float32, device=dev)
    idx = torch.empty(B, N1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(lib.dvq_nn_points(ps, sb, sp, sc, pt, tb, tp, tc, B, N1, N2, dist.data_ptr(), idx.data_ptr(), _stream(dev)),
              "dvq_nn_points")
    return dist, idx


def vertex_normals(verts: Tensor, faces: Tensor, vf_off: Tensor, vf_face: Tensor) -> Tensor:
    """Area-weighted unit vertex normals of B meshes with shared topology: verts [B,V,3]; faces [F,3], vf_off [V+1],
    vf_face [3F] int32 on the device (``contact.face_csr``)."""
    lib = _lib.load()
    dev = _require_gpu(verts, faces, vf_off, vf_face)
    _f32(verts, "verts")
    for t, n in ((faces, "faces"), (vf_off, "vf_off"), (vf_face, "vf_face")):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise RuntimeError(f"vertex_normals: {n} must be contiguous int32")
    if verts.dim() != 3 or verts.shape[2] != 3 or not verts.is_contiguous():
        raise RuntimeError("vertex_normals: verts must be contiguous [B,V,3]")
    B, V = verts.shape[0], verts.shape[1]
    if vf_off.numel() != V + 1 or vf_face.numel() != faces.numel():
        raise RuntimeError("vertex_normals: CSR does not match the mesh")
    out = torch.empty_like(verts)
    with torch.cuda.device(dev):
        check(lib.dvq_vertex_normals(verts.data_ptr(), B, V, faces.data_ptr(), vf_off.data_ptr(), vf_face.data_ptr(),
                                     out.data_ptr(), _stream(dev)), "dvq_vertex_normals")
    return out


def interior(normals: Tensor, hand: Tensor, obj: Tensor, nn_idx: Tensor) -> Tensor:
    """utils_loss.get_interior: bool [B,N], True where the object point lies behind its nearest hand vertex's surface."""
    lib = _lib.load()
    dev = _require_gpu(normals, hand, obj, nn_idx)
    _f32(normals, "normals"), _f32(hand, "hand"), _i64(nn_idx, "nn_idx")
    po, ob, op, oc = _points(obj, "obj")
    B, N, V = obj.shape[0], obj.shape[1], hand.shape[1]
    if not (normals.is_contiguous() and hand.is_contiguous() and nn_idx.is_contiguous()):
        raise RuntimeError("interior: normals, hand, nn_idx must be contiguous")
    if tuple(normals.shape) != tuple(hand.shape) or tuple(nn_idx.shape) != (B, N) or hand.shape[0] != B:
        raise RuntimeError("interior: shape mismatch")
    out = torch.empty(B, N, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib.dvq_interior(normals.data_ptr(), hand.data_ptr(), V, po, ob, op, oc, nn_idx.data_ptr(), B, N,
                               out.data_ptr(), _stream(dev)), "dvq_interior")
    return out.bool()
