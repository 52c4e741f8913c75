"""CPU restatement (numpy, fp32) of the contact / penetration proxies -- TEST INFRASTRUCTURE ONLY (tests/, smoke(),
bench cpu_baseline); the product never imports it.

Follows the reference's utils/utils_loss.py:7-45 (get_NN, get_interior) and utils/loss.py:154-160.  The arithmetic
behind them lives in pytorch3d (ops.knn_points K=1: brute-force squared L2 nearest neighbour; Meshes
.verts_normals_packed: area-weighted face normals summed onto vertices, normalised with eps 1e-6), a third-party
dependency that is ABSENT from /root/reference and from this image (no version is pinned by the reference: it has no
requirements file).  PARITY UNPINNED: no reference output exists to check this file against; it restates the published
algorithms with the operation order fixed as documented in include/dvq.h.
"""
import numpy as np

f32 = np.float32


def _fma(a, b, c):
    """fp32 fused multiply-add, emulated exactly in float64 (products of two fp32 values are exact in fp64; the sum of
    an exact product and an fp32 value rounds once to fp64 first -- double rounding can differ from a true fp32 fma only
    when the fp64 result sits exactly on an fp32 rounding boundary, which the 53-bit intermediate makes vanishingly
    rare; the tests use data where it does not occur and compare bit for bit)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def nn_points(src, trg):
    """src [B,N1,3], trg [B,N2,3] fp32 -> (dist2 [B,N1] fp32, idx [B,N1] int64); first minimum, NaN first."""
    src, trg = np.asarray(src, f32), np.asarray(trg, f32)
    B, N1 = src.shape[:2]
    dist = np.empty((B, N1), f32)
    idx = np.empty((B, N1), np.int64)
    for b in range(B):
        dx = src[b, :, None, 0] - trg[b, None, :, 0]
        dy = src[b, :, None, 1] - trg[b, None, :, 1]
        dz = src[b, :, None, 2] - trg[b, None, :, 2]
        d = _fma(dz, dz, _fma(dy, dy, (dx * dx).astype(f32)))
        nan = np.isnan(d)
        key = np.where(nan, -np.inf, d)                    # NaN beats everything; argmin returns the first
        i = np.argmin(key, axis=1)
        idx[b] = i
        dist[b] = d[np.arange(N1), i]
    return dist, idx


def vertex_normals(verts, faces):
    """verts [B,V,3] fp32, faces [F,3] -> unit normals [B,V,3]: per vertex the sum, in ascending face order, of
    cross(v1 - v0, v2 - v0) over its incident faces; / max(|n|, 1e-6)."""
    verts = np.asarray(verts, f32)
    faces = np.asarray(faces, np.int64)
    B, V = verts.shape[:2]
    out = np.zeros((B, V, 3), f32)
    a = verts[:, faces[:, 1]] - verts[:, faces[:, 0]]
    b = verts[:, faces[:, 2]] - verts[:, faces[:, 0]]
    fn = np.empty_like(a)
    fn[..., 0] = (a[..., 1] * b[..., 2]).astype(f32) - (a[..., 2] * b[..., 1]).astype(f32)
    fn[..., 1] = (a[..., 2] * b[..., 0]).astype(f32) - (a[..., 0] * b[..., 2]).astype(f32)
    fn[..., 2] = (a[..., 0] * b[..., 1]).astype(f32) - (a[..., 1] * b[..., 0]).astype(f32)
    for f in range(faces.shape[0]):                        # ascending face order per vertex
        for c in range(3):
            out[:, faces[f, c]] = (out[:, faces[f, c]] + fn[:, f]).astype(f32)
    nx, ny, nz = out[..., 0], out[..., 1], out[..., 2]
    ln = np.sqrt(_fma(nz, nz, _fma(ny, ny, (nx * nx).astype(f32)))).astype(f32)
    inv = (f32(1.0) / np.maximum(ln, f32(1e-6))).astype(f32)
    return (out * inv[..., None]).astype(f32)


def interior(normals, hand, obj, nn_idx):
    normals, hand, obj = np.asarray(normals, f32), np.asarray(hand, f32), np.asarray(obj, f32)
    B = hand.shape[0]
    bi = np.arange(B)[:, None]
    v = hand[bi, nn_idx] - obj
    n = normals[bi, nn_idx]
    dot = _fma(v[..., 2], n[..., 2], _fma(v[..., 1], n[..., 1], (v[..., 0] * n[..., 0]).astype(f32)))
    return dot > 0
