"""CPU check of the error bound the filtered PointNet trunk relies on (d-vqvae_amd/csrc/pointnet_filter.hip).

The kernel keeps every point whose fp16 matrix-core score is within 2 E of the best one, with
    E = |r_n| max|d_p| + |w_n| max|rd_p| + C_ID |w_n| max|d_p| + 2 DELTA |w_n| max|h_p|
(r_n, rd_p = the rounding residuals of the fp16 images, measured).  This test restates the arithmetic with numpy / torch on the
CPU -- power-of-two scales, round-to-nearest fp16 images, fp32 accumulation, id bits in the low mantissa -- on random and on
adversarial data and checks  |approx - exact| <= E  for EVERY (point, channel), i.e. that the true maximum can never be dropped.
It does not run the HIP kernels (tests/test_gpu_parity.py::test_pointnet_filter_* do: filtered == exhaustive, bit for bit)."""
import numpy as np
import pytest
import torch

C_ID, DELTA = 5.0e-5, 1.0e-6          # the constants of pointnet_filter.hip


def _pow2_scale(amax):
    """2^k with amax * 2^k in [2^14, 2^15) (the kernel's exponent arithmetic); 1 for degenerate input."""
    out = np.ones_like(amax, dtype=np.float32)
    ok = (amax > 2.0 ** -106) & (amax < 2.0 ** 108)
    e = np.floor(np.log2(amax[ok].astype(np.float64)))
    out[ok] = (2.0 ** (14 - e)).astype(np.float32)
    return out


def _filter_scores(h, w, c, rng):
    """h [P,128] rows, w [N,128] weights, c [128] centre -> (approx [P,N] in real units, E [N], exact [P,N] in fp64)."""
    d = (h - c[None, :]).astype(np.float32)
    dn = np.sqrt((d.astype(np.float64) ** 2).sum(1))
    s = _pow2_scale(np.array([dn.max() * 1.0001], dtype=np.float32))[0]           # one wave: one scale (largest row norm)
    d16 = torch.from_numpy(d * s).to(torch.float16)                                # round to nearest even
    rd = np.sqrt((((d * s).astype(np.float64) - d16.double().numpy()) ** 2).sum(1)) / s
    t = _pow2_scale(np.abs(w).max(1))
    w16 = torch.from_numpy(w * t[:, None]).to(torch.float16)
    rn = np.sqrt((((w * t[:, None]).astype(np.float64) - w16.double().numpy()) ** 2).sum(1)) / t
    acc = (d16.float() @ w16.float().t()).numpy()                                  # fp32 accumulation of exact products
    bits = acc.view(np.uint32).copy()
    bits = (bits & np.uint32(0xFFFFFF00)) | rng.integers(0, 256, size=bits.shape, dtype=np.uint32)   # 8 id bits
    approx = bits.view(np.float32).astype(np.float64) / (float(s) * t[None, :].astype(np.float64))
    wn = np.sqrt((w.astype(np.float64) ** 2).sum(1)) * 1.00001
    hm = (dn.max() + np.sqrt((c.astype(np.float64) ** 2).sum())) * 1.0001
    E = rn * 1.00001 * dn.max() * 1.00001 + wn * rd.max() * 1.00001 + C_ID * wn * dn.max() + 2 * DELTA * wn * hm
    exact = d.astype(np.float64) @ w.astype(np.float64).T                          # w . (h - c): the centre term is common
    return approx, E, exact


@pytest.mark.parametrize("case", ["random", "relu_sparse", "tiny_spread", "huge_range", "one_hot_weights"])
def test_filter_score_error_is_within_the_bound(case):
    rng = np.random.default_rng(7)
    P, N = 64, 256
    h = np.maximum(rng.standard_normal((P, 128)), 0).astype(np.float32) * 3.0
    w = (rng.standard_normal((N, 128)) * 0.1).astype(np.float32)
    if case == "relu_sparse":
        h *= (rng.random((P, 128)) < 0.2)
    elif case == "tiny_spread":                      # rows nearly identical: the centred rows are 1e-4 of the rows
        h = (h[:1] + 1e-4 * rng.standard_normal((P, 128))).astype(np.float32)
    elif case == "huge_range":                       # 1e6 of dynamic range inside rows and weights (fp16 subnormals in the images)
        h *= np.exp(rng.uniform(-7, 7, (1, 128))).astype(np.float32)
        w *= np.exp(rng.uniform(-7, 7, (N, 1))).astype(np.float32) * np.exp(rng.uniform(-5, 5, (1, 128))).astype(np.float32)
    elif case == "one_hot_weights":
        w = np.zeros((N, 128), np.float32)
        w[np.arange(N), rng.integers(0, 128, N)] = rng.standard_normal(N).astype(np.float32)
    c = h[[0, P // 4, P // 2, 3 * P // 4]].mean(0).astype(np.float32)
    approx, E, exact = _filter_scores(h, w, c, rng)
    err = np.abs(approx - exact)
    worst = (err / np.maximum(E[None, :], 1e-300)).max()
    assert worst <= 1.0, f"{case}: |approx - exact| reaches {worst:.3f} x the bound"
    # and the consequence the kernel uses: the point with the largest exact score is within 2 E of the largest approximate one
    best = exact.argmax(0)
    assert (approx[best, np.arange(N)] >= approx.max(0) - 2 * E).all()
    assert worst > 1e-3, "the bound should not be vacuous (within 1000x of the observed error)"
