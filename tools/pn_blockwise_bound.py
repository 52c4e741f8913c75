#!/usr/bin/env python3
"""CPU: would a BLOCKWISE Cauchy-Schwarz bound (four 32-wide k-chunks, per-chunk norms and per-chunk maxima over the points) be tighter
than the whole-vector bound E of the PointNet filter (pointnet_filter.hip; tests/test_pointnet_filter_bound.py restates its terms)?
Prints, for the two residual terms of E, blockwise / whole-vector and observed / whole-vector on the bound test's data."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_pointnet_filter_bound as T

rng = np.random.default_rng(7)
nrm = lambda x, ax: np.sqrt((x.astype(np.float64) ** 2).sum(ax))
for case in ("random", "relu_sparse"):
    P, N = 256, 1024
    h = np.maximum(rng.standard_normal((P, 128)), 0).astype(np.float32) * 3.0
    if case == "relu_sparse":
        h *= (rng.random((P, 128)) < 0.2)
    w = (rng.standard_normal((N, 128)) * 0.1).astype(np.float32)
    c = h[[0, P // 4, P // 2, 3 * P // 4]].mean(0).astype(np.float32)
    d = (h - c[None, :]).astype(np.float32)
    dn = nrm(d, 1)
    s = T._pow2_scale(np.array([dn.max() * 1.0001], dtype=np.float32))[0]
    rdv = ((d * s).astype(np.float64) - torch.from_numpy(d * s).to(torch.float16).double().numpy()) / s
    t = T._pow2_scale(np.abs(w).max(1))
    rv = ((w * t[:, None]).astype(np.float64) - torch.from_numpy(w * t[:, None]).to(torch.float16).double().numpy()) / t[:, None]
    E1, E2 = nrm(rv, 1) * dn.max(), nrm(w, 1) * nrm(rdv, 1).max()
    B1 = sum(nrm(rv[:, k * 32:(k + 1) * 32], 1) * nrm(d[:, k * 32:(k + 1) * 32], 1).max() for k in range(4))
    B2 = sum(nrm(w[:, k * 32:(k + 1) * 32], 1) * nrm(rdv[:, k * 32:(k + 1) * 32], 1).max() for k in range(4))
    o1, o2 = np.abs(d.astype(np.float64) @ rv.T).max(0), np.abs(rdv @ w.astype(np.float64).T).max(0)
    print(f"{case:12s} blockwise / whole-vector: weight-residual term {(B1 / E1).mean():.3f}, row-residual term {(B2 / E2).mean():.3f};"
          f"  observed / whole-vector: {(o1 / E1).mean():.3f}, {(o2 / E2).mean():.3f}")
