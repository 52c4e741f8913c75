"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the committed golden vectors
(produced by the real reference) and size-independent properties.  Indices bit-exact; floats within the
tolerance BASELINE.json states for the path (1e-5 abs on MANO pose/shape)."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import SEED
from util import assert_close, load_synth
from dvqvae_amd import mano as dmano
from dvqvae_amd import ops, packing, synth
from oracle import dvq_oracle as O
from oracle import mano_oracle, vq_canonical

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def gpu(t):
    return t.to(DEV)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(1, 9, 256), (128, 128, 32), (130, 55, 2560), (777, 1024, 512), (4096, 6, 128)])
def test_linear(M, N, K):
    x = synth.synthetic_normal((M, K), 1, f"lin/x/{M}/{K}")
    w = synth.synthetic_normal((N, K), 1, f"lin/w/{N}/{K}", 1.0 / np.sqrt(K))
    b = synth.synthetic_normal((N,), 1, f"lin/b/{N}")
    for relu in (False, True):
        y = ops.linear(gpu(x), gpu(w), gpu(b), relu=relu)
        ref = x.double() @ w.double().t() + b.double()
        if relu:
            ref = ref.clamp_min(0)
        assert_close(y, ref.float(), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("kind", ["f16x2", "bf16x3"])
def test_linear_fuzz_shapes_planes_and_batch_invariance(kind):
    """60 random (M, N, K) per weight image (the fp16 three-product split = the default, and the six-product bf16 split): a
    pre-split image and one built on the fly give the same bits, results within fp32-GEMM accuracy of the fp64 product, and
    every row of a batched call equals the same row computed in a smaller batch (the accumulation order does not depend on
    the M tiling)."""
    from dvqvae_amd import _lib, packing
    k = _lib.PLANES_F16X2 if kind == "f16x2" else _lib.PLANES_BF16X3
    rng = np.random.default_rng(3)
    for it in range(60):
        M = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 300, 1000, 4097]))
        N = int(rng.choice([1, 3, 6, 31, 32, 33, 55, 127, 128, 129, 256, 384, 1000, 1024]))
        K = 32 * int(rng.integers(1, 50))
        x = torch.randn(M, K, device=DEV) * float(10 ** rng.uniform(-2, 2))
        w = torch.randn(N, K, device=DEV) / np.sqrt(K)
        if it % 7 == 3:
            w *= torch.exp2(torch.randint(-30, 30, (N, 1), device=DEV).float())     # row magnitudes 2^-30 .. 2^30
        b = torch.randn(N, device=DEV) if it % 3 else None
        relu = bool(it % 2)
        y = _with_env("DVQ_GEMM", kind, lambda: ops.linear(x, w, b, relu=relu))
        yp = ops.linear(x, w, b, relu=relu, planes=packing.split_planes(w, k))
        assert torch.equal(y, yp), f"case {it}: planes path differs"
        ref = x.double() @ w.double().t() + (b.double() if b is not None else 0)
        if relu:
            ref = ref.clamp_min(0)
        scale = (x.double().abs() @ w.double().abs().t()) + (b.double().abs() if b is not None else 0)
        assert bool(((y.double() - ref).abs() <= 4e-6 * scale.max(dim=0, keepdim=True).values + 1e-30).all()), f"case {it}: M={M} N={N} K={K}"
        lo = int(rng.integers(0, M))
        hi = min(M, lo + int(rng.integers(1, 70)))
        assert torch.equal(ops.linear(x[lo:hi], w, b, relu=relu, planes=packing.split_planes(w, k)), y[lo:hi]), f"case {it}: rows {lo}:{hi} depend on the batch"


def test_linear_f16x2_range_and_small_magnitudes():
    """The fp16 split keeps 22 bits of an activation down to |x| = 2^-14 and an absolute error of 2^-36 below; |x| >= 65 520
    (fp16's range) turns that ROW into NaN -- never a silently wrong number -- and leaves the other rows alone."""
    from dvqvae_amd import _lib, packing
    if os.environ.get("DVQ_GEMM", "").lower() == "fp32":
        pytest.skip("DVQ_GEMM=fp32: the library ignores the weight images")
    torch.manual_seed(17)
    M, N, K = 200, 256, 512
    w = torch.randn(N, K, device=DEV) / np.sqrt(K)
    pl = packing.split_planes(w, _lib.PLANES_F16X2)
    for mag in (1e-3, 1e-4, 3e-6):
        x = torch.randn(M, K, device=DEV) * mag
        y = ops.linear(x, w, None, planes=pl)
        ref = x.double() @ w.double().t()
        scale = float((x.double().abs() @ w.double().abs().t()).max())
        assert float((y.double() - ref).abs().max()) <= 4e-6 * scale + K * 2.0 ** -36 * float(w.abs().max())
    x = torch.randn(M, K, device=DEV)
    y0 = ops.linear(x, w, None, planes=pl)
    x[7, 100] = 7.0e4
    x[9, 3] = -1.0e30
    with ops.no_range_check():                              # the kernels' own behaviour, without the host mirror's check-and-retry
        y = ops.linear(x, w, None, planes=pl)
        assert bool(torch.isnan(y[7]).all()) and bool(torch.isnan(y[9]).all())
        assert bool(torch.isnan(ops.linear(x, w, None, relu=True, planes=pl)[7]).all()), "the ReLU epilogue must not turn the NaN into 0"
    keep = [i for i in range(M) if i not in (7, 9)]
    assert torch.equal(y[keep], y0[keep])
    yb = ops.linear(x, w, None, planes=packing.split_planes(w, _lib.PLANES_BF16X3))       # the six-product split has fp32's range
    assert bool(torch.isfinite(yb).all())
    assert torch.equal(ops.linear(x, w, None, planes=pl), yb), "the checked call runs again on the six-product images"


def test_linear_multi_source_and_strided_views():
    M = 300
    xs = [synth.synthetic_normal((M, k), 2, f"lm/x/{i}") for i, k in enumerate((64, 512, 1024))]
    ws = [synth.synthetic_normal((200, k), 2, f"lm/w/{i}", 0.05) for i, k in enumerate((64, 512, 1024))]
    big = torch.zeros(M, 1600, device=DEV)
    big[:, 0:64], big[:, 64:576], big[:, 576:1600] = gpu(xs[0]), gpu(xs[1]), gpu(xs[2])
    out = torch.full((M, 456), -7.0, device=DEV)
    ops.linear_multi([(big[:, 0:64], gpu(ws[0])), (big[:, 64:576], gpu(ws[1])), (big[:, 576:1600], gpu(ws[2]))],
                     out=out[:, 256:456])
    ref = sum(x.double() @ w.double().t() for x, w in zip(xs, ws)).float()
    assert_close(out[:, 256:456], ref, atol=3e-5)
    assert torch.all(out[:, :256] == -7.0)


def test_linear_rejects_bad_inputs():
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 48, device=DEV), torch.zeros(8, 48, device=DEV))       # K % 32 != 0
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(4, 64, device=DEV), torch.zeros(8, 32, device=DEV))       # K mismatch


@pytest.mark.parametrize("kind", [1, 0])
@pytest.mark.parametrize("sizes", [(2560, 1024, 256, 55), (2048, 1024, 128, 6), (1024, 1024, 512, 256)])
def test_mlp3_equals_three_linears_bitwise(sizes, kind):
    """dvq_mlp3 (Decoder / Encoder as one entry point) = the same three GEMM launches as three dvq_linear calls (both weight images)."""
    ws = [gpu(synth.synthetic_normal((sizes[i + 1], sizes[i]), SEED, f"mlp3/w/{i}", sizes[i] ** -0.5)) for i in range(3)]
    bs = [gpu(synth.synthetic_normal((sizes[i + 1],), SEED, f"mlp3/b/{i}", 0.1)) for i in range(3)]
    pls = [packing.split_planes(w, kind) for w in ws]
    for M in (1, 37, 3000):
        x = gpu(synth.synthetic_normal((M, sizes[0]), SEED, f"mlp3/x/{M}"))
        h = x
        for i in range(3):
            h = ops.linear(h, ws[i], bs[i], relu=i < 2, planes=pls[i])
        assert torch.equal(ops.mlp3(x, [(ws[i], bs[i], pls[i]) for i in range(3)]), h)
        if kind == 0:
            assert torch.equal(ops.mlp3(x, [(ws[i], bs[i], None) for i in range(3)]), h)      # no image: the on-the-fly bf16 split
    assert ops.mlp3(torch.zeros(0, sizes[0], device=DEV), [(ws[i], bs[i], None) for i in range(3)]).shape == (0, sizes[3])
    with pytest.raises(RuntimeError):
        ops.mlp3(torch.zeros(4, sizes[0], device=DEV), [(ws[0], bs[0], None), (ws[2], bs[2], None), (ws[1], bs[1], None)])


# ------------------------------------------------------------------------------------------ VQ
@pytest.mark.parametrize("K,D", [(128, 256), (128, 1024), (512, 256)])
def test_vq_argmin_golden_and_canonical(golden, K, D):
    g = golden("g2_vq")
    E = synth.synthetic_normal((K, D), SEED, f"vq/E/{K}/{D}")
    for M in (1, 7, 4096):
        z = synth.synthetic_normal((M, D), SEED, f"vq/z/{K}/{D}/{M}")
        idx, dmin = ops.vq_argmin(gpu(z), gpu(E), return_dist=True)
        ci, cd = vq_canonical.argmin(z.numpy(), E.numpy())
        assert np.array_equal(idx.cpu().numpy(), ci), "HIP != canonical oracle (indices)"
        assert np.array_equal(dmin.cpu().numpy().view(np.uint32), cd.view(np.uint32)), "HIP != canonical oracle (distance bits)"
        tag = f"K{K}_D{D}_M{M}"
        safe = g[tag + "_gap"] > 1e-3
        assert np.array_equal(idx.cpu().numpy()[safe], g[tag + "_idx"][safe]), "HIP != reference golden"
        print(f"{tag}: match rate vs reference {np.mean(idx.cpu().numpy() == g[tag + '_idx']):.6f}")
    # reference-init regime (tie-prone)
    Eu = synth.synthetic_uniform((K, D), SEED, f"vq/Eu/{K}/{D}", -1.0 / K, 1.0 / K)
    z = synth.synthetic_normal((512, D), SEED, f"vq/zu/{K}/{D}")
    idx = ops.vq_argmin(gpu(z), gpu(Eu)).cpu().numpy()
    ci, _ = vq_canonical.argmin(z.numpy(), Eu.numpy())
    assert np.array_equal(idx, ci)
    safe = g[f"K{K}_D{D}_uinit_gap"] > 1e-3
    assert np.array_equal(idx[safe], g[f"K{K}_D{D}_uinit_idx"][safe])


def test_vq_crafted_ties_nan_inf(golden):
    g = golden("g2_vq")
    E, z = torch.from_numpy(g["crafted_E"]), torch.from_numpy(g["crafted_z"])
    idx = ops.vq_argmin(gpu(z), gpu(E)).cpu().numpy()
    assert idx.tolist() == g["crafted_idx"].tolist()
    ci, _ = vq_canonical.argmin(z.numpy(), E.numpy())
    assert idx.tolist() == ci.tolist()


@pytest.mark.parametrize("K,D,M", [(5, 32, 3), (130, 64, 257), (512, 256, 1000)])
def test_vq_ragged_shapes(K, D, M):
    E = synth.synthetic_normal((K, D), 3, f"vqr/E/{K}")
    z = synth.synthetic_normal((M, D), 3, f"vqr/z/{M}")
    idx, dmin = ops.vq_argmin(gpu(z), gpu(E), return_dist=True)
    ci, cd = vq_canonical.argmin(z.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), ci)
    assert np.array_equal(dmin.cpu().numpy().view(np.uint32), cd.view(np.uint32))


def test_vq_empty_and_lookup():
    E = gpu(synth.synthetic_normal((128, 256), 4, "vql/E"))
    assert ops.vq_argmin(torch.zeros(0, 256, device=DEV), E).shape == (0,)
    idx = torch.tensor([0, 127, 5, 5], device=DEV)
    assert torch.equal(ops.vq_lookup(E, idx), E[idx])
    with pytest.raises(RuntimeError, match="out of bounds"):
        ops.vq_lookup(E, torch.tensor([128], device=DEV))
    with pytest.raises(RuntimeError, match="out of bounds"):
        ops.vq_lookup(E, torch.tensor([-1], device=DEV))


def test_vq_full_size_properties():
    """BASELINE config 2 sizes (M=65536, K=512, D=256): properties that need no oracle pass."""
    K, D, M = 512, 256, 65536
    E = gpu(synth.synthetic_normal((K, D), 5, "vqf/E"))
    z = gpu(synth.synthetic_normal((M, D), 5, "vqf/z"))
    idx, dmin = ops.vq_argmin(z, E, return_dist=True)
    # (1) idempotence: quantising the codebook rows themselves returns their own index
    assert torch.equal(ops.vq_argmin(E, E), torch.arange(K, device=DEV))
    assert torch.equal(ops.vq_argmin(ops.vq_lookup(E, idx), E), idx)
    # (2) permutation equivariance of the codebook
    perm = torch.randperm(K, generator=torch.Generator().manual_seed(0)).to(DEV)
    idx_p = ops.vq_argmin(z, E[perm].contiguous())
    assert torch.equal(perm[idx_p], idx)
    # (3) the winner is no worse than 64 random entries, in fp64
    rnd = torch.randint(0, K, (M, 64), device=DEV, generator=None)
    dw = ((z.double() - E[idx].double()) ** 2).sum(1)
    dr = ((z.double()[:, None, :] - E[rnd].double()) ** 2).sum(-1).min(1)[0]
    assert torch.all(dw <= dr + 1e-3)
    # (4) a sample of rows against the canonical oracle
    rows = torch.arange(0, M, 64)
    ci, _ = vq_canonical.argmin(z[rows].cpu().numpy(), E.cpu().numpy())
    assert np.array_equal(idx[rows].cpu().numpy(), ci)


# ------------------------------------------------------------------------------------------ PointNet
def _pointnet(C, seed):
    from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
    net = PointNetEncoder(channel=C)
    sd = load_synth(net, seed)
    return net.to(DEV), sd


@pytest.mark.parametrize("C,N,B", [(4, 64, 4), (4, 1024, 1), (4, 1024, 4), (4, 3000, 1), (3, 778, 4), (3, 100, 2)])
def test_pointnet_golden(golden, C, N, B):
    g = golden("g1_pointnet")
    net, sd = _pointnet(C, SEED + C)
    x = synth.synthetic_clouds(B, N, seed=100 + N, channels=C)
    feat, trans, none = net(gpu(x))
    assert none is None
    tag = f"C{C}_N{N}_B{B}"
    assert_close(trans, g[tag + "_trans"], atol=TOL, what="trans vs reference golden")
    assert_close(feat, g[tag + "_feat"], atol=TOL, what="feat vs reference golden")
    of, ot = O.pointnet_encode(sd, "", x)
    assert_close(feat, of, atol=TOL)
    assert_close(trans, ot, atol=TOL)


def test_pointnet_batched_equals_loop_and_strided_output():
    net, _ = _pointnet(4, SEED + 4)
    x = gpu(synth.synthetic_clouds(5, 333, seed=9))
    out = torch.zeros(5, 2048, device=DEV)
    net(x, out=out[:, 1024:])
    loop = torch.cat([net(x[b:b + 1])[0] for b in range(5)])
    assert torch.equal(out[:, 1024:], loop), "batched != loop of B=1 (canonical accumulation order should make them identical)"
    assert torch.all(out[:, :1024] == 0)


def _with_env(key, val, fn):
    """Run fn() with a library knob set: the library reads its environment once, dvq_reload_env() re-reads it."""
    import os
    from dvqvae_amd import _lib
    old = os.environ.get(key)
    os.environ[key] = val
    _lib.load().dvq_reload_env()
    try:
        return fn()
    finally:
        if old is None:
            del os.environ[key]
        else:
            os.environ[key] = old
        _lib.load().dvq_reload_env()


@pytest.mark.parametrize("C,N,B", [(4, 1024, 6), (3, 778, 5), (4, 100, 3), (4, 3000, 2), (3, 257, 3), (4, 1, 2),
                                   (4, 769, 5), (3, 800, 6), (4, 801, 3), (3, 288, 7), (4, 2056, 2)])
def test_pointnet_filter_equals_exhaustive_exact_evaluation(C, N, B):
    """The fp16 matrix-core filter of conv3 + max never changes the result: the feature is bit-identical to the maximum of
    the exact fp32 score over ALL points (DVQ_PN_EXHAUSTIVE=1 makes pn_exact_kernel evaluate exactly that), and within the
    path's tolerance of the six-product trunk kernel (DVQ_PN_FILTER=0)."""
    net, _ = _pointnet(C, SEED + 10 * C)
    x = gpu(synth.synthetic_clouds(B, N, seed=300 + N, channels=C))
    # DVQ_PN_FILTER=2: the filtered trunk also where the default prefers the six-product one (tiles less than 3/4 full)
    feat, trans, _ = _with_env("DVQ_PN_FILTER", "2", lambda: net(x))
    feat_all, trans_all, _ = _with_env("DVQ_PN_FILTER", "2", lambda: _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: net(x)))
    assert torch.equal(trans, trans_all), "STN trunk: filtered != exhaustive"
    assert torch.equal(feat, feat_all), "main trunk: filtered != exhaustive"
    feat6, trans6, _ = _with_env("DVQ_PN_FILTER", "0", lambda: net(x))
    assert_close(trans, trans6, atol=TOL, what="filtered trunk vs six-product trunk (trans)")
    assert_close(feat, feat6, atol=TOL, what="filtered trunk vs six-product trunk (feat)")


@pytest.mark.parametrize("C,N,B", [(3, 778, 1030), (4, 769, 9), (3, 800, 6), (4, 257, 5), (3, 2080, 3)])
def test_pointnet_filter_tail_tile_equals_full_tile(C, N, B):
    """A cloud with 1 .. 32 points beyond a multiple of 256 (the 778 MANO vertices) runs its last points as a ONE-block tail tile,
    four samples per workgroup (B = 1030: a last workgroup with two surplus waves), instead of a last full tile of padding
    (DVQ_PN_TAIL=0).  Both are the exact maximum over all points of the same conv2 rows: the features must agree bit for bit, and
    with the exhaustive evaluation."""
    net, _ = _pointnet(C, SEED + 3 * C)
    x = gpu(synth.synthetic_clouds(B, N, seed=900 + N, channels=C))
    # DVQ_PN_FILTER=2: the filtered trunk also where, without the tail tile, the default prefers the six-product one (N = 257)
    feat, trans, _ = _with_env("DVQ_PN_FILTER", "2", lambda: net(x))
    feat0, trans0, _ = _with_env("DVQ_PN_FILTER", "2", lambda: _with_env("DVQ_PN_TAIL", "0", lambda: net(x)))
    assert torch.equal(trans, trans0) and torch.equal(feat, feat0), "tail tile != full tile"
    n = min(B, 64)
    feat_all, trans_all, _ = _with_env("DVQ_PN_FILTER", "2", lambda: _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: net(x[:n].contiguous())))
    assert torch.equal(trans[:n], trans_all) and torch.equal(feat[:n], feat_all), "tail tile != exhaustive"


def test_pointnet_filter_full_machine_repeatability():
    """Regression (round 3): with compiler-formed packed-fp32 instructions pn_trunk_filter_kernel published a wrong
    top-three record about once per 1e6 (tile, channel) pairs when two workgroups shared a CU -- one feature of one cloud in
    ~4 % of 65 536-cloud calls (csrc/Makefile).  Two clouds that hit it within a few calls, 4 096 copies each
    (every CU busy with two workgroups): every copy must give the bits of the exhaustive evaluation, six calls in a row."""
    net, _ = _gennet()
    clouds = gpu(synth.synthetic_clouds(2048, 1024, seed=91))
    for enc in (net.obj_encoder_pos, net.obj_encoder_type):
        for sample in (1436, 1197):
            x = clouds[sample:sample + 1].repeat(4096, 1, 1).contiguous()
            want = _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: enc(x[:1].contiguous()))[0]
            for call in range(6):
                feat = enc(x)[0]
                bad = int((feat != want).any(1).sum())
                assert bad == 0, f"cloud {sample}, call {call}: {bad} of 4096 copies differ from the exhaustive evaluation"


def test_pointnet_filter_fresh_clouds_full_machine_stress():
    """Fence for the same fault on clouds nobody has seen: every run draws FRESH clouds (seed from the OS, printed on failure),
    4 096 distinct clouds x 16 copies = 65 536 per call (every CU busy with two workgroups), two encoders x three calls,
    each compared bit for bit with the exhaustive evaluation (DVQ_PN_EXHAUSTIVE=1: exact_dot of every point)."""
    net, _ = _gennet()
    seed = int.from_bytes(os.urandom(4), "little")
    blk = gpu(synth.synthetic_clouds(4096, 1024, seed=seed))
    big = blk.repeat(16, 1, 1).contiguous()
    for name, enc in (("obj_encoder_pos", net.obj_encoder_pos), ("obj_encoder_type", net.obj_encoder_type)):
        want = _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: enc(blk)[0]).repeat(16, 1)
        for call in range(3):
            feat = enc(big)[0]
            d = (feat != want).any(1)
            assert not bool(d.any()), (f"seed {seed}, {name}, call {call}: clouds {d.nonzero().flatten().tolist()[:5]} of 65 536 differ "
                                       f"from the exhaustive evaluation")
    # the hand encoder's shape: 778 points = three dealt tiles + the one-block tail tile (four samples per workgroup)
    blk = gpu(synth.synthetic_clouds(4096, 778, seed=seed + 1, channels=3))
    big = blk.repeat(16, 1, 1).contiguous()
    want = _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: net.recon_encoder(blk)[0]).repeat(16, 1)
    for call in range(3):
        feat = net.recon_encoder(big)[0]
        d = (feat != want).any(1)
        assert not bool(d.any()), (f"seed {seed}, recon_encoder (N = 778), call {call}: clouds {d.nonzero().flatten().tolist()[:5]} of 65 536 "
                                   f"differ from the exhaustive evaluation")


@pytest.mark.parametrize("N", [1500, 1290])
def test_pointnet_filter_list_overflow_paths(N):
    """pn_exact_kernel's candidate lists are finite; with the capacities shrunk (DVQ_PN_CAPS) the overflow paths -- a wave
    evaluated instead of a dropped pair, a thread walking a wave by itself -- must give the same bits (N = 1290: with a tail tile)."""
    net, _ = _pointnet(4, SEED + 5)
    x = gpu(synth.synthetic_clouds(3, N, seed=11, channels=4))
    feat, trans, _ = net(x)
    for caps in ("7,2048", "4096,3", "0,0", "100,1"):
        f2, t2, _ = _with_env("DVQ_PN_CAPS", caps, lambda: net(x))
        assert torch.equal(trans, t2) and torch.equal(feat, f2), f"caps {caps}"


@pytest.mark.parametrize("N", [700, 780])
def test_pointnet_filter_ties_scales_and_degenerate_clouds(N):
    """Duplicate points (exact ties: all three tracked scores equal -> whole-cloud evaluation), clouds scaled by 1e-4 and
    1e4 (per-wave power-of-two scaling), an all-zero cloud, one huge outlier point.  N = 780: three dealt tiles + a tail tile."""
    net, _ = _pointnet(4, SEED + 77)
    base = synth.synthetic_clouds(4, N, seed=5, channels=4)
    cases = {"dup": base[:, :, torch.randint(0, 40, (N,), generator=torch.Generator().manual_seed(1))],
             "tiny": base * 1e-4, "huge": base * 1e4, "zero": torch.zeros_like(base)}
    out = base.clone()
    out[:, :3, 13] = 5e3
    cases["outlier"] = out
    for name, x in cases.items():
        x = gpu(x.contiguous())
        feat, trans, _ = _with_env("DVQ_PN_FILTER", "2", lambda: net(x))
        feat_all, trans_all, _ = _with_env("DVQ_PN_FILTER", "2", lambda: _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: net(x)))
        assert torch.isfinite(feat).all(), name
        assert torch.equal(trans, trans_all) and torch.equal(feat, feat_all), f"{name}: filtered != exhaustive"
        feat6, _, _ = _with_env("DVQ_PN_FILTER", "0", lambda: net(x))
        scale = float(feat6.abs().max()) + 1.0
        assert_close(feat / scale, feat6 / scale, atol=TOL, what=f"{name}: filtered vs six-product trunk")


def test_pointnet_needs_eval_and_supported_config():
    from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
    net = PointNetEncoder(channel=4).to(DEV).train()
    with pytest.raises(RuntimeError, match="eval"):
        net(torch.zeros(1, 4, 16, device=DEV))
    with pytest.raises(NotImplementedError):
        PointNetEncoder(feature_transform=True)


# ------------------------------------------------------------------------------------------ PixelCNN
def _prior(cfg, seed):
    from dvqvae_amd.network.pixelcnn.models import GatedPixelCNN
    net = GatedPixelCNN(*cfg)
    sd = load_synth(net, seed)
    return net.to(DEV), sd


def test_pixelcnn_small_golden(golden):
    g = golden("g4_pixelcnn")
    net, sd = _prior((32, 64, 3, 16), SEED + 1)
    x, lab = torch.from_numpy(g["small_x"]), torch.from_numpy(g["small_label"])
    logits = net(gpu(x), gpu(lab))
    assert tuple(logits.shape) == (5, 32, 3, 3)
    assert_close(logits, g["small_logits"], atol=TOL)
    codes = net.generate(None, gpu(lab), shape=(3, 3), batch_size=5, noise=gpu(synth.exp1_noise(5, 9, 32, seed=5)))
    assert np.array_equal(codes.cpu().numpy(), g["small_codes"])


def test_pixelcnn_full_golden(golden):
    g = golden("g4_pixelcnn")
    net, sd = _prior((512, 512, 15, 128), SEED + 2)
    x, lab = torch.from_numpy(g["full_x"]), torch.from_numpy(g["full_label"])
    assert_close(net(gpu(x), gpu(lab)), g["full_logits"], atol=2e-5)
    labs = torch.from_numpy(g["full_gen_label"])
    q = synth.exp1_noise(4, 9, 512, seed=6)
    codes, logits = net.generate(None, gpu(labs), batch_size=4, noise=gpu(q), return_logits=True)   # one batched call
    assert np.array_equal(codes.cpu().numpy(), g["full_codes"]), "batched sampling != 4 reference B=1 calls"
    # the logits every draw was made from must equal a teacher-forced forward on the drawn codes
    tf = net(codes, gpu(labs)).permute(0, 2, 3, 1).reshape(4, 9, 512)
    assert torch.equal(tf, logits)


@pytest.mark.parametrize("cfg,B", [((32, 64, 3, 16), 40), ((512, 512, 15, 128), 300), ((512, 512, 15, 128), 700),
                                   ((64, 128, 4, 8), 16384 + 37)])          # two chunks: 16 384 rows and a small-batch one
def test_pixelcnn_class_tables_equal_per_row_evaluation(cfg, B):
    """Row 0 of the grid sees nothing above it and position (0, 0) nothing before it: their activations depend on the class label
    only.  Batches of at least two rows per class evaluate them once per class and read them through the labels
    (GemmSrc::arow, the draw kernel's row index); DVQ_PIXELCNN_TABLES=0 evaluates them per row.  Same kernels per row: the
    teacher-forced logits, the sampled codes and the logits they were drawn from must agree bit for bit (B = 40: the
    small-batch kernels read through the index, 300 / 700: the tiled ones, two and six row tiles)."""
    net, _ = _prior(cfg, SEED + 21)
    n_tok, n_cls = cfg[0], cfg[3]
    g = torch.Generator().manual_seed(B)
    x = gpu(torch.randint(0, n_tok, (B, 3, 3), generator=g))
    lab = gpu(torch.randint(0, n_cls, (B,), generator=g))
    q = gpu(synth.exp1_noise(B, 9, n_tok, seed=B))

    def run():
        codes, logits = net.generate(None, lab, batch_size=B, noise=q, return_logits=True)
        return net(x, lab), codes, logits
    a = run()
    b = _with_env("DVQ_PIXELCNN_TABLES", "0", run)
    for name, u, v in zip(("teacher-forced logits", "sampled codes", "sampling logits"), a, b):
        assert torch.equal(u, v), f"{name}: class tables != per-row evaluation"
    # the packer built the tables once for the weights (include/dvq.h: dvq_pixelcnn_build_tables); without them in the struct a
    # batch of at least two rows per class builds them per call in its workspace: the same bits again
    import ctypes
    from dvqvae_amd import _lib
    pk = net.packed()
    assert pk.cstruct.class_tables, "the packed prior carries no class tables"
    lib = _lib.load()
    ws_pre = lib.dvq_pixelcnn_workspace_bytes(ctypes.byref(pk.cstruct), B)
    keep = pk.cstruct.class_tables
    pk.cstruct.class_tables = None
    try:
        c = run()
        ws_call = lib.dvq_pixelcnn_workspace_bytes(ctypes.byref(pk.cstruct), B)
    finally:
        pk.cstruct.class_tables = keep
    for name, u, v in zip(("teacher-forced logits", "sampled codes", "sampling logits"), a, c):
        assert torch.equal(u, v), f"{name}: tables of the packer != tables built per call"
    assert ws_call > ws_pre, "per-call tables live in the workspace"


def test_pixelcnn_class_tables_serve_single_calls():
    """With the packer's tables a B = 1 call reads row 0's vertical stack and position (0, 0) instead of computing them: it
    must still equal the per-row evaluation, and a batch must equal its rows one by one."""
    net, _ = _prior((512, 512, 15, 128), SEED + 22)
    g = torch.Generator().manual_seed(3)
    lab = gpu(torch.randint(0, 128, (6,), generator=g))
    q = gpu(synth.exp1_noise(6, 9, 512, seed=33))
    codes, logits = net.generate(None, lab, batch_size=6, noise=q, return_logits=True)
    for i in range(6):
        ci, li = net.generate(None, lab[i:i + 1], batch_size=1, noise=q[i:i + 1].contiguous(), return_logits=True)
        assert torch.equal(ci, codes[i:i + 1]) and torch.equal(li, logits[i:i + 1]), f"row {i}: B = 1 call != its row of the batch"
    c0, l0 = _with_env("DVQ_PIXELCNN_TABLES", "0", lambda: net.generate(None, lab, batch_size=6, noise=q, return_logits=True))
    assert torch.equal(c0, codes) and torch.equal(l0, logits), "class tables != per-row evaluation at small batches"


def test_pixelcnn_label_out_of_range():
    net, _ = _prior((32, 64, 3, 16), SEED + 1)
    with pytest.raises(RuntimeError, match="out of range"):
        net.generate(None, torch.tensor([16], device=DEV), batch_size=1)


# ------------------------------------------------------------------------------------------ decoders
def test_decoders_golden(golden):
    from dvqvae_amd.network.DVQVAE import Decoder
    g = golden("g6_decoder")
    for tag, sizes, lat in [("dec", [1024, 256, 55], 2560), ("pos", [1024, 128, 6], 2048)]:
        dec = Decoder(layer_sizes=sizes, latent_size=lat)
        load_synth(dec, SEED + 3)
        z = synth.synthetic_normal((5, lat), SEED, f"dec/z/{tag}")
        assert_close(dec.to(DEV)(gpu(z)), g[tag + "_y"], atol=TOL)


# ------------------------------------------------------------------------------------------ MANO
def test_mano_vs_oracle():
    arrays = dmano.synthetic_mano_arrays()
    ref_arrays = mano_oracle.synthetic_mano_arrays()
    for k in arrays:
        assert np.array_equal(arrays[k], ref_arrays[k]), k
    layer = dmano.ManoLayer(arrays).to(DEV)
    oracle = mano_oracle.ManoOracle(ref_arrays)
    B = 37
    betas = synth.synthetic_normal((B, 10), 6, "mano/betas")
    pose = synth.synthetic_normal((B, 45), 6, "mano/pose", 0.8)
    go = synth.synthetic_normal((B, 3), 6, "mano/go")
    tr = synth.synthetic_normal((B, 3), 6, "mano/tr", 0.3)
    out = layer(betas=gpu(betas), global_orient=gpu(go), hand_pose=gpu(pose), transl=gpu(tr))
    ov, oj = oracle(betas, pose, go, tr, return_joints=True)
    assert_close(out.vertices, ov, atol=TOL)
    assert_close(out.joints, oj, atol=TOL)
    cm = layer.vertices_channel_major(gpu(betas), gpu(pose))
    assert_close(cm.permute(0, 2, 1), oracle(betas, pose), atol=TOL)
    # zero pose / zero shape -> the template
    z = layer(betas=torch.zeros(1, 10, device=DEV), global_orient=torch.zeros(1, 3, device=DEV),
              hand_pose=torch.zeros(1, 45, device=DEV), transl=torch.zeros(1, 3, device=DEV)).vertices
    assert_close(z[0], arrays["v_template"].astype(np.float32), atol=1e-6)


# ------------------------------------------------------------------------------------------ GenNet.gen
def _gennet():
    from conftest import GOLDEN, gen_state_dict
    from dvqvae_amd.network.gen_net import GenNet
    net = GenNet()
    # synthetic weights, prior restricted to the K=128 codebooks, object codebook = the reference's features of 128 seed clouds
    sd = gen_state_dict(net.state_dict(), np.load(os.path.join(GOLDEN, "g7_gen.npz")))
    net.load_state_dict(sd, strict=True)
    net.eval().to(DEV)
    net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(DEV))
    return net, sd


def test_gen_end_to_end_golden(golden):
    """BASELINE config 1: 8 objects, the reference called with B=1 per object; here ONE batched call."""
    g = golden("g7_gen")
    net, sd = _gennet()
    obj = synth.synthetic_clouds(8, int(g["n_points"]), seed=int(g["cloud_seed"]))
    q = synth.exp1_noise(8, 9, 512, seed=int(g["noise_seed"]))
    recon, pos, aux = net.gen(gpu(obj), noise=gpu(q), return_aux=True)
    assert_close(aux["feat_type"], g["feat_type"], atol=TOL)
    safe = g["idx6_gap"] > float(g["idx6_margin"])
    assert safe.sum() >= 6 and len(set(g["idx6"][safe].tolist())) >= 6, "the fixture must bite: distinct, well separated object codes"
    idx6 = aux["idx6"][:, 0].cpu().numpy()
    assert np.array_equal(idx6[safe], g["idx6"][safe])
    codes = aux["codes"].cpu().numpy()
    same = (idx6 == g["idx6"]) & np.all(codes.reshape(8, 9) == g["codes"].reshape(8, 9), axis=1)
    print(f"idx6 match {np.mean(idx6 == g['idx6']):.3f}; code match {np.mean(codes == g['codes']):.3f}; samples with all codes "
          f"equal: {same.sum()}/8; set aside by the gap check (fp64 top-2 distance gap of the object code <= {float(g['idx6_margin'])}): {(~safe).sum()}; "
          f"distinct object codes: {len(set(idx6.tolist()))}")
    assert same[safe].all(), "every grasp whose object code is well separated must reproduce the reference's codes"
    s = torch.from_numpy(same)
    assert_close(recon.cpu()[s], g["recon"][same], atol=TOL, what="MANO pose/shape vs reference")
    assert_close(pos.cpu()[s], g["recon_pos"][same], atol=TOL, what="wrist params vs reference")
    # 61-parameter assembly
    p61 = ops.assemble61(recon, pos)
    assert_close(p61, O.assemble61(recon.cpu(), pos.cpu()), atol=0)


def test_gen_real_object_cloud(golden):
    g = golden("g7_gen_juice")
    net, _ = _gennet()
    obj = torch.from_numpy(g["obj_f16"].astype(np.float32))
    q = synth.exp1_noise(8, 9, 512, seed=43)[:1]
    recon, pos = net.gen(gpu(obj), noise=gpu(q))
    assert_close(recon, g["recon"], atol=TOL)
    assert_close(pos, g["recon_pos"], atol=TOL)


def test_gen_batched_equals_loop():
    net, _ = _gennet()
    obj = gpu(synth.synthetic_clouds(6, 512, seed=77))
    q = gpu(synth.exp1_noise(6, 9, 512, seed=78))
    r, p = net.gen(obj, noise=q)
    for b in range(6):
        rb, pb = net.gen(obj[b:b + 1], noise=q[b:b + 1])
        assert torch.equal(rb, r[b:b + 1]) and torch.equal(pb, p[b:b + 1])


def test_gen_full_size_properties():
    """BASELINE's bench size (65 536 grasps, N = 1024) through size-independent properties: the batch is a 2 048-grasp
    block tiled 32 times, so every tile must reproduce the block's result bit for bit (rows are independent and the
    accumulation order does not depend on the M tiling), the block computed alone must give the same bits, a loop of
    B = 1 reference-style calls too, and the whole call is repeatable."""
    net, _ = _gennet()
    blk, reps, N = 2048, 32, 1024
    obj = gpu(synth.synthetic_clouds(blk, N, seed=91))
    q = gpu(synth.exp1_noise(blk, 9, 512, seed=92))
    big_obj, big_q = obj.repeat(reps, 1, 1), q.repeat(reps, 1, 1)
    r, p, aux = net.gen(big_obj, noise=big_q, return_aux=True)
    assert tuple(r.shape) == (blk * reps, 55) and tuple(p.shape) == (blk * reps, 6)
    assert bool(torch.isfinite(r).all()) and bool(torch.isfinite(p).all())
    r0, p0, aux0 = net.gen(obj, noise=q, return_aux=True)
    rt, pt = r.view(reps, blk, 55), p.view(reps, blk, 6)
    assert torch.equal(rt, r0.expand(reps, blk, 55)) and torch.equal(pt, p0.expand(reps, blk, 6))
    assert torch.equal(aux["codes"].view(reps, blk, -1), aux0["codes"].view(1, blk, -1).expand(reps, blk, -1))
    for b in (0, 1, 777, 2047):
        rb, pb = net.gen(obj[b:b + 1], noise=q[b:b + 1])
        assert torch.equal(rb, r0[b:b + 1]) and torch.equal(pb, p0[b:b + 1])
    r2, p2 = net.gen(big_obj, noise=big_q)
    assert torch.equal(r2, r) and torch.equal(p2, p)
    assert int(aux["codes"].min()) >= 0 and int(aux["codes"].max()) < 128


def test_gen_raises_when_prior_exceeds_codebook():
    from dvqvae_amd.network.gen_net import GenNet
    net = GenNet()
    load_synth(net, SEED)                 # unrestricted 512-class prior over K=128 codebooks
    net.to(DEV)
    net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(DEV))
    with pytest.raises(RuntimeError, match="out of range"):
        net.gen(gpu(synth.synthetic_clouds(4, 256, seed=1)))


# ------------------------------------------------------------------------------------------ DVQVAE eval
def test_dvqvae_eval_golden(golden):
    from dvqvae_amd.network.DVQVAE import DVQVAE
    g = golden("g8_dvqvae")
    from conftest import dvqvae_state_dict
    net = DVQVAE(obj_inchannel=4)
    net.load_state_dict(dvqvae_state_dict(net.state_dict(), g), strict=True)     # the seven codebooks come from the fixture
    net.eval().to(DEV)
    obj = synth.synthetic_clouds(3, 512, seed=80)
    hand = synth.synthetic_normal((3, 3, 778), SEED, "dvq/hand", 0.05)
    emb_idx, obj_emb = net(gpu(obj), gpu(hand))
    assert tuple(emb_idx.shape) == (21, 1) and emb_idx.dtype == torch.int64
    safe = g["emb_gap"] > float(g["emb_margin"])           # fp64 top-2 gaps; every sample has its own code in every codebook
    assert safe.sum() >= 18 and all(len(set(r.tolist())) == 3 for r in g["emb_idx"].reshape(7, 3)), "the fixture must bite"
    assert np.array_equal(emb_idx[:, 0].cpu().numpy()[safe], g["emb_idx"][safe])
    if safe[:3].all():
        assert_close(obj_emb, g["obj_emb"], atol=0, what="obj_emb is an exact codebook row")
    net.train()
    with pytest.raises(NotImplementedError):
        net(gpu(obj), gpu(hand))


def test_dvqvae_eval_full_size_properties(golden):
    """BASELINE config 3 (PointNet -> VQ -> embedding forward, N = 1024, batch 16 384) through size-independent properties: the batch
    is a 1 024-sample block tiled 16 times, so every tile must give the block's codes and embeddings bit for bit, the block computed
    alone the same, and single samples too."""
    from dvqvae_amd.network.DVQVAE import DVQVAE
    from conftest import dvqvae_state_dict
    g = golden("g8_dvqvae")
    net = DVQVAE(obj_inchannel=4)
    net.load_state_dict(dvqvae_state_dict(net.state_dict(), g), strict=True)
    net.eval().to(DEV)
    blk, reps, N = 1024, 16, 1024
    obj = gpu(synth.synthetic_clouds(blk, N, seed=81))
    hand = gpu(synth.synthetic_normal((blk, 3, 778), SEED, "dvq/hand/full", 0.05))
    idx_b, emb_b = net(obj, hand)
    idx, emb = net(obj.repeat(reps, 1, 1), hand.repeat(reps, 1, 1))
    B = blk * reps
    assert tuple(idx.shape) == (7 * B, 1) and tuple(emb.shape)[0] == B
    assert torch.equal(idx.view(7, reps, blk), idx_b.view(7, 1, blk).expand(7, reps, blk))
    assert torch.equal(emb.view(reps, blk, -1), emb_b.view(1, blk, -1).expand(reps, blk, -1))
    for b in (0, 513, 1023):
        i1, e1 = net(obj[b:b + 1], hand[b:b + 1])
        assert torch.equal(i1.view(7), idx_b.view(7, blk)[:, b]) and torch.equal(e1, emb_b[b:b + 1])
    assert int(idx.min()) >= 0 and int(idx.max()) < 128 and len(torch.unique(idx_b.view(7, blk)[6])) > 8


def test_gen_ho3d_size_properties():
    """BASELINE config 4 (prior sampling + decode at batch 8 192 on HO3D-sized clouds, N = 3000: twelve 256-point tiles per cloud, the
    filtered trunk): a 512-grasp block tiled 16 times reproduces the block, the block alone and single grasps, bit for bit."""
    net, _ = _gennet()
    blk, reps, N = 512, 16, 3000
    obj = gpu(synth.synthetic_clouds(blk, N, seed=93))
    q = gpu(synth.exp1_noise(blk, 9, 512, seed=94))
    r0, p0, aux0 = net.gen(obj, noise=q, return_aux=True)
    r, p, aux = net.gen(obj.repeat(reps, 1, 1), noise=q.repeat(reps, 1, 1), return_aux=True)
    assert bool(torch.isfinite(r).all()) and bool(torch.isfinite(p).all())
    assert torch.equal(r.view(reps, blk, 55), r0.expand(reps, blk, 55)) and torch.equal(p.view(reps, blk, 6), p0.expand(reps, blk, 6))
    assert torch.equal(aux["codes"].view(reps, blk, -1), aux0["codes"].view(1, blk, -1).expand(reps, blk, -1))
    for b in (0, 255, 511):
        rb, pb = net.gen(obj[b:b + 1], noise=q[b:b + 1])
        assert torch.equal(rb, r0[b:b + 1]) and torch.equal(pb, p0[b:b + 1])


# ------------------------------------------------------------------------------------------ pre/post steps
def test_transform_cloud_matches_numpy():
    pc = synth.synthetic_clouds(1, 300, seed=3)[0]
    B = 5
    ang = np.random.default_rng(0).uniform(0, 2 * np.pi, size=(B, 3))
    Rs = []
    for a in ang:
        cx, sx, cy, sy, cz, sz = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
        Rs.append(np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                  @ np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]))
    R = torch.tensor(np.stack(Rs), dtype=torch.float32)
    t = torch.tensor([-0.0793, 0.0208, -0.6924])
    out = ops.transform_cloud(gpu(pc), gpu(R), gpu(t)).cpu()
    ref = torch.einsum("bij,jn->bin", R.double(), pc[:3].double()) + t.double()[None, :, None]
    assert_close(out[:, :3], ref.float(), atol=1e-6)
    assert torch.equal(out[:, 3], pc[3].expand(B, -1))


# ------------------------------------------------------------------------------------------ VQ fast path
def _fast_vs_exact(z, E, what):
    fast = ops.vq_argmin(z, E, fast=True)
    exact = ops.vq_argmin(z, E, fast=False)
    assert torch.equal(fast, exact), f"{what}: filter+refine != exact kernel on {(fast != exact).sum().item()} rows"
    return fast


@pytest.mark.parametrize("M", [1, 31, 32, 33, 1000, 4096])
def test_vq_fast_equals_exact_and_canonical(M):
    E = synth.synthetic_normal((512, 256), SEED, "vq/E/512/256")
    z = synth.synthetic_normal((M, 256), SEED, f"vqfast/z/{M}")
    idx = _fast_vs_exact(gpu(z), gpu(E), f"M={M}")
    ci, _ = vq_canonical.argmin(z.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), ci)


def test_vq_fast_golden(golden):
    g = golden("g2_vq")
    E = synth.synthetic_normal((512, 256), SEED, "vq/E/512/256")
    z = synth.synthetic_normal((4096, 256), SEED, "vq/z/512/256/4096")
    idx = ops.vq_argmin(gpu(z), gpu(E), fast=True).cpu().numpy()
    safe = g["K512_D256_M4096_gap"] > 1e-3
    assert np.array_equal(idx[safe], g["K512_D256_M4096_idx"][safe])


def test_vq_fast_adversarial_rows():
    """Near-ties, exact ties (duplicated entries, > 8 duplicates -> full fallback), NaN / Inf rows, huge and tiny scales."""
    K, D = 512, 256
    E = synth.synthetic_normal((K, D), 9, "vqadv/E")
    E[100] = E[7]; E[300] = E[7]                              # 3-way exact tie
    for k in range(400, 412):                                  # 12 copies: candidate list overflows -> block fallback
        E[k] = E[399]
    E[50] = E[49] + 1e-6 * synth.synthetic_normal((D,), 9, "vqadv/eps")     # near tie below bf16 resolution
    z = synth.synthetic_normal((256, D), 9, "vqadv/z")
    z[0] = E[7]; z[1] = E[300]; z[2] = E[405]; z[3] = E[50]; z[4] = E[49]
    z[5, 17] = float("nan"); z[6, 3] = float("inf"); z[7] = 0.0; z[8] = 1e18 * z[8]; z[9] = 1e-20 * z[9]
    z[10] = 3e38; z[11] = -E[7]
    z[12:64] = E[torch.arange(52) * 9] + 1e-3 * z[12:64]      # rows sitting almost on codebook entries
    idx = _fast_vs_exact(gpu(z), gpu(E), "adversarial")
    ci, _ = vq_canonical.argmin(z.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), ci)
    assert idx[0] == 7 and idx[1] == 7 and idx[2] == 399


def test_vq_fast_incomplete_candidate_lists():
    """Rows with more near-minimal entries than a lane can track (four): the in-kernel second-level filter (plain fp32
    distances, then canonical chains for <= 64 entries), its overflow to the all-entries canonical scan (> 64 entries,
    or the workgroup's pair list full), many such rows in one workgroup."""
    K, D = 512, 256
    E = synth.synthetic_normal((K, D), 11, "vqinc/E")
    noise = synth.synthetic_normal((K, D), 11, "vqinc/n")
    for k in range(1, 100):                                    # 100 entries within 1e-6 of each other
        E[k] = E[0] + 1e-7 * noise[k]
    for k in range(201, 210):                                  # 10 entries, same
        E[k] = E[200] + 1e-7 * noise[k]
    for k in range(301, 340):                                  # 40 exact copies
        E[k] = E[300]
    z = synth.synthetic_normal((640, D), 11, "vqinc/z")
    z[0:200] = E[0] + 1e-3 * z[0:200]                          # > 64 second-level candidates: whole workgroups of them
    z[200:330] = E[200] + 1e-3 * z[200:330]                    # 10 candidates per row: > 768 pairs in one workgroup
    z[330:400] = E[300] + 1e-3 * z[330:400]                    # 40-way exact ties -> first index
    z[400] = E[300]
    idx = _fast_vs_exact(gpu(z), gpu(E), "incomplete lists")
    ci, _ = vq_canonical.argmin(z.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), ci)
    assert int(idx[400]) == 300 and bool((idx[330:400] == 300).all())


def test_quantizer_switches_to_exact_kernel_when_ill_conditioned():
    """Reference-initialised codebook U(+-1/n_e) against O(1) features: nearly every row needs the fast kernel's slow
    all-entries scan; the module must notice (device counter, no sync) and use the exact kernel.  Indices are the same
    either way.  A well-separated codebook stays on the fast kernel."""
    from dvqvae_amd.network.vqvae.quantizer import VectorQuantizer
    torch.manual_seed(0)
    vq = VectorQuantizer(512, 256, 0.25, 1.0).to(DEV).eval()                  # reference init
    z = gpu(synth.synthetic_normal((4096, 256), 12, "vqreg/z"))
    want = ops.vq_argmin(z, vq.embedding.weight.detach(), fast=False)
    for _ in range(4):
        idx, _ = vq(z, False)
        torch.cuda.synchronize()
        assert torch.equal(idx.squeeze(1), want)
    assert vq._regime_state["prefer_exact"]
    assert int(vq._regime_state["counter"][0]) > 4096 // 16
    with torch.no_grad():
        vq.embedding.weight.normal_()                                            # new codebook -> new pack, new state
    want = ops.vq_argmin(z, vq.embedding.weight.detach(), fast=False)
    for _ in range(4):
        idx, _ = vq(z, False)
        torch.cuda.synchronize()
        assert torch.equal(idx.squeeze(1), want)
    assert not vq._regime_state["prefer_exact"]


def test_vq_fast_fuzz_shapes_scales_and_degenerate_rows():
    """160 random problems: ragged and large M, global and per-row scales over 12 decades, rows sitting near entries,
    duplicated entries, sparse rows, offset data (cancellation), NaN / Inf rows -- fast kernel == exact kernel everywhere."""
    rng = np.random.default_rng(0)
    torch.manual_seed(1)
    sizes = [1, 2, 31, 32, 33, 127, 128, 129, 255, 257, 1000, 4097, 30000, 65536, 70001, 200000]
    for it in range(160):
        M = int(rng.choice(sizes)) if it % 4 else int(rng.integers(1, 3000))
        kind = it % 8
        E = torch.randn(512, 256, device=DEV)
        z = torch.randn(M, 256, device=DEV)
        if kind == 1:
            E, z = E * 10 ** float(rng.uniform(-6, 6)), z * 10 ** float(rng.uniform(-6, 6))
        elif kind == 2:
            z = E[torch.randint(0, 512, (M,), device=DEV)] + 10 ** float(rng.uniform(-6, -1)) * z
        elif kind == 3:
            E[torch.randint(0, 512, (40,), device=DEV)] = E[0].clone()
        elif kind == 4:
            z = z * torch.rand(M, 1, device=DEV) * 100
        elif kind == 5:
            z[:, 64:] = 0
        elif kind == 6:
            E, z = E + 5.0, z + 5.0
        elif kind == 7 and M > 10:
            z[int(rng.integers(0, M))] = float("nan")
            z[int(rng.integers(0, M)), 3] = float("inf")
        _fast_vs_exact(z, E, f"fuzz case {it} (M={M}, kind={kind})")


def test_ops_on_two_streams_do_not_share_scratch():
    """Kernels are enqueued on the caller's current stream and the scratch is per (device, stream): two streams running
    different problems concurrently must both get the single-stream answers."""
    E = gpu(synth.synthetic_normal((512, 256), 13, "vqstr/E"))
    za = gpu(synth.synthetic_normal((8192, 256), 13, "vqstr/za"))
    zb = gpu(synth.synthetic_normal((8192, 256), 13, "vqstr/zb"))
    pk = ops.vq_pack(E)
    want_a, want_b = ops.vq_argmin(za, E, fast=False), ops.vq_argmin(zb, E, fast=False)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(10):
        with torch.cuda.stream(sa):
            outs_a.append(ops.vq_argmin(za, E, fast=False))
            outs_a.append(ops.vq_argmin(za, E, packed=pk))
        with torch.cuda.stream(sb):
            outs_b.append(ops.vq_argmin(zb, E, packed=pk))
            outs_b.append(ops.vq_argmin(zb, E, fast=False))
    torch.cuda.synchronize()
    assert all(torch.equal(o, want_a) for o in outs_a) and all(torch.equal(o, want_b) for o in outs_b)


def test_pointnet_pipeline_on_two_streams_and_one():
    """dvq_pointnet_encode spreads a batch of >= 2 launches over the caller's stream and a library-owned second stream (exact stage and
    STN FCs of launch i beside the trunk kernel of launch i + 1, two scratch sets).  Two caller streams encoding different batches at
    the same time (each gets a side stream and scratch of its own), back to back, must both give the one-stream answers
    (DVQ_PN_STREAMS=0) bit for bit, and so must the default path on the default stream."""
    from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
    net = PointNetEncoder(channel=4)
    load_synth(net, 3)
    net = net.eval().to(DEV)
    xa, xb = gpu(synth.synthetic_clouds(2304, 1024, seed=21)), gpu(synth.synthetic_clouds(2100, 778, seed=22))
    want_a = _with_env("DVQ_PN_STREAMS", "0", lambda: net(xa))
    want_b = _with_env("DVQ_PN_STREAMS", "0", lambda: net(xb))
    fa, ta, _ = net(xa)
    assert torch.equal(fa, want_a[0]) and torch.equal(ta, want_a[1])
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    for _ in range(6):
        with torch.cuda.stream(sa):
            outs_a.append(net(xa))
        with torch.cuda.stream(sb):
            outs_b.append(net(xb))
    torch.cuda.synchronize()
    assert all(torch.equal(o[0], want_a[0]) and torch.equal(o[1], want_a[1]) for o in outs_a)
    assert all(torch.equal(o[0], want_b[0]) and torch.equal(o[1], want_b[1]) for o in outs_b)
    assert ops.pointnet_fault_counters() == (0, 0)


@pytest.mark.parametrize("K", [32, 128, 256, 480])
def test_vq_fast_padded_codebooks(K):
    """Round 6: codebooks of fewer than 512 entries at D = 256 (the model's six K = 128 codebooks) run the fast kernel on an image padded
    to 512 entries -- zero rows whose |e|^2 is 3e38: they never win, never become candidates, and the canonical all-entries scan stops
    at K.  Fast == exact == oracle/vq_canonical.c on random rows, rows on / next to entries, exact ties (the lowest index wins), NaN /
    Inf / zero / huge rows (the all-entries path), a duplicated LAST entry, scaled data.  (DVQVAE.forward's six codebooks take this path
    in the G8 golden test.)"""
    D = 256
    assert ops.vq_fast_supported(K, D) and not ops.vq_fast_supported(K + 1, D) and not ops.vq_fast_supported(544, D) and not ops.vq_fast_supported(K, 128)
    E = synth.synthetic_normal((K, D), 21, f"vqpad/E/{K}")
    E[K - 1] = E[3]                                            # a tie between a low entry and the last real one: 3 wins
    if K > 40:
        E[40] = E[39] + 1e-6 * synth.synthetic_normal((D,), 21, "vqpad/eps")
    z = synth.synthetic_normal((3000, D), 21, f"vqpad/z/{K}")
    z[0] = E[3]; z[1] = E[K - 1]; z[2] = E[K - 2]; z[3] = 0.0
    z[4, 5] = float("nan"); z[5, 7] = float("inf"); z[6] = 3e38; z[7] = 1e18 * z[7]; z[8] = 1e-20 * z[8]; z[9] = -E[0]
    z[10:10 + min(K, 200)] = E[:min(K, 200)] + 1e-3 * z[10:10 + min(K, 200)]
    idx = _fast_vs_exact(gpu(z), gpu(E), f"padded K={K}")
    ci, _ = vq_canonical.argmin(z.numpy(), E.numpy())
    assert np.array_equal(idx.cpu().numpy(), ci)
    assert idx[0] == 3 and idx[1] == 3 and idx[2] == K - 2 and int(idx.max()) < K
    for scale_z, scale_e in [(100.0, 0.01), (1e-3, 1e3), (1.0, 1.0 / K)]:
        Es = synth.synthetic_uniform((K, D), 22, f"vqpad/Es/{K}/{scale_e}", -scale_e, scale_e)
        zs = synth.synthetic_normal((2048, D), 22, f"vqpad/zs/{scale_z}", scale_z)
        _fast_vs_exact(gpu(zs), gpu(Es), f"padded K={K}, scales {scale_z},{scale_e}")
    for M in (1, 33, 70001):
        zz = torch.randn(M, D, device=DEV)
        _fast_vs_exact(zz, gpu(E), f"padded K={K}, M={M}")
    # the other kernels (DVQ_VQ_KERNEL=17 / 8 / 32) must not be handed a padded image: the selection falls back to the default one
    for kern in ("17", "8", "32"):
        got = _with_env("DVQ_VQ_KERNEL", kern, lambda: ops.vq_argmin(gpu(z), gpu(E), fast=True))
        assert torch.equal(got, idx), f"DVQ_VQ_KERNEL={kern} with a padded codebook"


def test_vq_fast_scales_and_tie_prone_codebook():
    for scale_z, scale_e in [(1.0, 1.0 / 512), (100.0, 0.01), (1e-3, 1e3), (30.0, 30.0)]:
        E = synth.synthetic_uniform((512, 256), 10, f"vqs/E/{scale_e}", -scale_e, scale_e)
        z = synth.synthetic_normal((2048, 256), 10, f"vqs/z/{scale_z}", scale_z)
        _fast_vs_exact(gpu(z), gpu(E), f"scales {scale_z},{scale_e}")


def test_vq_fast_full_size():
    """BASELINE config 2 (M=65536, K=512, D=256): fast == exact on every row; a strided sample vs the C oracle."""
    E = gpu(synth.synthetic_normal((512, 256), 5, "vqf/E"))
    z = gpu(synth.synthetic_normal((65536, 256), 5, "vqf/z"))
    packed = ops.vq_pack(E)
    idx = ops.vq_argmin(z, E, packed=packed)
    assert torch.equal(idx, ops.vq_argmin(z, E, fast=False))
    rows = torch.arange(0, 65536, 97)
    ci, _ = vq_canonical.argmin(z[rows].cpu().numpy(), E.cpu().numpy())
    assert np.array_equal(idx[rows].cpu().numpy(), ci)
    # repeatability (nothing may leak between launches)
    for _ in range(3):
        assert torch.equal(ops.vq_argmin(z, E, packed=packed), idx)


# ------------------------------------------------------------------------------------------ generation plumbing
def test_generate_for_object_and_sharding_equivalence(tmp_path):
    from dvqvae_amd import generate
    net, _ = _gennet()
    obj = synth.synthetic_clouds(1, 700, seed=3)[0]
    rng = np.random.default_rng(0)
    q = gpu(synth.exp1_noise(6, 9, 512, seed=8))
    out = generate.generate_for_object(net, obj, 6, True, rng, noise=q)
    p = out["params"]
    assert tuple(p.shape) == (6, 61) and tuple(out["vertices"].shape) == (6, 778, 3)
    js = out["json"]
    assert len(js["recon_params"]) == 6 and len(js["recon_params"][0]) == 1 and len(js["recon_params"][0][0]) == 61
    assert np.asarray(js["R_list"]).shape == (6, 3, 4) and np.asarray(js["r_list"]).shape == (6, 3)
    # batch sharding (what two ranks would compute) == the unsharded call: objects are independent
    R = torch.as_tensor(np.asarray(js["R_list"])[:, :, :3], dtype=torch.float32, device=DEV)
    t = torch.tensor(generate.CANONICAL_OFFSET, device=DEV)
    batch = ops.transform_cloud(gpu(obj), R, t)
    halves = [ops.assemble61(*net.gen(batch[lo:hi], noise=q[lo:hi])) for lo, hi in ((0, 3), (3, 6))]
    assert torch.equal(torch.cat(halves), p)
    # the final posed-MANO pass (gen_diverse_grasp_obman.py:252-253) against the oracle
    oracle = mano_oracle.ManoOracle(mano_oracle.synthetic_mano_arrays())
    pc = p.cpu()
    ov = oracle(pc[:, :10], pc[:, 13:58], pc[:, 10:13], pc[:, 58:61])
    assert_close(out["vertices"], ov, atol=TOL)


def test_entry_point_writes_reference_json(tmp_path):
    from dvqvae_amd import diversity, generate
    out_dir = str(tmp_path / "ho3d")
    paths = generate.main("ho3d", ["--num_grasp", "5", "--num_objects", "2", "--points", "300", "--out_dir", out_dir,
                                            "--checkpoint", "/nonexistent", "--mano_model", "/nonexistent"])
    assert len(paths) == 2
    params = diversity.load_params(paths)
    assert params.shape == (10, 61) and np.isfinite(params).all()


# ------------------------------------------------------------------------------------------ round-2 additions
def test_mfma_keeps_f16_subnormals():
    """The fast VQ kernel's error bound counts on the matrix core multiplying fp16 SUBNORMAL inputs exactly (it measures /
    bounds the rounding error of the conversion, not of a flush inside the MFMA)."""
    got, want = ops.probe_f16_subnormal(DEV)
    assert got == want and want > 0.0, f"fp16 MFMA with a subnormal input gave {got}, exact value {want}"


def test_vqvae_train_forward_golden(golden):
    """VQVAE.forward (train mode: loss, straight-through z_q, perplexity), network/VQVAE.py:29-42 and
    vqvae/quantizer.py:56-64, against the values the reference produced (G3)."""
    from dvqvae_amd.network.VQVAE import VQVAE
    g = golden("g2_vq")
    vq = VQVAE(128, 32, 2, 128, 256, 0.25, a=1).to(DEV)
    with torch.no_grad():
        vq.vector_quantization.embedding.weight.copy_(torch.from_numpy(g["crafted_E"]))
    vq.train()
    zt = gpu(synth.synthetic_normal((64, 256), SEED, "vq/z/train"))
    loss, zq, perp = vq(zt)
    assert_close(loss, g["train_loss"], atol=1e-6)
    assert_close(perp, g["train_perplexity"], atol=1e-4)
    assert_close(zq.double().sum(1).float(), g["train_zq_rowsum"], atol=1e-4)
    o_loss, o_zq, o_perp, o_onehot, o_idx = O.vq_train_forward(torch.from_numpy(g["crafted_E"]), zt.cpu(), beta=0.25, al=1)
    l5, z5, p5, onehot, idx = vq.vector_quantization(zt, True)
    assert torch.equal(idx.cpu(), o_idx) and torch.equal(onehot.cpu(), o_onehot)
    assert_close(z5, o_zq, atol=1e-6)
    # straight-through estimator: the gradient of z_q w.r.t. z is the identity, the codebook gets the commitment term
    zr = zt.clone().requires_grad_(True)
    loss_r, zq_r, _ = vq(zr)
    (zq_r.sum() + loss_r).backward()
    assert zr.grad is not None and vq.vector_quantization.embedding.weight.grad is not None
    assert_close(zr.grad, 1.0 + 2.0 * (zr.detach() - zq_r.detach()) / zr.numel(), atol=1e-6)


def test_gen_bench_config_vs_oracle():
    """The benchmark's network (K = 512 codebooks, 512 prior classes, N = 1024 points, synthetic weights) against the CPU
    oracle on 256 grasps: object codes and sampled codes exact, MANO pose/shape and wrist parameters within 1e-5.  Grasps
    whose decision an fp32 rounding difference could flip are set aside by stated margins and counted."""
    from dvqvae_amd.network.gen_net import GenNet
    B, N, K = 256, 1024, 512
    net = GenNet(n_embeddings=K, prior_tokens=K, prior_classes=K)
    sd = synth.synthetic_state_dict(net.state_dict(), 1234)
    net.load_state_dict(sd)
    net.eval().to(DEV)
    sd = synth.diversify_object_codebook(net, sd, N)       # object codebook = the net's own features of 512 seed clouds (as bench.py)
    arrays = dmano.synthetic_mano_arrays()
    net.set_rh_mano(dmano.ManoLayer(arrays).to(DEV))
    obj = synth.synthetic_clouds(B, N, seed=4242)
    q = synth.exp1_noise(B, 9, K, seed=4243)
    recon, pos, aux = net.gen(gpu(obj), noise=gpu(q), return_aux=True)
    with torch.no_grad():
        o_recon, o_pos, o_aux = O.gen({k: v.cpu() for k, v in sd.items()}, obj, q, mano_oracle.ManoOracle(arrays), return_aux=True)
    safe_idx = (o_aux["idx6_gap"] > O.object_code_margin(o_aux, aux["feat_type"])).numpy()   # margin from the measured feature difference
    safe_race = o_aux["race_gap"].numpy() > 1e-4             # fp32 logits differ by ~1e-6 relative between the two paths
    safe = safe_idx & safe_race
    idx_ok = (aux["idx6"].cpu() == o_aux["idx6"]).reshape(B, -1).all(1).numpy()
    code_ok = (aux["codes"].cpu() == o_aux["codes"]).reshape(B, -1).all(1).numpy()
    print(f"bench-config parity on {B} grasps: object code match {idx_ok.mean():.4f}, sampled codes match {code_ok.mean():.4f}; "
          f"set aside: {(~safe_idx).sum()} by the object-code gap, {(safe_idx & ~safe_race).sum()} by the race margin")
    n_codes = len(set(o_aux["idx6"].reshape(-1).tolist()))
    print(f"distinct object codes over the {B} grasps: {n_codes}")
    assert n_codes >= 32, "the object-code argmin must be a real decision on the benchmark network"
    assert (~safe_idx).sum() <= 0.02 * B and safe.sum() >= 0.95 * B, "the derived margins should set aside at most a few grasps"
    assert idx_ok[safe_idx].all() and code_ok[safe].all()
    both = torch.from_numpy(idx_ok & code_ok)
    assert_close(recon.cpu()[both], o_recon[both], atol=TOL, what="MANO pose/shape")
    assert_close(pos.cpu()[both], o_pos[both], atol=TOL, what="wrist parameters")


def test_gen_without_host_sync_returns_the_same_results():
    """gen(check=False) enqueues and returns (no .item()): same tensors as the checked call, the error flag left on the device."""
    net, _ = _gennet()
    obj = gpu(synth.synthetic_clouds(5, 300, seed=31))
    q = gpu(synth.exp1_noise(5, 9, 512, seed=32))
    r0, p0 = net.gen(obj, noise=q)
    r1, p1, aux = net.gen(obj, noise=q, check=False, return_aux=True)
    assert torch.equal(r0, r1) and torch.equal(p0, p1) and int(aux["err"].item()) == 0


def test_gen_falls_back_to_bf16x3_beyond_fp16_range():
    """The fp16 three-product GEMMs turn a row with an activation beyond +-65 504 into NaN; gen() notices (one finite check in
    its single host sync) and generates the batch again on the six-product bf16 split.  A decoder weight scaled so that the
    hidden activations leave fp16's range: the result must be finite and equal the DVQ_GEMM=bf16x3 result."""
    net, _ = _gennet()
    obj = gpu(synth.synthetic_clouds(6, 300, seed=77))
    q = gpu(synth.exp1_noise(6, 9, 512, seed=78))
    with torch.no_grad():
        net.decoder.MLP.L0.weight.mul_(1.0e6)
        net.decoder.MLP.L1.weight.mul_(1.0e-6)
    try:
        r_def, p_def = net.gen(obj, noise=q)
        assert net.range_fallbacks == 1, "the scaled decoder must push a hidden activation beyond fp16's range"
        r_b, p_b = _with_env("DVQ_GEMM", "bf16x3", lambda: net.gen(obj, noise=q))
    finally:
        with torch.no_grad():
            net.decoder.MLP.L0.weight.mul_(1.0e-6)
            net.decoder.MLP.L1.weight.mul_(1.0e6)
    assert bool(torch.isfinite(r_def).all()) and bool(torch.isfinite(p_def).all())
    assert torch.equal(r_def, r_b) and torch.equal(p_def, p_b)
    r2, p2 = net.gen(obj, noise=q)                          # back on the default images
    assert bool(torch.isfinite(r2).all()) and net.range_fallbacks == 1
    # the same inside the prior: a residual stream beyond fp16's range makes every later logit NaN; the draw kernel reports it
    # (bit 2 of the flag) instead of silently drawing code 0
    lab = gpu(torch.arange(6) % 4)
    with torch.no_grad():
        net.GatedPixelCNN.layers[2].horiz_resid.weight.mul_(1.0e7)
    try:
        c_def = net.GatedPixelCNN.generate(None, lab, batch_size=6, noise=q)
        c_b = _with_env("DVQ_GEMM", "bf16x3", lambda: net.GatedPixelCNN.generate(None, lab, batch_size=6, noise=q))
    finally:
        with torch.no_grad():
            net.GatedPixelCNN.layers[2].horiz_resid.weight.mul_(1.0e-7)
    assert torch.equal(c_def, c_b)
    assert int((c_b != 0).sum()) > 0, "the bf16 split must draw real codes where the fp16 images had NaN logits"


def test_gen_range_fallback_regenerates_only_the_rows_that_need_it():
    """One grasp of a batch leaves fp16's range (its cloud is scaled until the PointNet feature, an input of the decoder, passes
    65 504): gen() regenerates THAT row on the six-product images under the same noise key -- the row equals the DVQ_GEMM=bf16x3
    result bit for bit, every other row keeps the bits of a clean call, and at the benchmark's batch a step with one such row costs
    what a clean step costs (the whole-batch re-run of round 4 cost 1.6 x)."""
    net, _ = _gennet()
    B, bad = 40, 17
    obj = gpu(synth.synthetic_clouds(B, 300, seed=91))
    clean = obj.clone()
    scale = None
    for s_try in (1.0e3, 1.0e4, 1.0e5, 1.0e6):                                   # the smallest scale that leaves fp16's range
        with torch.no_grad():
            f_bad, _, _ = net.obj_encoder_type(clean[bad:bad + 1] * s_try)
        if float(f_bad.abs().max()) > 7.0e4:
            scale = s_try
            break
    assert scale is not None, "no scale pushed the PointNet feature (a decoder input) beyond fp16's range"
    obj[bad] *= scale
    n0, r0 = net.range_fallbacks, net.range_fallback_rows
    r, p, aux = net.gen(obj, seed=5, row0=300, stream_id=2, return_aux=True)
    assert net.range_fallbacks == n0 + 1 and net.range_fallback_rows == r0 + 1 and aux["fallback_rows"].tolist() == [bad]
    assert bool(torch.isfinite(r).all()) and bool(torch.isfinite(p).all())
    rc, pc = net.gen(clean, seed=5, row0=300, stream_id=2)                       # no row out of range: no fallback
    assert net.range_fallbacks == n0 + 1
    keep = [i for i in range(B) if i != bad]
    assert torch.equal(r[keep], rc[keep]) and torch.equal(p[keep], pc[keep]), "rows inside the range must keep their bits"
    rb, pb = _with_env("DVQ_GEMM", "bf16x3", lambda: net.gen(obj, seed=5, row0=300, stream_id=2))
    assert torch.equal(r[bad], rb[bad]) and torch.equal(p[bad], pb[bad]), "the regenerated row must be the six-product result"
    # the cost at the benchmark's batch: ONE row of 65 536 out of range.  (Not through a scaled cloud: such a cloud is also a
    # degenerate input of the PointNet filter -- most of its points tie -- and costs milliseconds by itself.)  One entry of the first
    # hand codebook is made huge and the prior's noise is set so that exactly one row draws it: that row's decoder input overflows.
    Bb, row, j = 65536, 40000, 77
    big = clean[torch.arange(Bb, device=DEV) % B].contiguous()
    q = torch.empty(Bb, 9, 512, device=DEV).exponential_()
    q[:, 1, j] = 1.0e30                                                          # grid position (0, 1) -> vqvae0 (gen_net.CODE_SLOTS): never code j ...
    E0 = net.vqvae0.vector_quantization.embedding.weight
    saved = E0[j].clone()

    def timed(x, noise):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = net.gen(x, noise=noise, return_aux=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out
    try:
        with torch.no_grad():
            E0[j] = 3.0e6
        timed(big, q)
        t_clean = min(timed(big, q)[0], timed(big, q)[0])
        n1 = net.range_fallback_rows
        q[row, 1, j] = 1.0e-30                                                   # ... except this row
        timed(big, q)                                                            # (first fallback: the bf16x3 images are built)
        t_bad, (rr, pp, aux_b) = min((timed(big, q) for _ in range(2)), key=lambda v: v[0])
        assert net.range_fallback_rows == n1 + 3 and aux_b["fallback_rows"].tolist() == [row]
        assert int(aux_b["codes"].view(Bb, 9)[row, 1]) == j and bool(torch.isfinite(rr).all()) and bool(torch.isfinite(pp).all())
    finally:
        with torch.no_grad():
            E0[j] = saved
    assert t_bad <= 1.05 * t_clean + 0.002, f"one out-of-range row: {t_bad * 1e3:.1f} ms against {t_clean * 1e3:.1f} ms for a clean step"


def test_every_entry_point_survives_the_fp16_range():
    """ops.linear / ops.mlp3 / Decoder / GatedPixelCNN.forward / the MANO layer on the default fp16 images: an activation beyond
    +-65 504 must not come back as NaN where the fp32 reference is finite -- each entry point looks at its result and runs again on
    the six-product images (same result as DVQ_GEMM=bf16x3)."""
    from dvqvae_amd import packing
    torch.manual_seed(3)
    x = torch.randn(24, 256, device=DEV)
    x[5] *= 1.0e6
    w, b = torch.randn(64, 256, device=DEV) * 0.05, torch.randn(64, device=DEV)
    y = ops.linear(x, w, b, planes=packing.split_planes(w))
    y_ref = _with_env("DVQ_GEMM", "bf16x3", lambda: ops.linear(x, w, b))
    assert bool(torch.isfinite(y).all()) and torch.equal(y, y_ref)
    with ops.no_range_check():                                                  # what the check is for: the raw result has a NaN row
        y_raw = ops.linear(x, w, b, planes=packing.split_planes(w))
    assert bool(torch.isnan(y_raw[5]).all()) and bool(torch.isfinite(y_raw[:5]).all())
    net, _ = _gennet()
    z = torch.randn(24, 2560, device=DEV)
    z[7] *= 1.0e6
    d = net.decoder(z)
    d_ref = _with_env("DVQ_GEMM", "bf16x3", lambda: net.decoder(z))
    assert bool(torch.isfinite(d).all()) and torch.equal(d, d_ref)
    tok = gpu(torch.arange(54).reshape(6, 3, 3) % 128)
    lab = gpu(torch.arange(6) % 4)
    with torch.no_grad():
        net.GatedPixelCNN.layers[2].horiz_resid.weight.mul_(1.0e7)
    try:
        lg = net.GatedPixelCNN(tok, lab)
        lg_ref = _with_env("DVQ_GEMM", "bf16x3", lambda: net.GatedPixelCNN(tok, lab))
    finally:
        with torch.no_grad():
            net.GatedPixelCNN.layers[2].horiz_resid.weight.mul_(1.0e-7)
    assert bool(torch.isfinite(lg_ref).all()) and torch.equal(lg, lg_ref)
    betas, pose = torch.randn(8, 10, device=DEV), torch.randn(8, 45, device=DEV) * 0.3
    betas[2] *= 1.0e6
    v = net.rh_mano(betas=betas, global_orient=torch.zeros(8, 3, device=DEV), hand_pose=pose, transl=torch.zeros(8, 3, device=DEV)).vertices
    v_ref = _with_env("DVQ_GEMM", "bf16x3", lambda: net.rh_mano(betas=betas, global_orient=torch.zeros(8, 3, device=DEV), hand_pose=pose,
                                                                 transl=torch.zeros(8, 3, device=DEV)).vertices)
    assert bool(torch.isfinite(v).all()) and torch.equal(v, v_ref)


def test_checkpoint_files_load_like_the_reference(tmp_path):
    """The loading branch of the entry points with the files a user of the reference has (gen_diverse_grasp_obman.py:333-346):
    ``model_best.pth`` = {'network': state_dict} with keys GenNet does not have (the trainer's hand encoders) and without some it has
    (those keep the module's own initial values), filtered into GenNet.state_dict(); the prior's state_dict as a file of its own,
    loaded whole.  The JSON an entry point writes from the files equals, byte for byte, what an in-memory net loaded with the same
    tensors generates.  And: the reference's 128-row codebooks under an unrestricted 512-class prior raise, as there."""
    import json
    from dvqvae_amd import generate
    from dvqvae_amd.network.gen_net import GenNet
    seed = 11
    torch.manual_seed(seed)
    tmpl = GenNet(n_embeddings=128)
    full = synth.synthetic_state_dict(tmpl.state_dict(), 4321)
    full["GatedPixelCNN.output_conv.2.bias"][128:] = -1e4                        # codes stay inside the 128-row codebooks
    missing = [k for k in full if k.endswith("num_batches_tracked")] + ["pos_decoder.MLP.L2.bias"]
    network = {k: v for k, v in full.items() if not k.startswith("GatedPixelCNN.") and k not in missing}
    network["emb_0.MLP.L0.weight"] = torch.randn(512, 1024)                      # keys of the training net GenNet has no use for
    network["fing_3.stn.fc3.bias"] = torch.randn(9)
    prior = {k[len("GatedPixelCNN."):]: v for k, v in full.items() if k.startswith("GatedPixelCNN.")}
    ck, pck = str(tmp_path / "model_best.pth"), str(tmp_path / "LATENT_BLOCK_pixelcnn.pt")
    torch.save({"network": network, "epoch": 3}, ck)
    torch.save(prior, pck)
    out_dir = str(tmp_path / "out")
    argv = ["--num_grasp", "6", "--num_objects", "2", "--points", "700", "--seed", str(seed), "--out_dir", out_dir,
            "--checkpoint", ck, "--prior_checkpoint", pck, "--mano_model", "/nonexistent"]
    paths = generate.main("ho3d", argv)
    assert len(paths) == 2
    # the same tensors, loaded in memory the way the reference's script does
    torch.manual_seed(seed)                                                      # main() seeds before it builds the net: same initial values
    net = GenNet(n_embeddings=128)
    sd = net.state_dict()
    assert float((sd["pos_decoder.MLP.L2.bias"] - full["pos_decoder.MLP.L2.bias"]).abs().max()) > 0, "the missing key must matter"
    sd.update({k: v for k, v in network.items() if k in sd})
    net.load_state_dict(sd)
    net.GatedPixelCNN.load_state_dict(prior)
    net.eval().to(DEV)
    net.set_rh_mano(dmano.ManoLayer(dmano.synthetic_mano_arrays()).to(DEV))
    clouds = synth.synthetic_clouds(2, 700, seed=seed)
    for gi, path in enumerate(paths):
        out = generate.generate_for_object(net, clouds[gi], 6, True, np.random.default_rng([seed, gi]), seed=seed, object_index=gi)
        assert open(path).read() == json.dumps(out["json"]), f"{path}: the files loaded differently from the in-memory net"
    # an unrestricted prior draws codes beyond the 128 codebook rows: the reference's embedding lookup raises; so does the entry point
    prior_u = dict(prior)
    prior_u["output_conv.2.bias"] = torch.zeros_like(prior["output_conv.2.bias"])
    torch.save(prior_u, pck)
    with pytest.raises(RuntimeError, match="out of range"):
        generate.main("ho3d", argv + ["--n_embeddings", "128"])


_RUNTIME_CHECK_CHILD = r"""
import os, sys, json, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import dvqvae_amd
from dvqvae_amd import synth, ops, _lib
from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
from util import load_synth
dev = torch.device("cuda:0")
net = PointNetEncoder(channel=4); load_synth(net, 3); net = net.eval().to(dev)
x = synth.synthetic_clouds(48, 1024, seed=5).to(dev)
def run(**env):
    for k, v in env.items(): os.environ[k] = v
    _lib.load().dvq_reload_env()
    ops.pointnet_fault_counters(reset=True)
    f, tr, _ = net(x)
    torch.cuda.synchronize()
    c = ops.pointnet_fault_counters(reset=True)
    for k in env: del os.environ[k]
    _lib.load().dvq_reload_env()
    return f, tr, c
f_ref, tr_ref, c_ref = run(DVQ_PN_EXHAUSTIVE="1")
f, tr, c = run()
print(json.dumps({"equal": bool(torch.equal(f, f_ref) and torch.equal(tr, tr_ref)), "counters": c, "counters_exhaustive": c_ref}))
"""


def test_pointnet_runtime_checks():
    """The filtered trunk's run-time consistency checks, with faults INJECTED (diagnostics build of the library, child processes):
    (a) a tile record whose top score lies about its tile -> pn_exact_kernel finds the exact maximum outside the interval the
    records promise, evaluates the channel over all points and counts it; (b) a hand-over of top scores that does not happen (the
    ring slot keeps an older chunk's values) -> the publishing wave sees the wrong chunk tag, flags the whole tile and marks the
    record suspect.  Both times the features equal the exhaustive evaluation bit for bit and the counters are not zero; without
    an injected fault both counters stay zero."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "tools", "diag", "libdvq_hip_diag.so")
    if not os.path.exists(diag):
        r = subprocess.run(["make", "-C", os.path.join(root, "d-vqvae_amd", "csrc"), "-j", "8", "diag"], capture_output=True, text=True)
        assert r.returncode == 0 and os.path.exists(diag), r.stdout[-2000:] + r.stderr[-2000:]

    def child(abl):
        env = dict(os.environ, DVQ_DIAG_LIB="1", DVQ_PN_ABL=str(abl))
        r = subprocess.run([sys.executable, "-c", _RUNTIME_CHECK_CHILD, root], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    clean = child(0)
    assert clean["equal"] and clean["counters"] == [0, 0], clean
    lie = child(32768)
    assert lie["equal"], "a lying record must not change a feature"
    assert lie["counters"][1] > 0, lie
    lost = child(65536)
    assert lost["equal"], "a lost hand-over must not change a feature"
    assert lost["counters"][0] > 0, lost
    assert ops.pointnet_fault_counters() == (0, 0), "the product library must not have seen an inconsistency in this test session"


def test_gemm_kernels_repeat_bitwise_under_load():
    """The f16x2 GEMM kernels use explicit packed-fp32 arithmetic in their epilogues (the one place the disassembly fence of
    tests/test_abi.py allows it) and run two waves per SIMD -- the conditions under which round 3's PointNet fault appeared.  A soak
    of their own: the tiled kernel at the benchmark's gated / residual / bias shapes (through the teacher-forced PixelCNN forward of
    4 096 rows) and at a plain M = 16 384 linear, and the skinny kernel at M = 8, each call compared bit for bit with the first."""
    torch.manual_seed(5)
    net, _ = _gennet()
    x = torch.randn(16384, 1536, device=DEV)
    w, b = torch.randn(1024, 1536, device=DEV) / 40.0, torch.randn(1024, device=DEV)
    pl = packing.split_planes(w)
    y0 = ops.linear(x, w, b, relu=True, planes=pl)
    ys0 = ops.linear(x[:8], w, b, planes=pl)
    tok = gpu(torch.randint(0, 128, (4096, 3, 3), generator=torch.Generator().manual_seed(1)))
    lab = gpu(torch.randint(0, 128, (4096,), generator=torch.Generator().manual_seed(2)))
    lg0 = net.GatedPixelCNN(tok, lab)
    bad = 0
    for it in range(60):
        bad += int(not torch.equal(ops.linear(x, w, b, relu=True, planes=pl), y0))
        bad += int(not torch.equal(ops.linear(x[:8], w, b, planes=pl), ys0))
        if it % 4 == 0:
            bad += int(not torch.equal(net.GatedPixelCNN(tok, lab), lg0))
    assert bad == 0, f"{bad} GEMM calls differed from the first call of the same inputs"


def test_gen_sharded_equals_unsharded_with_device_noise():
    """with the device noise keyed by (seed, stream, global row), R contiguous shards of a batch -- each a
    separate gen() call with row0 = its first global row, as R ranks would make them -- reproduce the unsharded call bit for
    bit, in rank-major order (R = 2, 4, 8, ragged shards included)."""
    net, _ = _gennet()
    B = 52
    obj = gpu(synth.synthetic_clouds(B, 384, seed=505))
    r0, p0, a0 = net.gen(obj, seed=77, row0=1000, stream_id=5, return_aux=True)
    r1, p1 = net.gen(obj, seed=77, row0=1000, stream_id=5)
    assert torch.equal(r0, r1) and torch.equal(p0, p1)                      # same key, same draws
    r2, _, a2 = net.gen(obj, seed=77, row0=1000, stream_id=6, return_aux=True)
    assert not torch.equal(a0["codes"], a2["codes"])                        # another stream, other draws
    from dvqvae_amd import dist
    for R in (2, 4, 8):
        parts = []
        for rank in range(R):
            lo, hi = dist.shard_range(B, rank, R)
            rr, pp = net.gen(obj[lo:hi], seed=77, row0=1000 + lo, stream_id=5)
            parts.append(ops.assemble61(rr, pp))
        assert torch.equal(torch.cat(parts), ops.assemble61(r0, p0)), f"R={R}"


def test_fp32_gemm_branch_matches_goldens(tmp_path):
    """DVQ_GEMM=fp32 (v_mfma_f32_32x32x2_f32 GEMMs, unfused PointNet trunk) is chosen when the library loads: a fresh process
    runs the GEMM, PointNet, PixelCNN and end-to-end golden tests of this file on that branch."""
    import os, subprocess, sys
    env = dict(os.environ, DVQ_GEMM="fp32")
    sel = "test_linear or test_pointnet_golden or test_pixelcnn_small_golden or test_decoders_golden or test_gen_end_to_end_golden"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_pointnet_three_workgroups_per_cu_kernel(tmp_path):
    """Diagnostics build + DVQ_PN_TRUNK3=1 (read when the library loads) puts the full tiles on pn_trunk3_kernel -- the trunk kernel laid out for three
    workgroups per CU (168 registers, 52 KB of LDS, 32-channel chunks; measured 4.5 % slower than the default and therefore not the
    default).  A fresh process runs the PointNet tests of this file on it: goldens, filtered == exhaustive bit for
    bit, tail tiles, ties, non-finite inputs, the run-time checks with their fault injection, and the two full-machine stress tests --
    the ones that caught this kernel's barrier without an LDS wait (csrc/dvq_internal.h: dvq_lds_barrier)."""
    import os, subprocess, sys
    env = dict(os.environ, DVQ_PN_TRUNK3="1", DVQ_DIAG_LIB="1")          # (the kernel is compiled into the diagnostics build only)
    sel = ("test_pointnet_golden or test_pointnet_filter or test_pointnet_batched or test_pointnet_runtime_checks or "
           "test_pointnet_large_clouds or test_pointnet_pipeline")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_bf16x3_gemm_branch_matches_goldens(tmp_path):
    """DVQ_GEMM=bf16x3 packs every GEMM weight as the exact three-plane bf16 split (six products, fp32's range: what the
    fp16 three-product default falls back to): a fresh process runs the same golden subset on it."""
    import os, subprocess, sys
    env = dict(os.environ, DVQ_GEMM="bf16x3")
    sel = ("test_linear or test_pointnet_golden or test_pixelcnn_small_golden or test_pixelcnn_sampling_golden or test_decoders_golden "
           "or test_gen_end_to_end_golden or test_gen_batched_equals_loop")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


_GEMM_VARIANT_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[2] + "/tests")
import dvqvae_amd
from dvqvae_amd import ops, packing
from dvqvae_amd.network.pixelcnn.models import GatedPixelCNN
from util import load_synth
dev, out = "cuda:0", {}
torch.manual_seed(0)
for (M, N, K) in ((300, 512, 512), (4, 768, 1024), (16384, 1024, 512), (1000, 256, 96)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    out[f"lin{M}x{N}x{K}"] = ops.linear(x, w, b, relu=True, planes=packing.split_planes(w)).cpu()
net = GatedPixelCNN(512, 512, 15, 128); load_synth(net, 5); net = net.to(dev)
g = torch.Generator().manual_seed(1)
x = torch.randint(0, 512, (150, 3, 3), generator=g).to(dev); lab = torch.randint(0, 128, (150,), generator=g).to(dev)
out["pixelcnn_logits"] = net(x, lab).cpu()
x = torch.randint(0, 512, (700, 3, 3), generator=g).to(dev); lab = torch.randint(0, 128, (700,), generator=g).to(dev)
out["pixelcnn_logits_700"] = net(x, lab).cpu()                  # beyond the small-batch kernel: the tiled kernels' variants
torch.save(out, sys.argv[1])
"""


def test_gemm_tile_variants_agree_bitwise(tmp_path):
    """Six-product kernels (DVQ_GEMM=bf16x3): the 128x256 eight-wave tile (default where N % 256 == 0, N >= 512) and the 128x128
    tile (DVQ_GEMM_WIDE=0) use the same accumulation order.  Three-product kernels (default): the two feeding schedules of the
    tiled kernel (DVQ_GEMM_DEPHASE: ping-pong over three LDS stages = the default, two stages dephased, two stages in lock
    step).  Bias/ReLU GEMMs and the full 15-layer PixelCNN forward (gate, residual, multi-tap sources,
    ragged M) must agree bit for bit -- and repeat bit for bit (a missing wait before the K-loop barrier showed up as
    run-to-run noise)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode, knob in (("bf16x3", "DVQ_GEMM_WIDE"), ("f16x2", "DVQ_GEMM_DEPHASE")):
        outs = []
        # f16x2 also: the ping-pong kernel's tile width (DVQ_GEMM_TN: 128 x 128 / 128 x 256 forced; default = chosen per launch)
        variants = ((("a", "1"), ("b", "0"), ("a2", "1")) if mode == "bf16x3" else
                    (("pp", "2"), ("two-stage", "0"), ("pp2", "2"), ("dephased", "1"), ("tn128", "2"), ("tn256", "2")))
        for tag, val in variants:
            path = str(tmp_path / f"{mode}_{tag}.pt")
            extra = {"DVQ_GEMM_TN": tag[2:]} if tag.startswith("tn") else {}
            r = subprocess.run([sys.executable, "-c", _GEMM_VARIANT_SCRIPT, path, root], env=dict(os.environ, DVQ_GEMM=mode, **{knob: val}, **extra),
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            outs.append(torch.load(path))
        for k in outs[0]:
            for o in outs[1:]:
                assert torch.equal(outs[0][k], o[k]), f"{mode} {k}: the {knob} variants differ or do not repeat"


def test_pointnet_large_clouds_nonfinite_inputs_and_missing_filter_image():
    """N > 16384 runs the six-product trunk (the filter's tile records are sized for 64 tiles); a cloud with a NaN / Inf
    coordinate comes out NaN in every channel on the filtered path (what the reference's affine layers and torch.max make of it;
    the trunk kernel flags the tile, pn_exact_kernel writes the NaNs without evaluating anything), neither crashes nor leaks into
    the neighbouring samples, and filtered == exhaustive still holds; a weights struct without filter images (raw ABI users)
    selects the six-product trunk."""
    net, _ = _pointnet(4, SEED + 9)
    big = gpu(synth.synthetic_clouds(1, 17000, seed=3, channels=4))
    f_big, _, _ = net(big)
    f_big6, _, _ = _with_env("DVQ_PN_FILTER", "0", lambda: net(big))
    assert torch.equal(f_big, f_big6)
    x = gpu(synth.synthetic_clouds(3, 520, seed=4, channels=4))        # two dealt tiles + a tail tile of 8 points
    x[1, 0, 17] = float("nan")
    x[2, 2, 515] = float("inf")                                        # in the tail tile
    feat, _, _ = net(x)
    ref, _, _ = net(x[:1].contiguous())
    assert torch.equal(feat[0], ref[0]) and torch.isfinite(feat[0]).all(), "a finite sample next to non-finite ones must not change"
    assert bool(torch.isnan(feat[1:]).all()), "a cloud with a non-finite coordinate has NaN features (pointnet_encoder.py:156-158 on such an input)"
    feat_all, _, _ = _with_env("DVQ_PN_EXHAUSTIVE", "1", lambda: net(x))
    same = (feat == feat_all) | (torch.isnan(feat) & torch.isnan(feat_all))
    assert bool(same.all()), "non-finite inputs: filtered != exhaustive"
    packed = net.packed()
    saved = (packed.cstruct.w3f, packed.cstruct.s_w3f)
    try:
        packed.cstruct.w3f, packed.cstruct.s_w3f = None, None
        f_nofilter, _, _ = net(x[:1].contiguous())
    finally:
        packed.cstruct.w3f, packed.cstruct.s_w3f = saved
    f6, _, _ = _with_env("DVQ_PN_FILTER", "0", lambda: net(x[:1].contiguous()))
    assert torch.equal(f_nofilter, f6)
    # a struct without ANY pre-split plane (include/dvq.h: the planes are optional, all or nothing) takes the unfused trunk -- conv1 rows in
    # scratch of its own (one launch) or in the second scratch set's (several launches): within 1e-5 of the plane kernels' features
    cs = packed.cstruct
    names = ("w3f", "s_w3f", "w2p", "w3p", "s_w2p", "s_w3p", "s_f1p", "s_f2p", "s_f3p")
    saved = {n: getattr(cs, n) for n in names}
    xs = gpu(synth.synthetic_clouds(2100, 300, seed=6, channels=4))    # > 2 048 samples: several launches, two scratch sets
    try:
        want1, want_many = net(x[:1].contiguous())[0], net(xs)[0]
        for n in names:
            setattr(cs, n, None)
        got1, got_many = net(x[:1].contiguous())[0], net(xs)[0]
    finally:
        for n, v in saved.items():
            setattr(cs, n, v)
    assert_close(got1, want1, atol=TOL)
    assert_close(got_many, want_many, atol=TOL)


@pytest.mark.parametrize("dataset", ["obman", "grab", "FHAB"])
def test_other_entry_points_write_reference_json(tmp_path, dataset):
    """gen_diverse_grasp_{obman,grab,FHAB}.py end to end on two synthetic objects (the ho3d one is covered above): the shims'
    own `main` with the reference's default grasp counts, the reference's JSON layout
    (gen_diverse_grasp_grab.py:302-311)."""
    import importlib, json
    mod = importlib.import_module(f"dvqvae_amd.gen_diverse_grasp_{dataset}")
    out_dir = str(tmp_path / dataset)
    n = {"obman": 1, "grab": 20, "FHAB": 49}[dataset]
    written = mod.main(dataset, ["--num_objects", "2", "--points", "256", "--out_dir", out_dir, "--seed", "3",
                                 "--checkpoint", "/nonexistent", "--mano_model", "/nonexistent"])
    assert len(written) == 2
    for path in written:
        d = json.load(open(path))
        assert set(d) == {"recon_params", "R_list", "trans_list", "r_list"}
        assert len(d["recon_params"]) == n and len(d["recon_params"][0]) == 1 and len(d["recon_params"][0][0]) == 61
        assert np.asarray(d["R_list"]).shape == (n, 3, 4) and len(d["trans_list"]) == n and len(d["r_list"]) == n
        assert np.isfinite(np.asarray(d["recon_params"])).all()


def test_default_noise_is_fresh_per_call_and_follows_manual_seed():
    """Like the reference's multinomial draws (models.py:195), calls that name no noise draw fresh noise each time; the draws
    are governed by torch.manual_seed; prior_classes % 4 != 0 (9 * n_in not a multiple of the generator's quad) works."""
    from dvqvae_amd.network.pixelcnn.models import GatedPixelCNN
    net = GatedPixelCNN(input_dim=30, dim=64, n_layers=2, n_classes=8)
    load_synth(net, SEED + 21)
    net = net.to(DEV)
    lab = torch.arange(64, device=DEV) % 8
    torch.manual_seed(123)
    a = net.generate(None, lab, batch_size=64)
    b = net.generate(None, lab, batch_size=64)
    assert not torch.equal(a, b), "two calls without explicit noise must not repeat the draws"
    torch.manual_seed(123)
    net._noise_stream = 0
    assert torch.equal(net.generate(None, lab, batch_size=64), a), "torch.manual_seed + the same call sequence reproduces the draws"
    torch.manual_seed(124)
    net._noise_stream = 0
    assert not torch.equal(net.generate(None, lab, batch_size=64), a)
    c = net.generate(None, lab, batch_size=64, seed=5, row0=0, stream_id=0)
    assert torch.equal(net.generate(None, lab, batch_size=64, seed=5, row0=0, stream_id=0), c)
    g, _ = _gennet()
    obj = gpu(synth.synthetic_clouds(16, 256, seed=3))
    torch.manual_seed(9)
    r1, _, a1 = g.gen(obj, return_aux=True)
    r2, _, a2 = g.gen(obj, return_aux=True)
    assert not torch.equal(a1["codes"], a2["codes"])
    n = ops.exp1_noise(5, 30, 1, device=DEV)
    assert tuple(n.shape) == (5, 30) and n.is_contiguous() and torch.equal(n, ops.exp1_noise(5, 32, 1, device=DEV)[:, :30])


@pytest.mark.parametrize("kind", ["f16x2", "bf16x3"])
def test_skinny_gemm_equals_tiled_kernels_bitwise(kind):
    """Small M (the reference's B = 1 / 8 / 100 call pattern) runs the skinny kernels; they issue the tiled kernels' MFMA sequence,
    so bias / ReLU, residual, multi-source and gated results (the whole PixelCNN forward) must be bit-identical with
    DVQ_GEMM_SKINNY=0, and a batch must equal its rows computed one by one -- for both weight images."""
    from dvqvae_amd import _lib
    from dvqvae_amd.network.pixelcnn.models import GatedPixelCNN
    torch.manual_seed(5)
    k = _lib.PLANES_F16X2 if kind == "f16x2" else _lib.PLANES_BF16X3

    def run():
        out = {}
        for (M, N, K) in ((1, 9, 256), (5, 55, 256), (16, 512, 512), (17, 128, 96), (32, 512, 512), (33, 1024, 1024), (100, 2048, 512), (256, 128, 1024), (7, 6, 128)):
            g = torch.Generator().manual_seed(M * 1000 + N)
            x = gpu(torch.randn(M, K, generator=g)); w = gpu(torch.randn(N, K, generator=g) * 0.05); b = gpu(torch.randn(N, generator=g))
            out[f"lin{M}x{N}x{K}"] = ops.linear(x, w, b, relu=True, planes=packing.split_planes(w, k))
            x2 = gpu(torch.randn(M, 64, generator=g)); w2 = gpu(torch.randn(N, 64, generator=g))
            pls = packing.split_f16x2([w, w2]) if k == _lib.PLANES_F16X2 else [packing.split_bf16x3(w), packing.split_bf16x3(w2)]
            out[f"multi{M}x{N}"] = ops.linear_multi([(x, w), (x2, w2)], b, planes=pls)
        net = GatedPixelCNN(512, 512, 15, 128)
        load_synth(net, 5)
        net = net.to(DEV)
        g = torch.Generator().manual_seed(1)
        x = gpu(torch.randint(0, 512, (37, 3, 3), generator=g)); lab = gpu(torch.randint(0, 128, (37,), generator=g))
        out["logits"] = net(x, lab)
        out["logits_row3"] = net(x[3:4], lab[3:4])
        return out
    a = _with_env("DVQ_GEMM", kind, run)
    b = _with_env("DVQ_GEMM", kind, lambda: _with_env("DVQ_GEMM_SKINNY", "0", run))
    for key in a:
        assert torch.equal(a[key], b[key]), f"{key}: skinny kernel != tiled kernel"
    if kind == "bf16x3":
        c = _with_env("DVQ_GEMM", kind, lambda: _with_env("DVQ_GEMM_SKINNY", "2", run))   # the register-staged variant of the skinny kernel
        for key in a:
            assert torch.equal(a[key], c[key]), f"{key}: LDS-staged skinny kernel != register-staged one"
    assert torch.equal(a["logits"][3:4], a["logits_row3"])


def test_vq_kernel_variants_agree():
    """The streaming kernels (DVQ_VQ_KERNEL=16 default, 17 = round 6's generated instruction block, 32, 8) return the same indices on
    random rows, ragged sizes (one tile, two launches), ties, near-ties, non-finite and out-of-range rows."""
    torch.manual_seed(11)
    E = gpu(torch.randn(512, 256))
    cases = [gpu(torch.randn(M, 256)) for M in (1, 31, 33, 1000, 8192 + 5, 65536, 69632 + 7)]
    z = torch.randn(300, 256)
    z[0] = E[7].cpu(); z[1] = 0.5 * (E[3] + E[9]).cpu(); z[2, 5] = float("nan"); z[3, 9] = float("inf"); z[4] = 0.0
    z[5] = 7.0e4; z[6] = 1e-6 * z[6]; z[10:40] = E[100:130].cpu() + 1e-4 * torch.randn(30, 256)
    cases.append(gpu(z))
    Et = gpu((torch.rand(512, 256) * 2 - 1) / 512)                    # the reference's initial codebook: tie-prone
    pk, pkt = ops.vq_pack(E), ops.vq_pack(Et)

    def run():
        return [ops.vq_argmin(c_, E, packed=pk, fast=True) for c_ in cases] + [ops.vq_argmin(cases[3], Et, packed=pkt, fast=True)]
    a = run()
    for kern in ("16", "17", "32", "8"):                              # 32: rows resident, codebook streamed (vq_rows.hip)
        b = _with_env("DVQ_VQ_KERNEL", kern, run)
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"DVQ_VQ_KERNEL={kern}, case {i}: {int((x != y).sum())} rows differ"
    assert torch.equal(a[3], ops.vq_argmin(cases[3], E, fast=False))


def test_vq_pipe_kernel_passes_the_fast_path_tests():
    """DVQ_VQ_KERNEL=17 (vq_pipe.hip: the prologue and the eight tile periods as one generated instruction block, a-priori rounding
    bound, merge decided on the scalar unit) under the fast path's own tests: adversarial rows, incomplete candidate lists (pair
    list overflow -> all-entries scan), scales and the tie-prone codebook, ragged sizes against the C oracle, 160-case fuzz, full
    size with repeatability."""
    def run():
        for M in (1, 31, 32, 33, 1000, 4096):
            test_vq_fast_equals_exact_and_canonical(M)
        test_vq_fast_adversarial_rows()
        test_vq_fast_incomplete_candidate_lists()
        test_vq_fast_scales_and_tie_prone_codebook()
        test_vq_fast_fuzz_shapes_scales_and_degenerate_rows()
        test_vq_fast_full_size()
    _with_env("DVQ_VQ_KERNEL", "17", run)
