"""Contact / penetration proxies: oracle sanity on the CPU, HIP == oracle bit for bit on the GPU."""
import numpy as np
import pytest
import torch

import dvqvae_amd  # noqa: F401
from dvqvae_amd import contact, ops, synth
from oracle import contact_oracle

DEV = "cuda:0"


def gpu(t):
    return t.to(DEV)


def sphere_mesh(n_lat=18, n_lon=43, radius=0.05):
    """Closed lat-long sphere, outward winding: V = n_lat * n_lon + 2 (776 for the defaults)."""
    th = np.linspace(0, np.pi, n_lat + 2)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    v = [[0, 0, radius]]
    for t in th:
        for p in ph:
            v.append([radius * np.sin(t) * np.cos(p), radius * np.sin(t) * np.sin(p), radius * np.cos(t)])
    v.append([0, 0, -radius])
    v = np.asarray(v, np.float32)
    f = []
    ring = lambda i, j: 1 + i * n_lon + (j % n_lon)
    for j in range(n_lon):
        f.append([0, ring(0, j), ring(0, j + 1)])
        f.append([len(v) - 1, ring(n_lat - 1, j + 1), ring(n_lat - 1, j)])
    for i in range(n_lat - 1):
        for j in range(n_lon):
            f.append([ring(i, j), ring(i + 1, j), ring(i + 1, j + 1)])
            f.append([ring(i, j), ring(i + 1, j + 1), ring(i, j + 1)])
    return v, np.asarray(f, np.int64)


def test_face_csr_lists_incident_faces_in_order():
    v, f = sphere_mesh(4, 6)
    faces, off, vf = contact.face_csr(f, len(v))
    assert off[-1] == 3 * len(f) and faces.dtype == np.int32
    for vert in range(len(v)):
        inc = vf[off[vert]:off[vert + 1]]
        assert list(inc) == sorted(inc)
        assert set(inc) == {i for i in range(len(f)) if vert in f[i]}


def test_oracle_nn_matches_float64_bruteforce_and_first_min():
    rng = np.random.default_rng(0)
    src = rng.normal(size=(2, 50, 3)).astype(np.float32)
    trg = rng.normal(size=(2, 40, 3)).astype(np.float32)
    trg[0, 7] = trg[0, 3]                                        # exact duplicate: the lower index must win
    src[0, 0] = trg[0, 3]
    d, i = contact_oracle.nn_points(src, trg)
    ref = ((src[:, :, None, :].astype(np.float64) - trg[:, None, :, :]) ** 2).sum(-1)
    assert np.array_equal(i, ref.argmin(-1)) and i[0, 0] == 3
    assert np.allclose(d, ref.min(-1), rtol=1e-6, atol=1e-12)
    src[1, 5, 1] = np.nan
    d, i = contact_oracle.nn_points(src, trg)
    assert i[1, 5] == 0 and np.isnan(d[1, 5])                    # NaN first, like torch.argmin


def test_oracle_normals_point_outward_and_interior_detects_inside_points():
    v, f = sphere_mesh()
    verts = v[None]
    n = contact_oracle.vertex_normals(verts, f)
    radial = v / np.linalg.norm(v, axis=1, keepdims=True)
    assert np.all((n[0] * radial).sum(-1) > 0.99)
    assert np.allclose(np.linalg.norm(n[0], axis=1), 1.0, atol=1e-5)
    obj = np.stack([0.5 * v[::7], 1.5 * v[::7]])[:, :, :]       # inside / outside the sphere
    hand = np.repeat(verts, 2, axis=0)
    nn = contact_oracle.nn_points(obj, hand)[1]
    inside = contact_oracle.interior(np.repeat(n, 2, axis=0), hand, obj, nn)
    assert inside[0].all() and not inside[1].any()


# ------------------------------------------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("B,N1,N2", [(1, 1, 1), (3, 300, 778), (2, 1024, 778), (2, 257, 4096)])
def test_nn_points_equals_oracle(B, N1, N2):
    src = synth.synthetic_normal((B, N1, 3), 21, f"nn/src/{B}/{N1}", 0.1)
    trg = synth.synthetic_normal((B, N2, 3), 21, f"nn/trg/{B}/{N2}", 0.1)
    if N2 > 8:
        trg[0, 5] = trg[0, 2]                                    # tie -> first index
        src[0, 0] = trg[0, 2]
    d, i = ops.nn_points(gpu(src), gpu(trg))
    od, oi = contact_oracle.nn_points(src.numpy(), trg.numpy())
    assert np.array_equal(i.cpu().numpy(), oi)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))


@pytest.mark.gpu
def test_nn_points_reads_channel_first_clouds_in_place_and_nan():
    cloud = synth.synthetic_clouds(3, 500, seed=5)              # [B,4,N] as the generation path holds it
    hand = synth.synthetic_normal((3, 778, 3), 22, "nn/hand", 0.05)
    hand[..., 2] -= 0.69
    cloud[1, 1, 17] = float("nan")
    obj = gpu(cloud)[:, :3].transpose(1, 2)                     # a view: strides (4N, 1, N)
    assert not obj.is_contiguous()
    d, i = ops.nn_points(obj, gpu(hand))
    od, oi = contact_oracle.nn_points(cloud[:, :3].transpose(1, 2).numpy(), hand.numpy())
    assert np.array_equal(i.cpu().numpy(), oi)
    assert np.array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))
    assert int(i[1, 17]) == 0 and torch.isnan(d[1, 17])
    with pytest.raises(RuntimeError):
        ops.nn_points(obj, gpu(synth.synthetic_normal((3, 5000, 3), 1, "nn/big")))


@pytest.mark.gpu
def test_normals_interior_and_grasp_proxies_equal_oracle():
    v, f = sphere_mesh()
    B = 4
    noise = synth.synthetic_normal((B, len(v), 3), 23, "nn/noise", 0.002).numpy()
    hand = (v[None] * np.array([1.0, 1.2, 0.8, 1.0], np.float32)[:, None, None] + noise).astype(np.float32)
    topo = contact.HandTopology(f, len(v), DEV)
    n = topo.normals(gpu(torch.from_numpy(hand)))
    on = contact_oracle.vertex_normals(hand, f)
    assert np.array_equal(n.cpu().numpy().view(np.uint32), on.view(np.uint32))
    obj = synth.synthetic_normal((B, 700, 3), 24, "nn/obj", 0.04).numpy().astype(np.float32)
    out = contact.grasp_proxies(topo, gpu(torch.from_numpy(hand)), gpu(torch.from_numpy(obj)))
    od, oi = contact_oracle.nn_points(obj, hand)
    oin = contact_oracle.interior(on, hand, obj, oi)
    assert np.array_equal(out["nn_idx"].cpu().numpy(), oi)
    assert np.array_equal(out["interior"].cpu().numpy(), oin)
    assert oin.any() and not oin.all()
    pen = np.where(oin, od, 0).astype(np.float64).sum(1)
    assert np.allclose(out["penetration"].cpu().numpy(), pen, rtol=1e-5)
    assert np.array_equal(out["n_interior"].cpu().numpy(), oin.sum(1))
    assert np.array_equal(out["n_contact"].cpu().numpy(), (od < 0.02 ** 2).sum(1))
