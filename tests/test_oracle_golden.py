"""Pin the CPU oracle (oracle/dvq_oracle.py) against vectors produced by the REAL reference
(tools/make_golden.py imported /root/reference in the build container).  CPU only.

Floats: the oracle uses the same ATen CPU ops as the reference, so it is bit-identical in the build
container; another host CPU may pick different oneDNN/MKL kernels, hence a small tolerance here.
Indices / codes: exact."""
import numpy as np
import pytest
import torch

from conftest import SEED, dvqvae_state_dict, gen_state_dict
from dvqvae_amd import synth
from oracle import dvq_oracle as O
from oracle import mano_oracle

ATOL, RTOL = 2e-5, 1e-5


def close(a, b, atol=ATOL, rtol=RTOL):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), atol=atol, rtol=rtol)


def pointnet_template(C):
    shapes = {}
    for pre in ("stn.", ""):
        for i, (o, n) in enumerate([(64, C), (128, 64), (1024, 128)], 1):
            shapes[f"{pre}conv{i}.weight"] = (o, n, 1)
            shapes[f"{pre}conv{i}.bias"] = (o,)
        bns = [64, 128, 1024] + ([512, 256] if pre else [])
        for i, n in enumerate(bns, 1):
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                shapes[f"{pre}bn{i}.{leaf}"] = (n,)
            shapes[f"{pre}bn{i}.num_batches_tracked"] = ()
    for i, (o, n) in enumerate([(512, 1024), (256, 512), (9, 256)], 1):
        shapes[f"stn.fc{i}.weight"] = (o, n)
        shapes[f"stn.fc{i}.bias"] = (o,)
    return {k: (torch.zeros(v, dtype=torch.int64) if k.endswith("tracked") else torch.zeros(v)) for k, v in shapes.items()}


@pytest.mark.parametrize("C,N,B", [(4, 64, 4), (4, 1024, 1), (4, 1024, 4), (4, 3000, 1), (3, 778, 4), (3, 100, 2)])
def test_pointnet(golden, C, N, B):
    g = golden("g1_pointnet")
    sd = synth.synthetic_state_dict(pointnet_template(C), SEED + C)
    x = synth.synthetic_clouds(B, N, seed=100 + N, channels=C)
    feat, trans = O.pointnet_encode(sd, "", x)
    tag = f"C{C}_N{N}_B{B}"
    close(feat, g[tag + "_feat"])
    close(trans, g[tag + "_trans"])
    if N <= 100:
        assert np.array_equal(x.numpy(), g[tag + "_x"])


@pytest.mark.parametrize("K,D", [(128, 256), (128, 1024), (512, 256)])
def test_vq_inference(golden, K, D):
    g = golden("g2_vq")
    E = synth.synthetic_normal((K, D), SEED, f"vq/E/{K}/{D}")
    for M in (1, 7, 4096):
        z = synth.synthetic_normal((M, D), SEED, f"vq/z/{K}/{D}/{M}")
        idx, zq = O.vq_inference(E, z)
        tag = f"K{K}_D{D}_M{M}"
        safe = g[tag + "_gap"] > 1e-3            # rows whose fp64 top-2 gap is far above fp32 noise
        assert safe.mean() > 0.99
        assert np.array_equal(idx.squeeze(1).numpy()[safe], g[tag + "_idx"][safe])
        assert torch.equal(zq, E[idx.squeeze(1)])
        close(zq.double().sum(1).float()[torch.from_numpy(safe)], g[tag + "_zq_rowsum"][safe], atol=1e-4)
    Eu = synth.synthetic_uniform((K, D), SEED, f"vq/Eu/{K}/{D}", -1.0 / K, 1.0 / K)
    z = synth.synthetic_normal((512, D), SEED, f"vq/zu/{K}/{D}")
    idx, _ = O.vq_inference(Eu, z)
    safe = g[f"K{K}_D{D}_uinit_gap"] > 1e-3
    assert np.array_equal(idx.squeeze(1).numpy()[safe], g[f"K{K}_D{D}_uinit_idx"][safe])


def test_vq_crafted_and_train(golden):
    g = golden("g2_vq")
    E, z = torch.from_numpy(g["crafted_E"]), torch.from_numpy(g["crafted_z"])
    idx, _ = O.vq_inference(E, z)
    assert idx.squeeze(1).tolist() == g["crafted_idx"].tolist()
    assert idx[0, 0] == 5 and idx[2, 0] == 0          # exact tie -> lowest index; NaN row -> 0
    zt = synth.synthetic_normal((64, 256), SEED, "vq/z/train")
    loss, zq, perp, onehot, idx = O.vq_train_forward(E, zt, beta=0.25, al=1)
    close(loss, g["train_loss"], atol=1e-6)
    close(perp, g["train_perplexity"], atol=1e-4)
    close(zq.double().sum(1).float(), g["train_zq_rowsum"], atol=1e-4)
    with pytest.raises(RuntimeError):
        O.vq_lookup(E, torch.tensor([128]))


def pixelcnn_template(input_dim, dim, n_layers, n_classes):
    t = {"embedding.weight": (input_dim, dim)}
    for i in range(n_layers):
        k = 5 if i == 0 else 3
        p = f"layers.{i}."
        t[p + "class_cond_embedding.weight"] = (n_classes, 2 * dim)
        t[p + "vert_stack.weight"] = (2 * dim, dim, k // 2 + 1, k)
        t[p + "vert_stack.bias"] = (2 * dim,)
        t[p + "vert_to_horiz.weight"] = (2 * dim, 2 * dim, 1, 1)
        t[p + "vert_to_horiz.bias"] = (2 * dim,)
        t[p + "horiz_stack.weight"] = (2 * dim, dim, 1, k // 2 + 1)
        t[p + "horiz_stack.bias"] = (2 * dim,)
        t[p + "horiz_resid.weight"] = (dim, dim, 1, 1)
        t[p + "horiz_resid.bias"] = (dim,)
    t["output_conv.0.weight"], t["output_conv.0.bias"] = (2048, dim, 1, 1), (2048,)
    t["output_conv.2.weight"], t["output_conv.2.bias"] = (input_dim, 2048, 1, 1), (input_dim,)
    return {k: torch.zeros(v) for k, v in t.items()}


def test_pixelcnn_small(golden):
    g = golden("g4_pixelcnn")
    sd = synth.synthetic_state_dict(pixelcnn_template(32, 64, 3, 16), SEED + 1)
    x, lab = torch.from_numpy(g["small_x"]), torch.from_numpy(g["small_label"])
    close(O.pixelcnn_forward(sd, "", x, lab), g["small_logits"])
    codes = O.pixelcnn_generate(sd, "", lab, synth.exp1_noise(5, 9, 32, seed=5))
    assert np.array_equal(codes.numpy(), g["small_codes"])


def test_pixelcnn_full(golden):
    g = golden("g4_pixelcnn")
    sd = synth.synthetic_state_dict(pixelcnn_template(512, 512, 15, 128), SEED + 2)
    x, lab = torch.from_numpy(g["full_x"]), torch.from_numpy(g["full_label"])
    close(O.pixelcnn_forward(sd, "", x, lab), g["full_logits"])
    q = synth.exp1_noise(4, 9, 512, seed=6)
    labs = torch.from_numpy(g["full_gen_label"])
    codes = torch.cat([O.pixelcnn_generate(sd, "", labs[b:b + 1], q[b:b + 1]) for b in range(2)])
    assert np.array_equal(codes.numpy(), g["full_codes"][:2])


def test_decoders(golden):
    g = golden("g6_decoder")
    for tag, sizes, lat in [("dec", [1024, 256, 55], 2560), ("pos", [1024, 128, 6], 2048)]:
        t, n_in = {}, lat
        for i, n in enumerate(sizes):
            t[f"MLP.L{i}.weight"], t[f"MLP.L{i}.bias"] = torch.zeros(n, n_in), torch.zeros(n)
            n_in = n
        sd = synth.synthetic_state_dict(t, SEED + 3)
        z = synth.synthetic_normal((5, lat), SEED, f"dec/z/{tag}")
        close(O.mlp_decoder(sd, "", z), g[tag + "_y"])


def _gennet_sd(g7):
    from dvqvae_amd.network.gen_net import GenNet
    return gen_state_dict(GenNet().state_dict(), g7)


def test_gen_end_to_end_vs_reference(golden):
    """G7 / BASELINE config 1: oracle.gen batched over 8 objects == 8 reference GenNet.gen(B=1) calls."""
    g = golden("g7_gen")
    sd = _gennet_sd(g)
    mano = mano_oracle.ManoOracle(mano_oracle.synthetic_mano_arrays())
    obj = synth.synthetic_clouds(8, int(g["n_points"]), seed=int(g["cloud_seed"]))
    q = synth.exp1_noise(8, 9, 512, seed=int(g["noise_seed"]))
    with torch.no_grad():
        recon, pos, aux = O.gen(sd, obj, q, mano, return_aux=True)
    safe = g["idx6_gap"] > float(g["idx6_margin"])
    assert safe.sum() >= 6 and len(set(g["idx6"][safe].tolist())) >= 6, "the fixture must bite: distinct, well separated object codes"
    assert np.array_equal(aux["idx6"][:, 0].numpy()[safe], g["idx6"][safe])
    same = np.all(aux["codes"].numpy().reshape(8, 9) == g["codes"].reshape(8, 9), axis=1) & (aux["idx6"][:, 0].numpy() == g["idx6"])
    print(f"set aside by the gap check (fp64 top-2 distance gap of the object code <= {float(g['idx6_margin'])}): {(~safe).sum()} of 8")
    assert same[safe].all()
    close(recon[torch.from_numpy(same)], g["recon"][same])
    close(pos[torch.from_numpy(same)], g["recon_pos"][same])
    j = golden("g7_gen_juice")
    with torch.no_grad():
        r1, p1 = O.gen(sd, torch.from_numpy(j["obj_f16"].astype(np.float32)), q[:1], mano)
    close(r1, j["recon"])
    close(p1, j["recon_pos"])


def test_dvqvae_eval_vs_reference(golden):
    from dvqvae_amd.network.DVQVAE import DVQVAE, HAND_PARTS
    g = golden("g8_dvqvae")
    sd = dvqvae_state_dict(DVQVAE().state_dict(), g)
    obj = synth.synthetic_clouds(3, 512, seed=80)
    hand = synth.synthetic_normal((3, 3, 778), SEED, "dvq/hand", 0.05)
    with torch.no_grad():
        emb_idx, obj_emb = O.dvqvae_eval_forward(sd, obj, hand, HAND_PARTS)
    safe = g["emb_gap"] > float(g["emb_margin"])
    assert safe.sum() >= 18 and all(len(set(r.tolist())) == 3 for r in g["emb_idx"].reshape(7, 3)), "the fixture must bite"
    assert np.array_equal(emb_idx[:, 0].numpy()[safe], g["emb_idx"][safe])
    if safe[:3].all():
        close(obj_emb, g["obj_emb"], atol=0)
    assert HAND_PARTS[0] == O._thumb_vertices() and sorted(set(sum(HAND_PARTS, []))) == list(range(778))


def test_vq_canonical_vs_golden(golden):
    """The canonical-order C oracle (what the HIP kernels match bit for bit) against the reference's indices."""
    from oracle import vq_canonical
    g = golden("g2_vq")
    for K, D in [(128, 256), (128, 1024), (512, 256)]:
        E = synth.synthetic_normal((K, D), SEED, f"vq/E/{K}/{D}")
        z = synth.synthetic_normal((4096, D), SEED, f"vq/z/{K}/{D}/4096")[:512]
        idx, dmin = vq_canonical.argmin(z.numpy(), E.numpy())
        safe = g[f"K{K}_D{D}_M4096_gap"][:512] > 1e-3
        assert np.array_equal(idx[safe], g[f"K{K}_D{D}_M4096_idx"][:512][safe])
    E, z = g["crafted_E"], g["crafted_z"]
    idx, _ = vq_canonical.argmin(z, E)
    assert idx.tolist() == g["crafted_idx"].tolist()
