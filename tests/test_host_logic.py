"""CPU-only checks of the host logic around the C ABI: packing layouts, state_dict compatibility, sharding
arithmetic, rotations, synthetic-data determinism, JSON/diversity helpers."""
import json
import os

import numpy as np
import pytest
import torch

from dvqvae_amd import dist, diversity, generate, packing, synth
from dvqvae_amd import mano as dmano
from oracle import dvq_oracle as O
from oracle import mano_oracle


def test_gate_perm_is_a_permutation_and_pairs_channels():
    for dim in (64, 128, 512):
        P = packing.gate_perm(dim)
        assert sorted(P.tolist()) == list(range(2 * dim))
        # packed positions p (tanh) and p+32 (sigmoid) of every 64-block hold channels c and c+dim
        for blk in range(0, 2 * dim, 64):
            a, b = P[blk: blk + 32], P[blk + 32: blk + 64]
            assert torch.equal(b, a + dim) and int(a.max()) < dim


def test_fold_bn_matches_batchnorm():
    g = torch.Generator().manual_seed(0)
    w, b = torch.randn(8, 5, 1, generator=g), torch.randn(8, generator=g)
    sd = {"bn.weight": torch.rand(8, generator=g) + 0.5, "bn.bias": torch.randn(8, generator=g),
          "bn.running_mean": torch.randn(8, generator=g), "bn.running_var": torch.rand(8, generator=g) + 0.5}
    wf, bf = packing.fold_bn(w, b, sd, "bn")
    x = torch.randn(3, 5, 7, generator=g)
    ref = torch.nn.functional.batch_norm(torch.nn.functional.conv1d(x, w, b), sd["bn.running_mean"], sd["bn.running_var"],
                                         sd["bn.weight"], sd["bn.bias"], False, 0.0, 1e-5)
    got = torch.nn.functional.conv1d(x, wf.view(8, 5, 1), bf)
    assert torch.allclose(got, ref, atol=1e-5)


def test_state_dict_layout_matches_survey_appendix_b():
    from dvqvae_amd.network.gen_net import GenNet
    from dvqvae_amd.network.DVQVAE import DVQVAE
    sd = GenNet().state_dict()
    assert len(sd) == 333
    assert tuple(sd["obj_encoder_type.stn.conv1.weight"].shape) == (64, 4, 1)
    assert tuple(sd["recon_encoder.conv1.weight"].shape) == (64, 3, 1)
    assert tuple(sd["vqvae6.vector_quantization.embedding.weight"].shape) == (128, 1024)
    assert tuple(sd["decoder.MLP.L0.weight"].shape) == (1024, 2560)
    assert tuple(sd["pos_decoder.MLP.L2.weight"].shape) == (6, 128)
    assert tuple(sd["GatedPixelCNN.layers.0.vert_stack.weight"].shape) == (1024, 512, 3, 5)
    assert tuple(sd["GatedPixelCNN.layers.7.horiz_stack.weight"].shape) == (1024, 512, 1, 2)
    assert tuple(sd["GatedPixelCNN.output_conv.2.weight"].shape) == (512, 2048, 1, 1)
    dsd = DVQVAE().state_dict()
    assert "emb_3.linear_log_var.weight" in dsd and "fing_5.stn.fc3.bias" in dsd and "GatedPixelCNN.embedding.weight" not in dsd


def test_state_dict_layout_equals_the_reference_key_for_key():
    """tests/golden/g0_state_dict_layout.json was dumped from the imported reference (tools/make_golden.py layout):
    every key, shape and dtype of the modules a checkpoint is loaded into must be identical, in the same order."""
    import json
    import os
    from dvqvae_amd.network.DVQVAE import DVQVAE
    from dvqvae_amd.network.VQVAE import VQVAE
    from dvqvae_amd.network.gen_net import GenNet
    from dvqvae_amd.network.pixelcnn.models import GatedPixelCNN
    from dvqvae_amd.network.pointnet_encoder import PointNetEncoder
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g0_state_dict_layout.json")))
    mods = {"GenNet": GenNet(), "DVQVAE": DVQVAE(obj_inchannel=4), "GatedPixelCNN": GatedPixelCNN(512, 512, 15, 128),
            "PointNetEncoder4": PointNetEncoder(channel=4), "VQVAE": VQVAE(0, 0, 0, 128, 256, 0.25)}
    for name, mod in mods.items():
        mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in mod.state_dict().items()]
        assert mine == ref[name], f"{name}: state_dict layout differs from the reference"


def test_shard_range_covers_everything_once():
    for total in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_rotation_and_object_tensor():
    ang = np.array([[0.3, -1.2, 2.0], [0, 0, 0]])
    R = generate.rotation_xyz(ang)
    assert np.allclose(R @ R.transpose(0, 2, 1), np.eye(3), atol=1e-12) and np.allclose(np.linalg.det(R), 1)
    assert np.allclose(R[1], np.eye(3))
    pts = np.random.default_rng(0).uniform(-1, 1, size=(50, 3))
    t = generate.object_tensor(pts)
    assert tuple(t.shape) == (4, 50) and torch.all(t[3] == t[3, 0]) and abs(float(t[3, 0]) - np.linalg.norm(pts.max(0) - pts.min(0))) < 1e-6


def test_synthetic_data_is_deterministic():
    a = synth.synthetic_clouds(3, 17, seed=5)
    assert torch.equal(a, synth.synthetic_clouds(3, 17, seed=5)) and not torch.equal(a, synth.synthetic_clouds(3, 17, seed=6))
    t = {"x.weight": torch.zeros(4, 3), "bn1.running_var": torch.zeros(4), "bn1.num_batches_tracked": torch.zeros((), dtype=torch.int64)}
    s1, s2 = synth.synthetic_state_dict(t, 1), synth.synthetic_state_dict(t, 1)
    assert all(torch.equal(s1[k], s2[k]) for k in t) and float(s1["bn1.running_var"].min()) >= 0.5
    q = synth.exp1_noise(2, 9, 8, seed=0)
    assert float(q.min()) > 0


def test_mano_readers_agree():
    path = "/root/reference/models/mano/MANO_RIGHT.pkl"
    if not os.path.exists(path):
        pytest.skip("MANO_RIGHT.pkl only exists in the build container")
    a, b = dmano.read_mano_pkl(path), mano_oracle.load_mano_pkl(path)
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "hands_components", "hands_mean", "parents"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert a["parents"].tolist() == [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]
    # self-consistency of the oracle LBS on the real model: zero pose/shape -> template; regressed joints
    o = mano_oracle.ManoOracle(b)
    v, j = o(torch.zeros(1, 10), torch.zeros(1, 45), return_joints=True)
    assert torch.allclose(v[0], torch.tensor(b["v_template"], dtype=torch.float32), atol=1e-6)
    assert torch.allclose(j[0], torch.tensor(b["J_regressor"] @ b["v_template"], dtype=torch.float32), atol=1e-6)
    assert np.allclose(b["weights"].sum(1), 1.0)


def test_diversity_and_json_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    params = np.concatenate([rng.normal(c, 0.01, size=(30, 61)) for c in range(4)])
    ent, d = diversity.diversity(params, cls_num=4)
    assert abs(ent - np.log(4)) < 1e-6 and d < 0.2
    p = tmp_path / "obj_id_0.json"
    json.dump({"recon_params": [[row] for row in params.tolist()], "R_list": [], "trans_list": [], "r_list": []}, open(p, "w"))
    assert np.allclose(diversity.load_params([str(p)]), params)


def test_oracle_assemble61_layout():
    recon, pos = torch.arange(55.0).view(1, 55), 100 + torch.arange(6.0).view(1, 6)
    out = O.assemble61(recon, pos)[0]
    assert out[:10].tolist() == list(range(10)) and out[10:13].tolist() == [100, 101, 102]
    assert out[13:58].tolist() == list(range(10, 55)) and out[58:].tolist() == [103, 104, 105]


def test_bench_quotes_pmc_traffic_only_from_this_trees_kernels(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` is looked up in the newest committed profiles/*_pmc_hbm_traffic.json: only while that file carries
    the digest of THIS tree's kernel sources (tools/collect_profiles.sh stores `bench.py --sources-digest` beside the counters) and,
    with a git history at hand, a commit that is HEAD or an ancestor of it; otherwise traffic is None and the source says why."""
    import importlib.util
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    digest = bench.kernel_sources_sha256()
    assert len(digest) == 64 and digest == bench.kernel_sources_sha256()
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    ok, why = bench.pmc_file_is_current({"kernel_sources_sha256": digest, "commit": head or None})
    assert ok and why == ""
    ok, why = bench.pmc_file_is_current({"kernel_sources_sha256": "0" * 64, "commit": head or None})
    assert not ok and "digest differs" in why
    ok, why = bench.pmc_file_is_current({"commit": head or None})
    assert not ok and "no digest" in why
    if head:
        ok, why = bench.pmc_file_is_current({"kernel_sources_sha256": digest, "commit": "0123456789abcdef0123456789abcdef01234567"})
        assert not ok and "ancestor" in why
    # through pmc_traffic(): a profiles directory whose newest file is stale -> (None, "refused: ...")
    prof = tmp_path / "profiles"
    prof.mkdir()
    rows = [{"kernel": "void (anonymous namespace)::vq_stream16_kernel<0>(float const*", "launches": 10, "hbm_bytes_corrected": 7.0e7, "scope": "vq microbench"}]
    (prof / "r99_v1_pmc_hbm_traffic.json").write_text(json.dumps({"collected": "tag r99_v1", "commit": head or None, "kernel_sources_sha256": "f" * 64,
                                                                    "per_launch_bytes": rows}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_sha256", lambda: digest)
    val, src = bench.pmc_traffic("vq_fast", 65536)
    assert val is None and src.startswith("refused:") and "stale" in src
    (prof / "r99_v2_pmc_hbm_traffic.json").write_text(json.dumps({"collected": "tag r99_v2", "commit": None, "kernel_sources_sha256": digest,
                                                                    "per_launch_bytes": rows}))
    val, src = bench.pmc_traffic("vq_fast", 65536)
    assert val == 7.0e7 and "r99_v2" in src
