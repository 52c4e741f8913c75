"""CPU oracle for the MANO hand layer.  TEST INFRASTRUCTURE ONLY -- **parity unpinned**.

The reference calls the third-party ``mano`` pip package (git+https://github.com/otaheri/MANO, no
version pinned anywhere in the reference: there is no requirements/setup file) at
network/gen_net.py:116-118 and gen_diverse_grasp_obman.py:252-253,355-360
(``mano.load(model_path, model_type='mano', use_pca=True, num_pca_comps=45, flat_hand_mean=True)``).
That package is not installed in the build image and not vendored by the reference, and the
reference has no test that pins values at this boundary, so nothing can pin this file numerically.
It restates the published smplx-style linear-blend-skinning algorithm that package implements
(SURVEY.md Appendix E): PCA pose -> axis-angle, Rodrigues, shape + pose blend shapes, joint
regression, kinematic chain, skinning.  Self-consistency checks live in tests/test_mano.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import pickle
from typing import Dict

import numpy as np
import torch

Tensor = torch.Tensor

N_VERTS, N_JOINTS, N_BETAS, N_PCA = 778, 16, 10, 45


class _ChStub:
    """Stand-in for chumpy objects while unpickling MANO_*.pkl (chumpy is not installed)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {"_state": st})


class _Unpickler(pickle.Unpickler):
    def find_class(self, mod, name):
        if mod.startswith("chumpy"):
            return type(name, (_ChStub,), {})
        return super().find_class(mod, name)


def load_mano_pkl(path: str) -> Dict[str, np.ndarray]:
    """Read the arrays a MANO layer needs out of MANO_RIGHT.pkl / MANO_LEFT.pkl (float64 as stored)."""
    with open(path, "rb") as f:
        d = _Unpickler(f, encoding="latin1").load()
    sh = d["shapedirs"]
    if isinstance(sh, _ChStub):                      # chumpy Select(a=Ch(x=...), idxs=...)
        base = np.asarray(sh.a.x).reshape(-1)
        sh = base[np.asarray(sh.idxs)].reshape(sh.preferred_shape if hasattr(sh, "preferred_shape")
                                               and int(np.prod(sh.preferred_shape)) == len(sh.idxs)
                                               else (N_VERTS, 3, -1))
    sh = np.asarray(sh, dtype=np.float64).reshape(N_VERTS, 3, -1)[:, :, :N_BETAS]
    jr = d["J_regressor"]
    jr = np.asarray(jr.todense()) if hasattr(jr, "todense") else np.asarray(jr)
    parents = np.asarray(d["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    return dict(
        v_template=np.asarray(d["v_template"], dtype=np.float64),
        shapedirs=sh,
        posedirs=np.asarray(d["posedirs"], dtype=np.float64),
        J_regressor=jr.astype(np.float64),
        weights=np.asarray(d["weights"], dtype=np.float64),
        hands_components=np.asarray(d["hands_components"], dtype=np.float64),
        hands_mean=np.asarray(d["hands_mean"], dtype=np.float64),
        parents=parents,
        faces=np.asarray(d["f"]).astype(np.int64),
    )


def rodrigues(rot_vecs: Tensor) -> Tensor:
    """Axis-angle [M,3] -> rotation matrices [M,3,3]; angle = ||r + 1e-8|| as in smplx lbs."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    axis = rot_vecs / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    rx, ry, rz = axis[:, 0:1], axis[:, 1:2], axis[:, 2:3]
    zero = torch.zeros_like(rx)
    K = torch.cat([zero, -rz, ry, rz, zero, -rx, -ry, rx, zero], dim=1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=rot_vecs.dtype).unsqueeze(0)
    return eye + s * K + (1 - c) * torch.bmm(K, K)


class ManoOracle:
    """flat_hand_mean=True, use_pca=True, num_pca_comps=45 MANO layer on CPU/fp32."""

    def __init__(self, arrays: Dict[str, np.ndarray], flat_hand_mean: bool = True, n_comps: int = N_PCA):
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32)
        self.v_template = t(arrays["v_template"])                               # [778,3]
        self.shapedirs = t(arrays["shapedirs"])                                 # [778,3,10]
        self.posedirs = t(np.asarray(arrays["posedirs"]).reshape(N_VERTS * 3, -1).T)   # [135, 2334]
        self.J_regressor = t(arrays["J_regressor"])                             # [16,778]
        self.weights = t(arrays["weights"])                                     # [778,16]
        self.comps = t(arrays["hands_components"][:n_comps])                    # [45,45]
        mean = np.zeros(45) if flat_hand_mean else arrays["hands_mean"]
        self.pose_mean = t(np.concatenate([np.zeros(3), mean]))                 # [48]
        self.parents = [int(p) for p in arrays["parents"]]

    def __call__(self, betas: Tensor, hand_pose: Tensor, global_orient: Tensor = None,
                 transl: Tensor = None, return_joints: bool = False):
        B = betas.shape[0]
        if global_orient is None:
            global_orient = torch.zeros(B, 3)
        pose_aa = hand_pose @ self.comps
        full_pose = torch.cat([global_orient, pose_aa], dim=1) + self.pose_mean
        v_shaped = self.v_template + torch.einsum("bl,mkl->bmk", betas, self.shapedirs)
        J = torch.einsum("bik,ji->bjk", v_shaped, self.J_regressor)             # [B,16,3]
        R = rodrigues(full_pose.reshape(-1, 3)).view(B, N_JOINTS, 3, 3)
        pose_feat = (R[:, 1:] - torch.eye(3)).reshape(B, -1)                    # [B,135]
        v_posed = v_shaped + (pose_feat @ self.posedirs).view(B, N_VERTS, 3)
        # kinematic chain
        rel = J.clone()
        for j in range(1, N_JOINTS):
            rel[:, j] = J[:, j] - J[:, self.parents[j]]
        T = torch.zeros(B, N_JOINTS, 4, 4)
        T[:, :, :3, :3] = R
        T[:, :, :3, 3] = rel
        T[:, :, 3, 3] = 1
        chain = [T[:, 0]]
        for j in range(1, N_JOINTS):
            chain.append(torch.matmul(chain[self.parents[j]], T[:, j]))
        G = torch.stack(chain, dim=1)                                           # [B,16,4,4]
        posed_joints = G[:, :, :3, 3]
        Jh = torch.cat([J, torch.zeros(B, N_JOINTS, 1)], dim=2).unsqueeze(-1)   # [B,16,4,1]
        corr = torch.matmul(G, Jh)                                              # [B,16,4,1]
        A = G.clone()
        A[:, :, :, 3:4] = A[:, :, :, 3:4] - corr
        Tv = torch.matmul(self.weights.unsqueeze(0).expand(B, -1, -1), A.view(B, N_JOINTS, 16)).view(B, N_VERTS, 4, 4)
        vh = torch.cat([v_posed, torch.ones(B, N_VERTS, 1)], dim=2).unsqueeze(-1)
        verts = torch.matmul(Tv, vh)[:, :, :3, 0]
        if transl is not None:
            verts = verts + transl.unsqueeze(1)
            posed_joints = posed_joints + transl.unsqueeze(1)
        return (verts, posed_joints) if return_joints else verts


def synthetic_mano_arrays(seed: int = 7) -> Dict[str, np.ndarray]:
    """A deterministic MANO-shaped model (same array shapes, same kinematic tree, sparse skinning
    weights that sum to one) for benches/tests on machines without MANO_RIGHT.pkl."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    parents = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], dtype=np.int64)
    v_template = rng.uniform(-0.09, 0.09, size=(N_VERTS, 3))
    shapedirs = rng.normal(0, 0.004, size=(N_VERTS, 3, N_BETAS))
    posedirs = rng.normal(0, 0.002, size=(N_VERTS, 3, 135))
    w = np.zeros((N_VERTS, N_JOINTS))
    for v in range(N_VERTS):
        js = rng.choice(N_JOINTS, size=4, replace=False)
        ww = rng.uniform(0.05, 1.0, size=4)
        w[v, js] = ww / ww.sum()
    jr = np.zeros((N_JOINTS, N_VERTS))
    for j in range(N_JOINTS):
        vs = rng.choice(N_VERTS, size=24, replace=False)
        ww = rng.uniform(0.1, 1.0, size=24)
        jr[j, vs] = ww / ww.sum()
    comps = rng.normal(0, 0.25, size=(45, 45))
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=jr,
                weights=w, hands_components=comps, hands_mean=rng.normal(0, 0.2, size=45),
                parents=parents, faces=np.zeros((1538, 3), dtype=np.int64))
