"""Host mirror of the third-party ``mano`` layer the reference calls (``mano.load(model_path=..., model_type='mano',
use_pca=True, num_pca_comps=45, flat_hand_mean=True)``, gen_diverse_grasp_obman.py:355-360): same call
signature (``layer(betas=, global_orient=, hand_pose=, transl=).vertices``), arithmetic in dvq_mano_forward.
"parity unpinned": that package is not installed/vendored; the algorithm is the published smplx-style LBS."""
from __future__ import annotations

import pickle
import types
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing


class _ChumpyStub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


class _ManoUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "chumpy":
            return type(name, (_ChumpyStub,), {})
        return super().find_class(module, name)


def read_mano_pkl(path: str) -> Dict[str, np.ndarray]:
    """Arrays of MANO_RIGHT.pkl / MANO_LEFT.pkl without chumpy installed."""
    with open(path, "rb") as f:
        raw = _ManoUnpickler(f, encoding="latin1").load()
    sd = raw["shapedirs"]
    if isinstance(sd, _ChumpyStub):                       # chumpy Select over a flat Ch
        sd = np.asarray(sd.a.x).reshape(-1)[np.asarray(sd.idxs)]
    sd = np.asarray(sd, dtype=np.float64).reshape(778, 3, -1)
    jreg = raw["J_regressor"]
    jreg = np.asarray(jreg.todense()) if hasattr(jreg, "todense") else np.asarray(jreg)
    parents = np.asarray(raw["kintree_table"])[0].astype(np.int64)
    parents[0] = -1
    return {"v_template": np.asarray(raw["v_template"], np.float64), "shapedirs": sd[:, :, :10],
            "posedirs": np.asarray(raw["posedirs"], np.float64), "J_regressor": jreg.astype(np.float64),
            "weights": np.asarray(raw["weights"], np.float64),
            "hands_components": np.asarray(raw["hands_components"], np.float64),
            "hands_mean": np.asarray(raw["hands_mean"], np.float64), "parents": parents,
            "faces": np.asarray(raw["f"]).astype(np.int64)}


def synthetic_mano_arrays(seed: int = 7) -> Dict[str, np.ndarray]:
    """Deterministic MANO-shaped model (shapes, kinematic tree, convex skinning weights) for machines without
    MANO_RIGHT.pkl (the GPU box).  Must stay byte-identical to oracle/mano_oracle.synthetic_mano_arrays."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    parents = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], dtype=np.int64)
    out = {"v_template": rng.uniform(-0.09, 0.09, size=(778, 3)),
           "shapedirs": rng.normal(0, 0.004, size=(778, 3, 10)),
           "posedirs": rng.normal(0, 0.002, size=(778, 3, 135))}
    w = np.zeros((778, 16))
    for v in range(778):
        js = rng.choice(16, size=4, replace=False)
        ww = rng.uniform(0.05, 1.0, size=4)
        w[v, js] = ww / ww.sum()
    jr = np.zeros((16, 778))
    for j in range(16):
        vs = rng.choice(778, size=24, replace=False)
        ww = rng.uniform(0.1, 1.0, size=24)
        jr[j, vs] = ww / ww.sum()
    out.update(J_regressor=jr, weights=w, hands_components=rng.normal(0, 0.25, size=(45, 45)),
               hands_mean=rng.normal(0, 0.2, size=45), parents=parents, faces=np.zeros((1538, 3), dtype=np.int64))
    return out


class ManoLayer(nn.Module):
    def __init__(self, arrays: Dict[str, np.ndarray], use_pca: bool = True, num_pca_comps: int = 45,
                 flat_hand_mean: bool = True):
        super().__init__()
        if not use_pca:
            raise NotImplementedError("the grasp path calls MANO with use_pca=True (gen_diverse_grasp_obman.py:357)")
        self._packed = packing.PackedMano(arrays, flat_hand_mean=flat_hand_mean, n_comps=num_pca_comps)
        self.faces = self._packed.faces

    @classmethod
    def from_module(cls, layer) -> "ManoLayer":
        """Adopt the buffers of an already-loaded ``mano``/smplx layer object."""
        g = lambda n: getattr(layer, n).detach().cpu().double().numpy()
        posedirs = g("posedirs")                                    # [135, 2334] in smplx-style layers
        arrays = {"v_template": g("v_template"), "shapedirs": g("shapedirs")[:, :, :10],
                  "posedirs": posedirs.T.reshape(778, 3, 135), "J_regressor": g("J_regressor"),
                  "weights": g("lbs_weights"), "hands_components": g("hand_components"),
                  "hands_mean": g("pose_mean")[3:], "parents": np.asarray(layer.parents.cpu()),
                  "faces": np.asarray(getattr(layer, "faces", np.zeros((0, 3))))}
        return cls(arrays, flat_hand_mean=False)

    def forward(self, betas, global_orient=None, hand_pose=None, transl=None, **_):
        verts, joints = ops.mano_forward(self._packed, betas.contiguous(), hand_pose.contiguous(),
                                         None if global_orient is None else global_orient.contiguous(),
                                         None if transl is None else transl.contiguous(), want_joints=True)
        return types.SimpleNamespace(vertices=verts, joints=joints, betas=betas, global_orient=global_orient,
                                     hand_pose=hand_pose)

    def vertices_channel_major(self, betas, hand_pose):
        """[B,3,778] vertices for zero global pose: the layout the hand PointNet consumes (gen_net.py:116-120)."""
        return ops.mano_forward(self._packed, betas, hand_pose, channel_major=True)


def load(model_path: str, model_type: str = "mano", use_pca: bool = True, num_pca_comps: int = 45,
         batch_size: int = 1, flat_hand_mean: bool = True, **_) -> ManoLayer:
    """Same call shape as ``mano.load`` at gen_diverse_grasp_obman.py:355-360."""
    return ManoLayer(read_mano_pkl(model_path), use_pca=use_pca, num_pca_comps=num_pca_comps,
                     flat_hand_mean=flat_hand_mean)
