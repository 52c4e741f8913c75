"""Deterministic synthetic weights and inputs (there are no trained checkpoints and no datasets on
the build or GPU boxes: the reference's weights sit behind a Google-Drive link, README.md:4).

Every tensor is drawn from a numpy Philox stream keyed by (seed, crc32(tensor name)), so the same
bytes are regenerated on any machine with the same numpy, in any order, without the reference.
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, Mapping

import numpy as np
import torch

_BN = re.compile(r"(^|\.)bn\d+\.")


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def fill_like(name: str, ref: torch.Tensor, seed: int) -> torch.Tensor:
    """One synthetic tensor shaped/typed like ``ref`` with a distribution chosen by its name."""
    shape = tuple(ref.shape)
    g = _rng(seed, name)
    if name.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=ref.dtype)
    if _BN.search(name):
        leaf = name.rsplit(".", 1)[1]
        if leaf == "running_var":
            a = g.uniform(0.5, 1.5, size=shape)
        elif leaf == "weight":
            a = g.uniform(0.8, 1.2, size=shape)
        else:                                   # running_mean, bias
            a = g.normal(0.0, 0.1, size=shape)
    elif name.endswith("embedding.weight") or name.endswith("class_cond_embedding.weight"):
        a = g.normal(0.0, 0.5, size=shape)      # codebooks, token and class embeddings
    elif name.endswith(".bias"):
        a = g.uniform(-0.1, 0.1, size=shape)
    else:                                       # conv / linear weights: variance 1/fan_in
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        b = np.sqrt(3.0 / max(fan_in, 1))
        a = g.uniform(-b, b, size=shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(ref.dtype)


def synthetic_state_dict(template: Mapping[str, torch.Tensor], seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Synthetic values for every entry of ``template`` (a ``module.state_dict()``)."""
    return {k: fill_like(k, v, seed) for k, v in template.items()}


def synthetic_clouds(B: int, N: int, seed: int = 0, channels: int = 4) -> torch.Tensor:
    """Object clouds shaped like the datasets' tensors: ``[B, 4, N]`` f32, xyz ~ U(-0.1,0.1)^3 under a
    per-object random rotation, shifted by the reference's canonical offset
    (gen_diverse_grasp_ho3d.py:221), channel 3 = per-object constant bbox-diagonal "scale"
    (dataset/utils_HO3D_FPHA.py:75-84)."""
    g = _rng(seed, f"clouds/{B}/{N}")
    xyz = g.uniform(-0.1, 0.1, size=(B, N, 3))
    ang = g.uniform(0, 2 * np.pi, size=(B, 3))
    cx, sx = np.cos(ang[:, 0]), np.sin(ang[:, 0])
    cy, sy = np.cos(ang[:, 1]), np.sin(ang[:, 1])
    cz, sz = np.cos(ang[:, 2]), np.sin(ang[:, 2])
    one, zero = np.ones(B), np.zeros(B)
    Rx = np.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], 1).reshape(B, 3, 3)
    Ry = np.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], 1).reshape(B, 3, 3)
    Rz = np.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], 1).reshape(B, 3, 3)
    R = Rx @ Ry @ Rz
    xyz = np.einsum("bij,bnj->bni", R, xyz) + np.array([-0.0793, 0.0208, -0.6924])
    out = np.empty((B, channels, N), dtype=np.float32)
    out[:, :3] = xyz.transpose(0, 2, 1)
    if channels > 3:
        out[:, 3:] = g.uniform(0.05, 0.35, size=(B, 1, 1))
    return torch.from_numpy(out)


OBJECT_CODEBOOK = "vqvae6.vector_quantization.embedding.weight"


def feature_codebook(features: torch.Tensor, spread: float = 1.0) -> torch.Tensor:
    """Codebook rows = the given encoder features rounded to fp16 (exactly representable in fp32, so fixtures store them as
    float16).  Random N(0, 0.5) codebook rows are all about equally far from every PointNet feature of the synthetic
    clouds (the features of different clouds differ by ~0.02 per channel around a common mean), so every object would get
    the same code; rows that ARE features of distinct seed clouds make the in-path argmin a real decision.
    ``spread`` > 1 moves the rows away from their mean by that factor (wider top-2 gaps, fewer distinct codes in use)."""
    f = features.detach().float().cpu()
    if spread != 1.0:
        m = f.mean(0, keepdim=True)
        f = m + spread * (f - m)
    return f.half().float().contiguous()


def seed_clouds(K: int, N: int, seed: int = 4100, channels: int = 4) -> torch.Tensor:
    """The K clouds whose object-type features become the K rows of the object codebook (``feature_codebook``)."""
    return synthetic_clouds(K, N, seed=seed, channels=channels)


def diversify_object_codebook(net, sd: Dict[str, torch.Tensor], N: int, seed: int = 4100, spread: float = 2.0) -> Dict[str, torch.Tensor]:
    """Replace the object codebook of ``sd`` (already loaded into ``net``, which sits on its device in eval mode) by the
    net's OWN object-type features of K seed clouds (K = codebook rows) and load it back.  Used where no reference exists
    (benchmark network, smoke): the codebook is a free synthetic parameter, and the CPU oracle is then given the same ``sd``."""
    E = sd[OBJECT_CODEBOOK]
    dev = next(net.parameters()).device
    with torch.no_grad():
        feats = []
        clouds = seed_clouds(E.shape[0], N, seed=seed)
        for b0 in range(0, E.shape[0], 256):
            feats.append(net.obj_encoder_type(clouds[b0:b0 + 256].to(dev))[0].float().cpu())
    sd = dict(sd)
    sd[OBJECT_CODEBOOK] = feature_codebook(torch.cat(feats), spread)
    net.load_state_dict(sd)
    net.eval().to(dev)
    return sd


def synthetic_normal(shape, seed: int, name: str, scale: float = 1.0) -> torch.Tensor:
    a = _rng(seed, name).normal(0.0, scale, size=tuple(shape))
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synthetic_uniform(shape, seed: int, name: str, lo: float, hi: float) -> torch.Tensor:
    a = _rng(seed, name).uniform(lo, hi, size=tuple(shape))
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def exp1_noise(B: int, n_pos: int, n_out: int, seed: int) -> torch.Tensor:
    """Exp(1) race noise q[B, n_pos, n_out] for the prior's sampler (host-side generator; parity runs
    feed recorded noise)."""
    a = _rng(seed, f"exp1/{B}/{n_pos}/{n_out}").standard_exponential(size=(B, n_pos, n_out))
    a = np.maximum(a, np.finfo(np.float32).tiny)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
