"""DVQVAE / Encoder / Decoder mirror (reference: network/DVQVAE.py:11-185).

Only the eval (encode) branch of ``DVQVAE.forward`` is implemented (:130-142); the training branch
(:100-129) is out of scope of the inference path.  The reference's ``forward`` raises NameError at HEAD
because ``f0hand`` is undefined (:93); here it is the 83 MANO vertices no other part list covers
(hand_parts.json, recorded by tools/extract_hand_parts.py) -- a documented assumption."""
import json
import os

import torch
import torch.nn as nn

from .. import ops, packing
from .pointnet_encoder import PointNetEncoder
from .VQVAE import VQVAE

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hand_parts.json")) as _f:
    HAND_PARTS = json.load(_f)["parts"]                       # f0hand..f4hand, handc


def _codebooks(mod, n_embeddings=128):
    for k in range(6):
        setattr(mod, f"vqvae{k}", VQVAE(h_dim=128, res_h_dim=32, n_res_layers=2, n_embeddings=n_embeddings,
                                        embedding_dim=256, beta=0.25, a=1))
    mod.vqvae6 = VQVAE(h_dim=128, res_h_dim=32, n_res_layers=2, n_embeddings=n_embeddings, embedding_dim=1024, beta=2, a=0)


class _MLP(nn.Module):
    def _planes(self, lin):
        """weight image of a Linear (packing.split_planes), rebuilt when the parameter is replaced, moved or edited in place"""
        w = lin.weight
        key = (w.data_ptr(), w._version, str(w.device), packing.gemm_kind())
        cache = self.__dict__.setdefault("_plane_cache", {})
        hit = cache.get(id(lin))
        if hit is None or hit[0] != key:
            hit = (key, packing.split_planes(w.detach()))
            cache[id(lin)] = hit
        return hit[1]

    def _run(self, x, final=None):
        layers = [m for m in self.MLP if isinstance(m, nn.Linear)] + ([final] if final is not None else [])
        if len(layers) == 3:                                   # Decoder (3 Linear) and Encoder (2 Linear + linear_means): one ABI call
            return ops.mlp3(x if x.is_contiguous() else x.contiguous(),
                            [(l.weight.detach(), l.bias.detach(), self._planes(l)) for l in layers])
        n = len(layers)
        for i, lin in enumerate(layers):
            x = ops.linear(x if x.is_contiguous() else x.contiguous(), lin.weight.detach(), lin.bias.detach(),
                           relu=i + 1 < n, planes=self._planes(lin))
        return x


class Encoder(_MLP):
    """(Linear+ReLU)* then linear_means (DVQVAE.py:145-166); linear_log_var is allocated and unused, as there."""

    def __init__(self, layer_sizes, latent_size):
        super().__init__()
        self.MLP = nn.Sequential()
        for i, (n_in, n_out) in enumerate(zip(layer_sizes[:-1], layer_sizes[1:])):
            self.MLP.add_module(f"L{i}", nn.Linear(n_in, n_out))
            self.MLP.add_module(f"A{i}", nn.ReLU())
        self.linear_means = nn.Linear(layer_sizes[-1], latent_size)
        self.linear_log_var = nn.Linear(layer_sizes[-1], latent_size)

    def forward(self, x):
        return self._run(x, final=self.linear_means)


class Decoder(_MLP):
    """Linear+ReLU ... Linear, no final activation (DVQVAE.py:169-185)."""

    def __init__(self, layer_sizes, latent_size):
        super().__init__()
        self.MLP = nn.Sequential()
        sizes = [latent_size] + list(layer_sizes)
        for i, (n_in, n_out) in enumerate(zip(sizes[:-1], sizes[1:])):
            self.MLP.add_module(f"L{i}", nn.Linear(n_in, n_out))
            if i + 1 < len(layer_sizes):
                self.MLP.add_module(f"A{i}", nn.ReLU())

    def forward(self, z):
        return self._run(z)


class DVQVAE(nn.Module):
    def __init__(self, obj_inchannel=4, n_embeddings=128):
        super().__init__()
        self.obj_inchannel = obj_inchannel
        self.handembnns = [Encoder([1024, 512], 256) for _ in range(6)]
        for i, m in enumerate(self.handembnns):
            self.add_module(f"emb_{i}", m)
        self.obj_encoder_type = PointNetEncoder(global_feat=True, feature_transform=False, channel=obj_inchannel)
        self.obj_encoder_pos = PointNetEncoder(global_feat=True, feature_transform=False, channel=obj_inchannel)
        self.hand_encoders = [PointNetEncoder(global_feat=True, feature_transform=False, channel=3) for _ in range(6)]
        for i, m in enumerate(self.hand_encoders):
            self.add_module(f"fing_{i}", m)
        _codebooks(self, n_embeddings)
        self.decoder = Decoder(layer_sizes=[1024, 256, 55], latent_size=2560)
        self.rh_mano = None
        self.recon_encoder = PointNetEncoder(global_feat=True, feature_transform=False, channel=3)
        self.pos_decoder = Decoder(layer_sizes=[1024, 128, 6], latent_size=2048)

    def set_rh_mano(self, Rh_mano):
        self.rh_mano = Rh_mano
        Rh_mano.eval()

    def forward(self, obj_pc, hand_xyz):
        """eval: obj_pc [B,4,N], hand_xyz [B,3,778] -> (emb_idx [7B,1] int64 ordered idx6,idx0..5, obj_emb [B,1024])"""
        if self.training:
            raise NotImplementedError("DVQVAE training forward (network/DVQVAE.py:100-129) is outside the inference path")
        hand = hand_xyz - hand_xyz.mean(dim=2, keepdim=True)                       # :48-51
        feat_type, _, _ = self.obj_encoder_type(obj_pc)
        idxs = []
        for enc, mlp, k, part in zip(self.hand_encoders, self.handembnns, range(6), HAND_PARTS):
            sel = torch.as_tensor(part, device=hand.device)
            f, _, _ = enc(hand.index_select(2, sel).contiguous())
            i_k, _ = getattr(self, f"vqvae{k}").inference(mlp(f))
            idxs.append(i_k)
        idx6, obj_emb = self.vqvae6.inference(feat_type)
        return torch.cat([idx6] + idxs, 0), obj_emb
