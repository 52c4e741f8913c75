"""PointNetEncoder / STN3d mirror (reference: network/pointnet_encoder.py:10-45,125-169)."""
import torch
import torch.nn as nn

from .. import ops, packing
from ._base import PackedModule

_TRUNK = ((64, None), (128, 64), (1024, 128))          # (out, in) of conv1..3; conv1's in = channel


def _add_convs(mod: nn.Module, channel: int):
    for i, (o, n) in enumerate(_TRUNK, 1):
        setattr(mod, f"conv{i}", nn.Conv1d(channel if n is None else n, o, 1))


def _add_bns(mod: nn.Module, n_bn: int):
    for i, width in enumerate((64, 128, 1024, 512, 256)[:n_bn], 1):
        setattr(mod, f"bn{i}", nn.BatchNorm1d(width))


class STN3d(nn.Module):
    """Parameter container of the input-transform net (its arithmetic is fused into PointNetEncoder's op)."""

    def __init__(self, channel):
        super().__init__()
        _add_convs(self, channel)                         # registration order = the reference's (state_dict key order)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, 9)
        self.relu = nn.ReLU()
        _add_bns(self, 5)


class PointNetEncoder(PackedModule):
    def __init__(self, global_feat=True, feature_transform=False, channel=3):
        super().__init__()
        if feature_transform or not global_feat:
            raise NotImplementedError("the grasp path uses global_feat=True, feature_transform=False "
                                      "(every call site: gen_net.py:17-18,30, DVQVAE.py:18-20,35)")
        self.stn = STN3d(channel)
        _add_convs(self, channel)
        _add_bns(self, 3)
        self.global_feat, self.feature_transform = global_feat, feature_transform

    def _pack(self):
        if self.training:
            raise RuntimeError("PointNetEncoder: the HIP path folds eval-mode BatchNorm; call .eval() first")
        return packing.PackedPointNet(self.state_dict())

    def forward(self, x, out=None):
        """x [B,C,N] -> (feat [B,1024], trans [B,3,3], None)   (pointnet_encoder.py:140-169)"""
        feat, trans = ops.pointnet_encode(self.packed(), x.contiguous(), out=out)
        return feat, trans, None
