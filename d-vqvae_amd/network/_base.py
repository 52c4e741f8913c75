"""Packed-weight cache shared by the mirrored modules."""
import torch.nn as nn


class PackedModule(nn.Module):
    """nn.Module whose HIP-side weight image is built lazily and dropped whenever the parameters are
    re-loaded or moved (``load_state_dict``, ``.to()``, ``.cuda()``).  In-place edits of parameters need an
    explicit ``repack()``."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_packed", None)

    def repack(self):
        object.__setattr__(self, "_packed", None)
        for m in self.children():
            if isinstance(m, PackedModule):
                m.repack()
        return self

    def _load_from_state_dict(self, *args, **kwargs):
        object.__setattr__(self, "_packed", None)
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        object.__setattr__(self, "_packed", None)
        return super()._apply(fn, *args, **kwargs)

    def _pack(self):
        raise NotImplementedError

    def packed(self):
        if self._packed is None:
            object.__setattr__(self, "_packed", self._pack())
        return self._packed
