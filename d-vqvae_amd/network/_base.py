"""Packed-weight cache shared by the mirrored modules."""
import torch.nn as nn


class PackedModule(nn.Module):
    """nn.Module whose HIP-side weight image is built lazily and rebuilt whenever the state it was built from changes:
    ``load_state_dict`` / ``.to()`` / ``.cuda()`` (dropped eagerly), ``train()`` / ``eval()``, optimizer steps and any other
    in-place update that bumps a tensor's version counter (checked on every ``packed()`` call: the key is the training flag
    plus (data_ptr, _version) of every parameter and buffer).  Writes through ``tensor.data`` bypass the version counter:
    call ``repack()`` after those."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_packed", None)
        object.__setattr__(self, "_packed_key", None)

    def repack(self):
        object.__setattr__(self, "_packed", None)
        object.__setattr__(self, "_packed_key", None)
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

    def _state_key(self):
        ts = list(self.parameters()) + list(self.buffers())
        return (self.training,) + tuple((t.data_ptr(), t._version) for t in ts)

    def packed(self):
        key = self._state_key()
        if self._packed is None or self._packed_key != key:
            object.__setattr__(self, "_packed", self._pack())
            object.__setattr__(self, "_packed_key", key)
        return self._packed
