"""CPU-side checks of the C-ABI library: it is built, loads, and exports every symbol include/dvq.h declares
(no compute calls: there is no GPU here)."""
import os
import re

from dvqvae_amd import _lib


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    text = open(_lib.HEADER).read()
    declared = set(re.findall(r"\b(dvq_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found in include/dvq.h"
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f"declared in dvq.h but not exported: {missing}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.dvq_abi_version() == 5


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.dvq_vq_argmin_workspace_bytes(65536, 512) >= 65536 * 4
    assert lib.dvq_pointnet_workspace_bytes(8, 1024) > 8 * 1024 * 192 * 4
    assert lib.dvq_pointnet_workspace_bytes(0, 1024) > 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from dvqvae_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.vq_argmin(torch.zeros(4, 32), torch.zeros(8, 32))
