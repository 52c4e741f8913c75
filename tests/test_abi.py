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
    assert lib.dvq_abi_version() == 6


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


def test_kernels_are_built_without_the_slp_vectorizer():
    """Round 3: compiler-formed packed-fp32 instructions (v_pk_mul_f32 with op_sel broadcasts) made pn_trunk_filter_kernel publish a
    wrong record about once per 1e6 (tile, channel) pairs.  The flag that removes them must stay in both builds,
    and the PointNet filter must compile to code without any packed-fp32 multiply / fma / move."""
    import os, shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "d-vqvae_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("CXXFLAGS")]
    assert flags and all("-fno-slp-vectorize" in ln for ln in flags), "csrc/Makefile: CXXFLAGS lost -fno-slp-vectorize"
    assert "$(CXXFLAGS) -DDVQ_DIAG" in mk, "the diagnostics build must use the same CXXFLAGS"
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S",
                        "--cuda-device-only", "-o", "-", "pointnet_filter.hip"], cwd=csrc, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad = [ln for ln in r.stdout.splitlines() if any(op in ln for op in ("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_mov_b32", "v_pk_add_f32"))]
    assert not bad, f"packed-fp32 instructions in pointnet_filter.hip: {bad[:3]}"
