"""Loader (and in-tree builder) of libdvq_hip.so, the C-ABI library declared in include/dvq.h.

There is NO fallback: if the library cannot be built/loaded, every op raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdvq_hip.so")
if os.environ.get("DVQ_DIAG_LIB") == "1":          # tools/ only: the diagnostics build (`make -C csrc diag`), never the product
    LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "diag", "libdvq_hip_diag.so")
    print("[dvq] DVQ_DIAG_LIB=1: loading the DIAGNOSTICS build; results are invalid when a DVQ_*_ABL variable is set",
          file=__import__("sys").stderr)
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "dvq.h")

_lock = threading.Lock()
_lib = None

DVQ_MAX_SRC = 8
ABI_VERSION = 9            # DVQ_ABI_VERSION of include/dvq.h, which the struct mirrors below follow (tests/test_abi.py compares both with the library's)
PLANES_BF16X3, PLANES_F16X2 = 0, 1

c_f32p = C.c_void_p      # device pointers travel as integers
c_i64p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p


class GemmSrc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("ldx", C.c_int64), ("ldw", C.c_int64),
                ("K", C.c_int32), ("wp_kind", C.c_int32), ("wp", C.c_void_p), ("wp_plane", C.c_int64), ("w_scale", C.c_void_p)]


class MlpLayer(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("wp", C.c_void_p), ("n_out", C.c_int32), ("k_in", C.c_int32),
                ("w_scale", C.c_void_p), ("wp_kind", C.c_int32), ("_pad", C.c_int32)]


class PointnetWeights(C.Structure):
    _fields_ = [("C", C.c_int32), ("_pad", C.c_int32)] + [
        (n, C.c_void_p) for n in (
            "s_w1", "s_b1", "s_w2", "s_b2", "s_w3", "s_b3", "s_f1", "s_c1", "s_f2", "s_c2", "s_f3", "s_c3",
            "w1", "b1", "w2", "b2", "w3", "b3", "s_w2p", "s_w3p", "s_f1p", "s_f2p", "s_f3p", "w2p", "w3p",
            "s_w3f", "w3f")]


class PixelcnnLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wv", "bv", "wh", "wv2h", "bh", "cls", "wr", "br", "wv_p", "wh_p", "wv2h_p", "wr_p",
                                          "sv", "sh", "sr")]


class PixelcnnWeights(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dim", C.c_int32), ("n_in", C.c_int32), ("n_classes", C.c_int32),
                ("n_hidden", C.c_int32), ("planes_kind", C.c_int32), ("tok_emb", C.c_void_p),
                ("layers_host", C.POINTER(PixelcnnLayer)), ("w0", C.c_void_p), ("b0", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p), ("w0_p", C.c_void_p), ("w2_p", C.c_void_p),
                ("s0", C.c_void_p), ("s2", C.c_void_p), ("class_tables", C.c_void_p)]


class ManoModel(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("v_template", "blend_w", "blend_w_planes", "j_template", "j_shapedirs",
                                          "weights", "comps", "pose_mean")] + [
        ("parents", C.c_int32 * 16), ("blend_w_scale", C.c_void_p), ("planes_kind", C.c_int32), ("_pad", C.c_int32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("count", C.c_int64), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# name -> (restype, argtypes); kept in the order of include/dvq.h
SIGNATURES = {
    "dvq_abi_version": (C.c_int, []),
    "dvq_last_error": (C.c_char_p, []),
    "dvq_device_count": (C.c_int, []),
    "dvq_reload_env": (C.c_int, []),
    "dvq_linear": (C.c_int, [C.POINTER(GemmSrc), C.c_int, C.c_int64, C.c_int, c_f32p, C.c_int, c_f32p, C.c_int64, c_stream]),
    "dvq_mlp3_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "dvq_mlp3": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.POINTER(MlpLayer), c_f32p, C.c_int64, C.c_void_p, C.c_size_t, c_stream]),
    "dvq_split_bf16x3": (C.c_int, [c_f32p, C.c_int64, C.c_void_p, c_stream]),
    "dvq_f16x2_row_absmax": (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, c_stream]),
    "dvq_split_f16x2": (C.c_int, [c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, C.c_void_p, c_f32p, c_stream]),
    "dvq_vq_argmin_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "dvq_vq_argmin": (C.c_int, [c_f32p, C.c_int64, c_f32p, C.c_int64, C.c_int, C.c_int, c_i64p, c_f32p, C.c_void_p,
                                C.c_size_t, c_stream]),
    "dvq_vq_fast_supported": (C.c_int, [C.c_int, C.c_int]),
    "dvq_vq_pack_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dvq_vq_pack": (C.c_int, [c_f32p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, c_stream]),
    "dvq_vq_fast_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "dvq_vq_argmin_fast": (C.c_int, [c_f32p, c_f32p, C.c_void_p, C.c_int64, C.c_int, C.c_int, c_i64p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, c_stream]),
    "dvq_vq_lookup": (C.c_int, [c_f32p, c_i64p, C.c_int64, C.c_int64, C.c_int, C.c_int, c_f32p, C.c_int64, c_i32p, c_stream]),
    "dvq_pointnet_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "dvq_pointnet_fault_counters": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "dvq_pointnet_filter_bytes": (C.c_size_t, []),
    "dvq_pointnet_pack_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvq_pointnet_encode": (C.c_int, [C.POINTER(PointnetWeights), c_f32p, C.c_int64, C.c_int, c_f32p, C.c_int64, c_f32p,
                                      C.c_void_p, C.c_size_t, c_stream]),
    "dvq_pixelcnn_tables_bytes": (C.c_size_t, [C.POINTER(PixelcnnWeights)]),
    "dvq_pixelcnn_build_tables": (C.c_int, [C.POINTER(PixelcnnWeights), C.c_void_p, C.c_size_t, c_stream]),
    "dvq_pixelcnn_workspace_bytes": (C.c_size_t, [C.POINTER(PixelcnnWeights), C.c_int64]),
    "dvq_pixelcnn_sample": (C.c_int, [C.POINTER(PixelcnnWeights), c_i64p, c_f32p, C.c_int64, c_i64p, c_f32p, c_i32p,
                                      C.c_void_p, C.c_size_t, c_stream]),
    "dvq_pixelcnn_forward": (C.c_int, [C.POINTER(PixelcnnWeights), c_i64p, c_i64p, C.c_int64, c_f32p, c_i32p, C.c_void_p,
                                       C.c_size_t, c_stream]),
    "dvq_mano_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "dvq_mano_forward": (C.c_int, [C.POINTER(ManoModel), c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p, C.c_int64, c_f32p,
                                   C.c_int64, C.c_int64, c_f32p, C.c_int, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "dvq_copy_cols": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int, c_f32p, C.c_int64, c_stream]),
    "dvq_assemble61": (C.c_int, [c_f32p, c_f32p, C.c_int64, c_f32p, c_stream]),
    "dvq_prof_enable": (C.c_int, [C.c_int]),
    "dvq_prof_reset": (C.c_int, []),
    "dvq_prof_read": (C.c_int, [C.POINTER(ProfEntry), C.c_int]),
    "dvq_transform_cloud": (C.c_int, [c_f32p, C.c_int64, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, c_f32p, c_stream]),
    "dvq_exp1_noise": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int64, C.c_int64, C.c_int, c_f32p, c_stream]),
    "dvq_exp1_noise_rows": (C.c_int, [C.c_uint64, C.c_uint32, C.c_int64, c_i64p, C.c_int64, C.c_int, c_f32p, c_stream]),
    "dvq_probe_f16_subnormal": (C.c_int, [c_f32p, c_stream]),
    "dvq_nn_points": (C.c_int, [c_f32p, C.c_int64, C.c_int64, C.c_int64, c_f32p, C.c_int64, C.c_int64, C.c_int64,
                                C.c_int64, C.c_int, C.c_int, c_f32p, c_i64p, c_stream]),
    "dvq_vertex_normals": (C.c_int, [c_f32p, C.c_int64, C.c_int, c_i32p, c_i32p, c_i32p, c_f32p, c_stream]),
    "dvq_comm_available": (C.c_int, []),
    "dvq_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "dvq_comm_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "dvq_allgather_params": (C.c_int, [C.c_void_p, c_f32p, C.c_int64, C.c_int, c_f32p, c_stream]),
    "dvq_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "dvq_comm_destroy": (C.c_int, [C.c_void_p]),
    "dvq_interior": (C.c_int, [c_f32p, c_f32p, C.c_int, c_f32p, C.c_int64, C.c_int64, C.c_int64, c_i64p, C.c_int64,
                               C.c_int, C.c_void_p, c_stream]),
}


def build(force: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libdvq_hip.so (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    jobs = str(min(8, os.cpu_count() or 1))
    # "diag": the diagnostics twin (tools/diag/, -DDVQ_DIAG: timing ablations, phase stamps, fault injection for the run-time checks);
    # never loaded by the product path -- tests/test_gpu_parity.py::test_pointnet_runtime_checks loads it in a child process
    r = subprocess.run(["make", "-C", CSRC, "-j", jobs, "all", "diag"], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libdvq_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def load() -> C.CDLL:
    """dlopen the library and bind every symbol of include/dvq.h.  Raises if it is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # PyTorch owns the device memory and streams this library works on, and its wheel bundles its own HIP runtime
        # (libamdhip64 with the same SONAME as /opt/rocm's).  Whichever copy is loaded first serves the whole process:
        # load torch's first, or torch would run on the system runtime and see no devices.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C d-vqvae_amd/csrc`). "
                "There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        got = lib.dvq_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} implements ABI version {got}, these bindings mirror version {ABI_VERSION} of include/dvq.h: "
                               "rebuild the library (`make -C d-vqvae_amd/csrc`)")
        _lib = lib
        return lib


class DvqError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().dvq_last_error()
        raise DvqError(f"{what}: {msg.decode() if msg else 'error'} (status {rc})")
