"""CPU-side checks of the C-ABI library: it is built, loads, and exports every symbol include/dvq.h declares
(no compute calls: there is no GPU here)."""
import os
import sys
import re

import pytest

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
    header_version = int(re.search(r"#define\s+DVQ_ABI_VERSION\s+(\d+)", text).group(1))
    assert lib.dvq_abi_version() == header_version == _lib.ABI_VERSION, "library, header and Python struct mirrors must be one ABI version"


def test_graft_entry_build_passes():
    """The driver's build check: compiles every HIP source, the oracle's C part, loads the library and imports the package."""
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    importlib.import_module("__graft_entry__").build()


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.dvq_vq_argmin_workspace_bytes(65536, 512) >= 65536 * 4
    assert lib.dvq_pointnet_workspace_bytes(8, 1024) > 8 * 1024 * 128 * 4      # the conv2 rows of the filtered trunk, at least
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
        pytest.skip("hipcc not found: the listing checks need the compiler")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S",
                        "--cuda-device-only", "-o", "-", "pointnet_filter.hip"], cwd=csrc, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    bad = [ln for ln in r.stdout.splitlines() if any(op in ln for op in ("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_mov_b32", "v_pk_add_f32"))]
    assert not bad, f"packed-fp32 instructions in pointnet_filter.hip: {bad[:3]}"


def test_no_packed_fp32_instructions_outside_the_gemm_epilogues():
    """Same fault, whole library: every kernel of the built libdvq_hip.so is disassembled (the gfx950 code objects of its fat
    binary) and searched for packed-fp32 multiply / fma / add / move.  Allowed only in the GEMM kernels, whose epilogues use
    explicit four-float vector arithmetic (scale, bias, residual); the PointNet, VQ, MANO, PixelCNN-draw and contact kernels
    must have none -- a compiler bump or a source change that brings them back fails here, on the CPU, before any GPU run."""
    import collections, re, shutil, struct, subprocess, tempfile
    objcopy, objdump = "/opt/rocm/lib/llvm/bin/llvm-objcopy", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objcopy) and os.path.exists(objdump)):
        return
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    hits, kernels = collections.defaultdict(collections.Counter), 0
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", _lib.LIB_PATH, os.path.join(tmp, "copy.so")], check=True, capture_output=True)
        data = open(fat, "rb").read()
        magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
        while True:
            j = data.find(magic, pos)
            if j < 0:
                break
            pos = j + 1
            n, off = struct.unpack_from("<Q", data, j + 24)[0], j + 32
            for _ in range(n):
                o, size, ln = struct.unpack_from("<QQQ", data, off)
                name = data[off + 24: off + 24 + ln].decode()
                off += 24 + ln
                if "gfx950" not in name or not size:
                    continue
                co = os.path.join(tmp, "dev.co")
                open(co, "wb").write(data[j + o: j + o + size])
                r = subprocess.run([objdump, "-d", "--mcpu=gfx950", co], capture_output=True, text=True, check=True)
                fn = None
                for line in r.stdout.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                    if m:
                        fn, kernels = m.group(1), kernels + 1
                        continue
                    m = re.search(r"\b(v_pk_(?:mul|fma|add|mov)_(?:f32|b32))\b", line)
                    if m:
                        hits[fn][m.group(1)] += 1
    assert kernels > 50, f"only {kernels} device functions found in {_lib.LIB_PATH}"
    bad = {fn: dict(c) for fn, c in hits.items() if "gemm_" not in fn}
    assert not bad, f"packed-fp32 instructions outside the GEMM kernels: {bad}"
    assert not any("pn_" in fn or "vq_" in fn for fn in hits), "PointNet / VQ kernels must stay free of packed-fp32 instructions"


def test_every_barrier_waits_for_the_waves_own_lds_operations():
    """Round 5: the barrier at the head of pn_trunk3_kernel's conv3 loop was compiled WITHOUT "s_waitcnt lgkmcnt(0)" although the
    back edge carries LDS stores -- waves passed it with stores in flight and tile records came out wrong whenever workgroups shared
    a CU (the shape of the round-3 fault; DESIGN.md 3.3, csrc/dvq_internal.h: dvq_lds_barrier).  Every HIP source is compiled to
    assembly with the library's flags and tools/check_barriers.py follows every path to every s_barrier: none may be reachable with an
    LDS operation of the wave possibly still pending."""
    import concurrent.futures, glob, shutil, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "d-vqvae_amd", "csrc")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found: the listing checks need the compiler")
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_barriers, check_hazards, check_waitcnt
    if not os.path.exists(os.path.join(csrc, "vq_pipe_loop.h")):         # generated, not tracked (the Makefile has the same rule)
        subprocess.run([sys.executable, os.path.join(root, "tools", "gen_vq_pipe.py")], check=True, capture_output=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only"]
    with tempfile.TemporaryDirectory() as tmp:
        def build(src):
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            r = subprocess.run([hipcc] + flags + ["-o", out, os.path.basename(src)], cwd=csrc, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            return out
        with concurrent.futures.ThreadPoolExecutor(max_workers=6) as ex:
            listings = list(ex.map(build, sorted(glob.glob(os.path.join(csrc, "*.hip")))))
        bad, barriers, hazards, mfmas, unwaited = [], 0, [], 0, []
        for path in listings:
            text = open(path).read()
            barriers += text.count("s_barrier")
            mfmas += text.count("v_mfma_")
            for name, body in check_barriers.functions(text):
                if "s_barrier" in body:
                    bad += [(os.path.basename(path), name[:80], hit) for hit in check_barriers.check(body)]
                # round 6: the same listings through tools/check_hazards.py -- producer / consumer pairs that need wait states the
                # hardware does not insert (the compiler does not look inside the inline-asm loops: vq_pipe_loop.h is 6 000 lines of it)
                hazards += [msg for _, _, msg in check_hazards.check_lines(body.splitlines(), strict=True, name=os.path.basename(path) + ":" + name[:60])]
                # ... and tools/check_waitcnt.py: no register is read or overwritten while a load into it may be in flight on some path
                # (s_waitcnt lgkmcnt / vmcnt re-derived from the listing: the compiler's own counts and the hand-counted ones of the asm loops)
                unwaited += check_waitcnt.check(body, os.path.basename(path) + ":" + name[:60])
    assert barriers > 100, f"only {barriers} s_barrier instructions found"
    assert not bad, f"s_barrier reachable with an LDS operation of the wave in flight: {bad}"
    assert mfmas > 1000 and not hazards, f"missing wait states: {hazards[:5]}"
    assert not unwaited, f"a register is used with a load possibly in flight: {unwaited[:5]}"


def test_waitcnt_checker_rules():
    """tools/check_waitcnt.py on small listings: in-order LDS / vector-memory counters, scalar reads out of order, joins, branches
    in the middle of a block, a load overwriting an older load's destination."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_waitcnt as cw
    run = lambda lines: cw.check("\n".join(lines), "t")
    use = "v_add_f32_e32 v3, v1, v1"
    assert len(run(["ds_read_b32 v1, v2", use])) == 1
    assert run(["ds_read_b32 v1, v2", "s_waitcnt lgkmcnt(0)", use]) == []
    assert run(["ds_read_b32 v1, v2", "ds_read_b32 v4, v2", "s_waitcnt lgkmcnt(1)", use]) == []                 # in order: the older one is done
    assert len(run(["ds_read_b32 v1, v2", "ds_read_b32 v4, v2", "s_waitcnt lgkmcnt(1)", "v_add_f32_e32 v3, v4, v4"])) == 1
    assert len(run(["ds_read_b32 v1, v2", "s_load_dword s4, s[0:1], 0x0", "ds_read_b32 v4, v2", "s_waitcnt lgkmcnt(1)", use])) == 1   # scalar reads return out of order
    assert run(["global_load_dword v1, v2, s[0:1]", "global_store_dword v2, v5, s[0:1]", "s_waitcnt vmcnt(1)", use]) == []
    assert len(run(["global_load_dword v1, v2, s[0:1]", "s_waitcnt lgkmcnt(0)", use])) == 1                      # the wrong counter
    assert len(run(["global_load_dword v1, v2, s[0:1]", "s_cbranch_scc1 .LBB0_2", ".LBB0_1:", "s_waitcnt vmcnt(0)", ".LBB0_2:", use])) == 1   # one path skips the wait
    assert run(["s_cbranch_scc1 .LBB0_2", "global_load_dword v1, v2, s[0:1]", "s_waitcnt vmcnt(0)", ".LBB0_2:", use]) == []      # the branch target sees the state AT the branch
    assert run(["global_load_dword v1, v2, s[0:1]", "global_load_dword v1, v6, s[0:1]", "s_waitcnt vmcnt(0)", use]) == []         # load over an older load's destination
    assert len(run(["global_load_dword v1, v2, s[0:1]", "v_mov_b32_e32 v1, 0"])) == 1                             # overwrite under a load in flight
    assert len(run([".LBB0_1:", use, "ds_read_b32 v1, v2", "s_cbranch_scc1 .LBB0_1"])) == 1                       # around a loop's back edge


def test_hazard_checker_rules():
    """tools/check_hazards.py on two-instruction listings: each rule fires without the wait states and is quiet with them."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_hazards as ch
    mfma = "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]"
    cases = [("R3", ["v_add_f32_e32 v1, v2, v3", "v_mov_b32_dpp v4, v1 row_shr:1 row_mask:0xf bank_mask:0xf"], "s_nop 1"),
             ("R1", ["v_readfirstlane_b32 s4, v1", "global_load_dword v5, v6, s[4:5]"], "s_nop 4"),
             ("R2", ["v_cmp_le_f32_e64 s[64:65], v1, v2", "v_readlane_b32 s3, v5, s64"], "s_nop 3"),
             ("R4", ["v_cmpx_le_f32_e32 v1, v2", "v_mov_b32_dpp v4, v9 row_shr:1"], "s_nop 4"),
             ("R5", ["v_sqrt_f32_e32 v1, v1", "v_mul_f32_e32 v2, v1, v3"], "s_nop 0"),
             ("R6", ["v_cvt_f16_f32_sdwa v1, v2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD", "v_add_f32_e32 v3, v1, v1"], "s_nop 0"),
             ("R7", [mfma, "v_add_f32_e32 v30, v0, v1"], "s_nop 10"),
             ("R7", [mfma, "ds_write_b32 v40, v3"], "s_nop 10"),
             ("R8", ["v_cmp_lt_f32_e32 vcc, v1, v2", "v_div_fmas_f32 v3, v4, v5, v6"], "s_nop 3"),
             ("R9", ["s_mov_b32 m0, s4", "buffer_load_dword v1, s[8:11], 0 offen lds"], "s_nop 0")]
    for rule, pair, nop in cases:
        assert [b[1] for b in ch.check_lines(pair)] == [rule], (rule, pair)
        assert ch.check_lines([pair[0], nop, pair[1]]) == [], (rule, nop)
        short = "s_nop %d" % (int(nop.split()[1]) - 1)
        if int(nop.split()[1]) > 0:
            assert [b[1] for b in ch.check_lines([pair[0], short, pair[1]])] == [rule], (rule, short)
    assert ch.check_lines([mfma, mfma]) == []                       # an accumulator chain is interlocked
    assert ch.check_lines(["v_add_f32_e32 v1, v2, v3", "L1:", "v_mov_b32_dpp v4, v1 row_shr:1"]) == []       # a label ends the window ...
    assert len(ch.check_lines(["v_add_f32_e32 v1, v2, v3", "L1:", "v_mov_b32_dpp v4, v1 row_shr:1"], strict=True)) == 1   # ... unless strict


def test_barrier_checker_sees_the_round5_shape():
    """tools/check_barriers.py on two synthetic listings: the loop of pn_trunk3_kernel as the compiler emitted it first (LDS stores on
    the back edge, no wait in front of the barrier at the loop head) must be flagged; the same with the wait must pass, and so must a
    trailing barrier in front of s_endpgm."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import check_barriers
    def listing(wait):
        return f"""
kern:                                   ; @kern
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\tds_write_b32 v0, v1
\ts_waitcnt lgkmcnt(0)
\ts_barrier
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
{wait}\ts_barrier
\tds_read_b128 v[4:7], v2
\ts_waitcnt lgkmcnt(0)
\tv_add_f32_e32 v4, v4, v5
\tds_write_b128 v3, v[4:7]
\ts_add_i32 s2, s2, 1
\ts_cmp_lg_u32 s2, 16
\ts_cbranch_scc1 .LBB0_1
\ts_waitcnt lgkmcnt(0)
\ts_barrier
\tds_read_b32 v0, v2
\ts_waitcnt lgkmcnt(0)
\tds_write_b32 v2, v0
\ts_barrier
\ts_endpgm
.Lfunc_end0:
"""
    bad = [check_barriers.check(body) for _, body in check_barriers.functions(listing(""))]
    assert bad == [[(".LBB0_1", 0)]], bad
    good = [check_barriers.check(body) for _, body in check_barriers.functions(listing("\ts_waitcnt lgkmcnt(0)\n"))]
    assert good == [[]], good
