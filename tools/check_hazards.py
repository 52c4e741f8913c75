#!/usr/bin/env python3
"""Static check of gfx950 listings (`hipcc -S --cuda-device-only`, or an inline-asm text) for the data hazards the hardware does NOT
interlock: the ones that need a number of independent instructions ("wait states": every instruction is one, `s_nop N` is N + 1)
between a producer and a consumer.  The compiler's hazard recognizer inserts them in code it generates -- it does not look inside
`asm volatile` blocks, which is where this library's hand-written loops live (vq_pipe_loop.h is one block of ~6 000 instructions), and
a miscompile of this kind is one candidate for the round-3 PointNet fault (timing dependent, gone after a change of code generation).

Rules (CDNA3 ISA guide section 4.5 "manually inserted wait states" and the gfx940 rows of LLVM's GCNHazardRecognizer):
  R1  VALU writes an SGPR / VCC            -> VMEM instruction reads that SGPR                      5
  R2  VALU writes an SGPR / VCC            -> v_readlane / v_writelane lane select                  4
  R3  VALU writes a VGPR                   -> DPP instruction reads that VGPR                       2
  R4  VALU writes EXEC                     -> DPP instruction                                       5
  R5  transcendental VALU writes a VGPR    -> non-transcendental VALU reads it                      1
  R6  VALU writes the HIGH half of a VGPR (SDWA dst_sel, op_sel dst bit, mixhi) -> VALU reads that VGPR  1
  R7  MFMA writes VGPRs                    -> VALU / LDS / VMEM reads or VALU overwrites them       passes-dependent: 4x4 5, 16x16 7 / 11
                                              (4 / 8 passes), 32x32 11 / 19 (8 / 16 passes)
  R8  VALU writes VCC                      -> v_div_fmas                                            4
  R9  SALU writes M0                       -> LDS DMA (buffer/global_load ... lds), s_sendmsg, ds_*_addtid 1
Within a basic block the scan is exact; at a label the history is cleared when `--strict` is not given (a branch costs more than any
of these counts), with `--strict` it is kept across fall-through edges.

    check_hazards.py listing.s [...] [--strict]     exit code 1 if a rule is violated
"""
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
DPP_MARK = re.compile(r"\b(quad_perm|row_shl|row_shr|row_ror|row_rol|row_mirror|row_half_mirror|row_bcast|row_newbcast|row_share|row_xmask|wave_shl|wave_shr|wave_rol|wave_ror|dpp8)\b")
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|vcc_lo|vcc_hi|exec|exec_lo|exec_hi|m0)\b")


def regs(tok):
    """set of ('v', n) / ('s', n) / ('vcc',) ... named by one operand text"""
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for n in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), n))
        else:
            g = m.group(6)
            out.add(("vcc",) if g.startswith("vcc") else ("exec",) if g.startswith("exec") else ("m0",))
    return out


def split_operands(line):
    parts = line.split(None, 1)
    op = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    # modifiers after the operands (offset:, dpp controls, op_sel:[..] ...) are kept in `mods`
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    mods = ""
    if ops:
        last = ops[-1].split(None, 1)
        if len(last) > 1:
            ops[-1], mods = last[0], last[1]
    return op, ops, mods


def mfma_passes(op):
    if not op.startswith("v_mfma") and not op.startswith("v_smfmac"):
        return 0
    if "_4x4x" in op:
        return 2
    if "_16x16x" in op:
        return 8 if ("f64" in op or "x4_" in op or "xf32" in op) else 4
    if "_32x32x" in op:
        # 32x32x16 f16/bf16, 32x32x64 f8f6f4: 8 passes on gfx950; 32x32x8 (gfx942 shapes), x2 f32: 16
        return 8 if re.search(r"_32x32x(16_|64_|32_i8|32_fp8|32_bf8)", op) else 16
    return 8


MFMA_WAIT = {2: 5, 4: 7, 8: 11, 16: 19}


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def check_lines(lines, strict=False, name="?"):
    """lines: instruction texts (labels as 'X:').  Returns a list of (index, rule, text)."""
    hist = []          # newest last: dicts {age, kind, op, dst, half, trans, passes}
    bad = []

    def advance(n):
        for h in hist:
            h["age"] += n
        while hist and hist[0]["age"] > 24:
            hist.pop(0)

    for i, raw in enumerate(lines):
        line = raw.split(";")[0].split("//")[0].strip()
        if not line:
            continue
        if line.endswith(":"):
            if not strict:
                hist.clear()
            continue
        if line.startswith("."):
            continue
        op, ops, mods = split_operands(line)
        kind = classify(op)
        if op == "s_nop":
            advance(int(ops[0], 0) + 1 if ops else 1)
            continue
        if op in ("s_branch",) or op.startswith("s_cbranch") or op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            advance(1)
            if not strict:
                hist.clear()
            continue
        is_dpp = bool(DPP_MARK.search(line)) or op.endswith("_dpp")
        # ---- sources / destination
        dst, src = set(), set()
        if kind in ("valu", "mfma"):
            if op.startswith("v_cmpx"):
                dst = {("exec",)} | (regs(ops[0]) if len(ops) == 3 else set())
                src = set().union(*[regs(o) for o in ops[-2:]]) if ops else set()
            elif op.startswith("v_cmp") and (op.endswith("_e32") or len(ops) == 2):
                dst = {("vcc",)}
                src = set().union(*[regs(o) for o in ops]) if ops else set()
            elif op.startswith(("v_readlane", "v_readfirstlane")):
                dst = regs(ops[0]); src = set().union(*[regs(o) for o in ops[1:]])
            else:
                dst = regs(ops[0]) if ops else set()
                src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co", "v_div_scale", "v_mad_u64", "v_mad_i64")) and len(ops) > 1:
                    dst |= regs(ops[1]); src = set().union(*[regs(o) for o in ops[2:]]) if len(ops) > 2 else set()
                if op.startswith(("v_mac_", "v_fmac_", "v_dot2c_", "v_dot4c_", "v_dot8c_", "v_pk_fmac")):
                    src |= dst
                if "vcc" in mods:
                    src.add(("vcc",))
        elif kind == "lds":
            if op.startswith(("ds_read", "ds_swizzle", "ds_bpermute", "ds_permute")) or "_rtn" in op:
                dst = regs(ops[0]); src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            else:
                src = set().union(*[regs(o) for o in ops]) if ops else set()
        elif kind == "vmem":
            if "_load" in op or ("atomic" in op and "glc" in (mods + " " + line)):
                if " lds" in (" " + mods) or line.rstrip().endswith(" lds"):
                    src = set().union(*[regs(o) for o in ops]) if ops else set()
                    src.add(("m0",))
                else:
                    dst = regs(ops[0]); src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
            else:
                src = set().union(*[regs(o) for o in ops]) if ops else set()
        elif kind == "salu":
            if op.startswith(("s_cmp", "s_bitcmp", "s_waitcnt", "s_barrier", "s_setprio", "s_sleep", "s_sendmsg")):
                src = set().union(*[regs(o) for o in ops]) if ops else set()
                if op.startswith("s_sendmsg"):
                    src.add(("m0",))
            else:
                dst = regs(ops[0]) if ops else set()
                src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()

        def viol(rule, h, need):
            bad.append((i, rule, f"{name}: {rule}: '{h['text']}' -> '{line}' needs {need} wait states, has {h['age']}"))

        for h in hist:
            a = h["age"]
            if h["kind"] == "valu":
                sg = {r for r in h["dst"] if r[0] in ("s", "vcc")}
                if kind == "vmem" and a < 5 and sg & {r for r in src if r[0] in ("s", "vcc")}:
                    viol("R1", h, 5)
                if op.startswith(("v_readlane", "v_writelane")) and a < 4 and len(ops) >= 3 and sg & regs(ops[2]):
                    viol("R2", h, 4)
                if is_dpp and a < 2 and len(ops) >= 2 and {r for r in h["dst"] if r[0] == "v"} & regs(ops[1]):
                    viol("R3", h, 2)
                if is_dpp and a < 5 and ("exec",) in h["dst"]:
                    viol("R4", h, 5)
                if h["trans"] and kind == "valu" and not op.startswith(TRANS) and a < 1 and {r for r in h["dst"] if r[0] == "v"} & src:
                    viol("R5", h, 1)
                if h["half"] and kind in ("valu", "mfma") and a < 1 and {r for r in h["dst"] if r[0] == "v"} & src:
                    viol("R6", h, 1)
                if op.startswith("v_div_fmas") and a < 4 and ("vcc",) in h["dst"]:
                    viol("R8", h, 4)
            elif h["kind"] == "mfma":
                need = MFMA_WAIT[h["passes"]]
                d = {r for r in h["dst"] if r[0] in ("v", "a")}
                if kind in ("valu", "lds", "vmem") and a < need and d & (src | (dst if kind == "valu" else set())):
                    viol("R7", h, need)
            elif h["kind"] == "salu":
                if ("m0",) in h["dst"] and a < 1 and ("m0",) in src:
                    viol("R9", h, 1)
        # a write of half a register: SDWA dst_sel, the dst bit of op_sel on 16-bit VOP3 (the mix instructions select SOURCE halves with
        # op_sel: their mixlo / mixhi forms are partial writes by definition), not the packed ops (they write both halves)
        partial_opsel = bool(re.search(r"op_sel:\[[01],[01],[01],1\]|op_sel:\[[01],[01],1\]|op_sel:\[[01],1\]", line)) and not op.startswith(("v_pk_", "v_fma_mix", "v_mad_mix"))
        # (the hazard is the SHIFT of a 16-bit result into bits 16..31 -- LLVM calls it "Shift16Def" --: mixlo and dst_sel:WORD_0 write in place)
        half = kind == "valu" and (op.startswith(("v_fma_mixhi", "v_mad_mixhi"))
                                   or ("dst_sel:" in line and "dst_sel:DWORD" not in line and "dst_sel:WORD_0" not in line and "dst_sel:BYTE_0" not in line)
                                   or partial_opsel)
        advance(1)
        hist.append({"age": 0, "kind": kind, "op": op, "dst": dst, "half": half, "trans": op.startswith(TRANS), "passes": mfma_passes(op) or 8,
                     "text": line})
    return bad


def functions(text):
    for m in re.finditer(r"^(\S+):\s*; @\1\n", text, re.M):
        end = text.find("\n.Lfunc_end", m.end())
        yield m.group(1), text[m.end(): end if end > 0 else len(text)]


def main():
    strict = "--strict" in sys.argv
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    total = 0
    for f in files:
        text = open(f).read()
        fns = list(functions(text))
        if not fns:                                         # an inline-asm text / header: every line (or "\n"-separated string) an instruction
            body = re.sub(r'\\n\\t|\\n', "\n", text).replace('"', "")
            fns = [(f, body)]
        for name, body in fns:
            bad = check_lines(body.splitlines(), strict, name[:60])
            for _, _, msg in bad[:40]:
                print(msg)
            total += len(bad)
    print(f"{total} hazard(s) in {len(files)} file(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
