#!/usr/bin/env python3
"""Writes d-vqvae_amd/csrc/vq_pipe_loop.h: the prologue and tile loop of vq_pipe.hip as ONE hand-written instruction block.

Why by hand: the loop keeps 104-112 of a wave's 128 registers busy for its whole life (64 codebook fragment registers, 16 accumulators,
tile fragments in flight, 16 registers of rows in flight from HBM).  Written in C++ (five attempts, round 6) the register allocator
either spilled codebook fragments, or moved registers whose loads were still in flight (an asm load's destination counts as written
at the end of the statement), or parked values in the accumulator registers the block had been given.  Here every register has one
owner:

  v0-v15    accumulators of the tile (16 MFMAs per tile, 32 rows x 32 entries)
  v16-v23   tile fragments in flight (FD = 2)
  v24-v31, v57-v59   temporaries of the vector work
  v32-v39   rows of the even tile in flight (two 16-byte pieces of each of the wave's two rows)       -- "set A"
  v40-v47   rows of the odd tile in flight                                                              -- "set B"
  v48-v55   lane constants (operands of the statement, pinned):
            v48 zbase   LDS address of the lane's fragment column in fp16 tile buffer 0:  L_Z16 + (lane % 32) * Z16_ROW + 16 * (lane / 32)
            v49 slotw   LDS address of the lane's merge slot, tile 0:   L_MS + (lane % 32) * MS_ROW + (2 wave + lane / 32) * 8
            v50 eesa    LDS address of the lane's accumulator start values:  L_EES + (32 wave + 4 (lane / 32)) * 4
            v51 loadoff byte offset of the lane's first 16 bytes in the wave's two rows:  (lane / 32) * 1024 + (lane % 32) * 16
            v52 conva   LDS address of the lane's 8 bytes in the fp16 image of its row, buffer 0:  L_Z16 + (2 wave + lane/32) * Z16_ROW + 8 (lane % 32)
            v53 rsa     LDS address of {eps sE, flag} of the lane's row, buffer 0:  L_RS + (2 wave + lane / 32) * 8
            v54 mrga    LDS address of the lane's slot to merge, tile 0:  L_MS + (2 wave + lane / 32) * MS_ROW + 8 (lane % 32)
            v55 entbase entry of accumulator register 0 of the lane's merge slot:  32 * (s / 2) + 4 * (s % 2), s = lane % 32
  v56       fragment base address of the current tile buffer
  v60-v63   NOT used (the compiler needs a few registers of its own around the statement)
  v64-v127  the wave's 16 codebook fragments (A operands)
  s64-s99   the block's scalars (the operands, read in place, live below)

Per period t (between two workgroup barriers) every wave runs the same program (round 6, second structure; the first one ran the
pieces one after the other in an order rotated over the four waves of a SIMD: the per-wave stamps showed vector pieces of 500-1 300
cycles each -- chains of dependent instructions, LDS round trips, scalar decisions -- and the matrix pipe idle 60 % of the period):
  P  the 16 MFMAs of tile t (buffer t % 3), and IN THEIR GAPS, instruction by instruction,
  C  the conversion of the wave's two rows of tile t + 2 from the register set t % 2 and the loads of tile t + 4 into that set,
  M  the merge of the wave's two rows of tile t - 1 (decision on the scalar unit), both written without branches;
  S  (min, second) of the lane's 16 scores -> slot of tile t;   R  the (rare) record of a row the merge could not decide.
Every LDS wait is COUNTED: the scheduler below merges the streams and derives each s_waitcnt lgkmcnt(N) from the program order of the
always-executed LDS operations behind the one that is needed (operations of a wave complete in order; a conditional one in between
only makes a wait longer).

usage: gen_vq_pipe.py [FD]   (FD = tile fragments in flight, 2 (default) .. 2)"""
import os
import sys

FD = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2

# ---- LDS layout: must match vq_pipe.hip (static_assert'ed there through the VQP_* macros this file emits)
K, TILE, MAX_TILES, NWV = 512, 32, 8, 16
Z16_ROW = 528
Z16_BUF = TILE * Z16_ROW
NZB = 3
MS_ROW = 272
MS_BUF = TILE * MS_ROW
PAIR_CAP = 2048
L_Z16 = 0
L_MS = L_Z16 + NZB * Z16_BUF
L_RS = L_MS + MAX_TILES * MS_BUF
L_EES = L_RS + 4 * TILE * 8
L_RES = L_EES + K * 4
L_REC = L_RES + MAX_TILES * TILE * 8
L_PAIR = L_REC + MAX_TILES * TILE * 8
L_SLOW = L_PAIR + PAIR_CAP * 4
L_UND = L_SLOW + MAX_TILES * TILE * 2
UND_PER_WAVE = 16
L_CNT = L_UND + NWV * UND_PER_WAVE * 2
L_DBG = L_CNT + 128
LDS_BYTES = L_DBG + 128 + NWV * 16 * 4            # [16] u64 prologue stamps, then [16 waves][2 tiles][8] u32 period stamps

# ---- registers
assert FD == 2, "the temporaries of the interleaved vector work live in v24-v31"
FRAG = [f"v[{16 + 4 * i}:{16 + 4 * i + 3}]" for i in range(FD)]
XSET = {0: list(range(32, 40)), 1: list(range(40, 48))}
V_ZBASE, V_SLOTW, V_EESA, V_LOADOFF, V_CONVA, V_RSA, V_MRGA, V_ENTB = 48, 49, 50, 51, 52, 53, 54, 55
V_ZA = 56
W = [24, 25, 26, 27, 28, 29, 30, 31]           # temporaries of the vector work inside the matrix phase
V_THR, V_VAL, V_ADR = 58, 59, 57               # threshold of the merged rows (kept for the record; (v58, v59) is stored as a pair), scratch, scratch
TS = [16, 17, 18]                              # temporaries of the scores (fragment registers: dead behind the matrix phase)

# the block's own scalars: s64-s99 (the operands and whatever the compiler keeps across the block live below)
S_T, S_ZRD, S_ZWR, S_MS, S_RSWR, S_RSRD, S_LDROW, S_MROW, S_UND = 84, 85, 86, 87, 88, 89, 90, 91, 92
S_MASK, S_BIG, S_MM1 = 93, 94, 95
S_BADE = 96                                    # s[96:97] all ones when the codebook image is invalid
S_FIRST, S_LAST = 64, 99
# s64-s79 temporaries; s80 / s81 undecided flags of the two rows, s82 / s83 their slow flags (live to the end of the period)
# operands read in place (never written)
S_M, S_NTL, S_ROWSTEP, S_EMAX, S_DEMAX, S_SEF, S_LDSB, S_WAVE = ("%[M]", "%[ntl]", "%[rowstep]", "%[emax]", "%[demax]", "%[sef]",
                                                                 "%[ldsb]", "%[wave]")

uid = [0]


def label(name):
    uid[0] += 1
    return f".Lvqp_{name}_{uid[0]}_%="


# ------------------------------------------------------------------------------------------------ instruction records
def I(text, lds=None, need=(), glue=False):
    """one instruction: lds = tag of the always-executed LDS operation it issues (None: not an LDS operation; "" an untagged one),
    need = tags of LDS operations whose results it uses, glue = stays with the next instruction (exec-masked groups, SCC pairs)"""
    return {"t": text, "lds": lds, "need": tuple(need), "glue": glue}


def dpp_reduce(op, dst, src):
    """all-reduce over the 16 lanes of a DPP row: four steps, two wait states in front of each DPP read of a fresh register"""
    o = [I("s_nop 1", glue=True), I(f"{op} v{dst}, v{src}, v{src} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")]
    for ctl in ("quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
        o += [I("s_nop 1", glue=True), I(f"{op} v{dst}, v{dst}, v{dst} {ctl} row_mask:0xf bank_mask:0xf bound_ctrl:1")]
    return o


def products(mfma=True, reads=True):
    """matrix phase: head, then 16 steps (the instructions in front of each gap)"""
    head = [I(f"v_add_u32_e32 v{V_ZA}, s{S_ZRD}, v{V_ZBASE}"),
            I(f"ds_read_b128 v[0:3], v{V_EESA}", lds="ci0"), I(f"ds_read_b128 v[4:7], v{V_EESA} offset:32", lds="ci1"),
            I(f"ds_read_b128 v[8:11], v{V_EESA} offset:64", lds="ci2"), I(f"ds_read_b128 v[12:15], v{V_EESA} offset:96", lds="ci3")]
    if not reads:
        return head, [[I("s_nop 0", need=("ci3",))]] + [[] for _ in range(15)]
    for i in range(FD):
        head.append(I(f"ds_read_b128 {FRAG[i]}, v{V_ZA} offset:{32 * i}", lds=f"f{i}"))
    steps = []
    for s in range(16):
        st = []
        need = (f"f{s}",) + (("ci3",) if s == 0 else ())
        if mfma:
            st.append(I(f"v_mfma_f32_32x32x16_f16 v[0:15], v[{64 + 4 * s}:{64 + 4 * s + 3}], {FRAG[s % FD]}, v[0:15]", need=need))
        else:
            st.append(I("s_nop 0", need=need))
        if s + FD < 16:
            st.append(I(f"ds_read_b128 {FRAG[s % FD]}, v{V_ZA} offset:{32 * (s + FD)}", lds=f"f{s + FD}"))
        steps.append(st)
    return head, steps


def scores(on=True):
    """(min, second) of the lane's 16 scores, id in the low mantissa bits -> slot of tile t"""
    m1, m2, a = TS
    o = [I("s_nop 15"),                         # MFMA result -> vector ALU read (8-pass XDL: 12 states; 16 here)
         I(f"v_and_or_b32 v{m1}, v0, s{S_MASK}, 0"), I(f"v_mov_b32_e32 v{m2}, 0x7f800000")]
    for e in range(1, 16) if on else []:
        o += [I(f"v_and_or_b32 v{e}, v{e}, s{S_MASK}, {e}"), I(f"v_med3_f32 v{m2}, v{m1}, v{m2}, v{e}"), I(f"v_min_f32_e32 v{m1}, v{m1}, v{e}")]
    o += [I(f"v_add_u32_e32 v{a}, s{S_MS}, v{V_SLOTW}"), I(f"ds_write_b64 v{a}, v[{m1}:{m2}]", lds="")]
    return o


def convert(st, on=True, wait=True):
    """C without branches: tile t + 2 from register set st -> fp16 image + {eps sE, flag}.  Behind the workgroup's last tile it
    converts the (valid, never used) rows the always-issued loads brought: the buffers it writes are free."""
    if not on:
        return []
    x = XSET[st]
    lo0, hi0, lo1, hi1, hh, u, v, w = W
    o = []
    if wait:
        o.append(I("s_waitcnt vmcnt(2)"))       # the older set has arrived: at most the other set's two loads are outstanding
    o += [I(f"v_cvt_pk_f16_f32 v{lo0}, v{x[0]}, v{x[1]}"), I(f"v_cvt_pk_f16_f32 v{hi0}, v{x[2]}, v{x[3]}"),
          I(f"v_cvt_pk_f16_f32 v{lo1}, v{x[4]}, v{x[5]}"), I(f"v_cvt_pk_f16_f32 v{hi1}, v{x[6]}, v{x[7]}"),
          I(f"v_mov_b32_e32 v{hh}, 0"),
          I(f"v_dot2c_f32_f16_e32 v{hh}, v{lo0}, v{lo0}"), I(f"v_dot2c_f32_f16_e32 v{hh}, v{hi0}, v{hi0}"),
          I(f"v_dot2c_f32_f16_e32 v{hh}, v{lo1}, v{lo1}"), I(f"v_dot2c_f32_f16_e32 v{hh}, v{hi1}, v{hi1}"),
          I(f"v_add_u32_e32 v{u}, s{S_ZWR}, v{V_CONVA}"),
          I(f"ds_write_b64 v{u}, v[{lo0}:{hi0}]", lds=""), I(f"ds_write_b64 v{u}, v[{lo1}:{hi1}] offset:256", lds="")]
    o += dpp_reduce("v_add_f32_dpp", hh, hh)
    o += [I(f"ds_swizzle_b32 v{u}, v{hh} offset:swizzle(SWAP,16)", lds="csw"),
          I(f"v_add_f32_e32 v{hh}, v{hh}, v{u}", need=("csw",)),                 # |h(z)|^2 of the row
          I(f"v_sqrt_f32_e32 v{u}, v{hh}", glue=True), I("s_nop 0"),              # hn
          # a-priori rounding bound: the same operations, in the same order, as the C++ expression the compiler built for
          # vq_stream16.hip with DVQ_MEASURE_DZ=0 (see the header of vq_pipe.hip)
          I(f"v_mul_f32_e32 v{v}, 0x3a0020cd, v{u}"), I(f"v_add_f32_e32 v{v}, 0x3500d959, v{v}"),     # dzn = hn * 4.8877e-4 + 4.8e-7
          I(f"v_add_f32_e32 v{w}, v{u}, v{v}"), I(f"v_mul_f32_e32 v{w}, 0x3f800347, v{w}"),           # zn = (hn + dzn) * 1.0001
          I(f"v_add_f32_e32 v{lo0}, {S_EMAX}, v{w}"),                                                  # u = zn + emax
          I(f"v_mul_f32_e32 v{hi0}, {S_EMAX}, v{v}"), I(f"v_mul_f32_e32 v{lo1}, {S_DEMAX}, v{w}"),    # dzn emax, zn demax
          I(f"v_add_f32_e32 v{hi0}, v{hi0}, v{lo1}"), I(f"v_mul_f32_e32 v{lo1}, {S_DEMAX}, v{v}"),    # + dzn demax
          I(f"v_add_f32_e32 v{hi0}, v{lo1}, v{hi0}"),
          I(f"v_mul_f32_e32 v{lo1}, 0x38d3cff6, v{lo0}"), I(f"v_mul_f32_e32 v{hi0}, 0x408020c5, v{hi0}"),   # 1.01e-4 u, 4.004 (...)
          I(f"v_mul_f32_e32 v{lo1}, v{lo0}, v{lo1}"), I(f"v_add_f32_e32 v{hi0}, v{lo1}, v{hi0}"),           # eps
          I(f"v_mul_f32_e32 v{lo1}, {S_SEF}, v{hi0}"),                                                   # eps sE  -> lo1
          I(f"v_cmp_nge_f32_e32 vcc, s{S_BIG}, v{hh}"),                   # !(hh <= 3e38): NaN / Inf / fp16 overflow
          I(f"v_cmp_nge_f32_e64 s[64:65], s{S_BIG}, v{lo1}", glue=True),   # !(eps sE <= 3e38)
          I("s_nop 3", glue=True),
          I("s_or_b64 s[64:65], s[64:65], vcc", glue=True), I(f"s_or_b64 s[64:65], s[64:65], s[{S_BADE}:{S_BADE + 1}]", glue=True),
          I("s_nop 1", glue=True),
          I(f"v_cndmask_b32_e64 v{hi1}, 0, 1, s[64:65]"),                  # flag -> hi1 (= lo1 + 1: the pair {eps sE, flag})
          I(f"v_add_u32_e32 v{u}, s{S_RSWR}, v{V_RSA}"),
          I(f"ds_write_b64 v{u}, v[{lo1}:{hi1}]", lds="")]
    return o


def loads(st, on=True, nt=True, hot=False):
    """the rows of tile t + 4 into register set st.  Always issued (the counted vmcnt wait needs the same number of loads in flight
    in every period): tiles behind the workgroup's last one and rows behind the end of the data read the last row again."""
    x = XSET[st]
    o = []
    if on:
        ntm = " nt" if nt else ""
        u = W[5]
        o += [I("s_mov_b32 s64, %[row0]") if hot else I(f"s_mov_b32 s64, s{S_LDROW}"),
              I(f"s_min_i32 s64, s64, s{S_MM1}"), I("s_max_i32 s64, s64, 0"),
              I(f"s_add_i32 s65, s{S_LDROW}, 1"), I(f"s_cmp_lt_i32 s65, {S_M}", glue=True), I("s_cselect_b32 s65, -1, 0x3ff"),
              I(f"v_and_b32_e32 v{u}, s65, v{V_LOADOFF}"),
              I("s_mov_b32 s65, 0"), I("s_lshl_b64 s[64:65], s[64:65], 10"),
              I("s_add_u32 s64, s64, %[zplo]", glue=True), I("s_addc_u32 s65, s65, %[zphi]"),
              I(f"global_load_dwordx4 v[{x[0]}:{x[3]}], v{u}, s[64:65]{ntm}"),
              I(f"global_load_dwordx4 v[{x[4]}:{x[7]}], v{u}, s[64:65] offset:512{ntm}")]
    o.append(I(f"s_add_i32 s{S_LDROW}, s{S_LDROW}, {S_ROWSTEP}"))
    return o


def half_decision(b1, b2, bad, rowadd, wm, und, slowf):
    """scalar decision for one of the wave's two rows: winner mask (b1 if the row is live and decided, else 0), undecided flag,
    slow flag (bad row or no score within eps)"""
    t = ["s_bcnt1_i32_b32 s74, s%d" % b1, "s_bcnt1_i32_b32 s75, s%d" % b2,
         "s_cmp_eq_u32 s74, 1", "s_cselect_b32 s76, 1, 0",
         "s_cmp_eq_u32 s75, 0", "s_cselect_b32 s77, 1, 0", "s_and_b32 s76, s76, s77",
         "s_cmp_eq_u32 s%d, 0" % bad, "s_cselect_b32 s77, 1, 0", "s_and_b32 s76, s76, s77",                    # unique
         f"s_add_i32 s78, s{S_MROW}, {rowadd}", f"s_cmp_lt_i32 s78, {S_M}", "s_cselect_b32 s77, 1, 0",           # live
         f"s_cmp_ge_i32 s{S_T}, 1", "s_cselect_b32 s78, 1, 0", "s_and_b32 s77, s77, s78",                        # (tile -1 does not exist)
         "s_and_b32 s78, s76, s77", "s_cmp_lg_u32 s78, 0", "s_cselect_b32 s%d, s%d, 0" % (wm, b1),
         "s_andn2_b32 s%d, s77, s76" % und,
         "s_cmp_eq_u32 s74, 0", "s_cselect_b32 s78, 1, 0",
         "s_cmp_lg_u32 s%d, 0" % bad, "s_cselect_b32 s79, 1, 0", "s_or_b32 s%d, s78, s79" % slowf]
    # a compare and the select that reads its SCC stay together (the matrix phase's own instructions do not touch SCC, but the
    # scheduler may cut the stream anywhere else)
    return [I(x, glue=x.startswith("s_cmp")) for x in t]


def merge(on=True):
    """M without branches: merge of the wave's two rows of tile t - 1 (32 lanes per row, one slot per lane).  In period 0 it
    looks at whatever lies in front of the slot array: the rows are not live (t < 1), nothing is stored.
    Leaves: s80 / s81 undecided flags, s82 / s83 slow flags, v58 the threshold (for the record behind the matrix phase)."""
    if not on:
        return [I("s_mov_b32 s80, 0"), I("s_mov_b32 s81, 0")]
    m1, m2, eps, flg, u, a, ent, zero = W
    thr, val = V_THR, V_VAL
    o = [I(f"s_sub_u32 s64, s{S_MS}, {MS_BUF}"),
         I(f"v_add_u32_e32 v{a}, s64, v{V_MRGA}"), I(f"ds_read_b64 v[{m1}:{m2}], v{a}", lds="ms"),
         I(f"v_add_u32_e32 v{a}, s{S_RSRD}, v{V_RSA}"), I(f"ds_read_b64 v[{eps}:{flg}], v{a}", lds="mr")]
    d = dpp_reduce("v_min_f32_dpp", thr, m1)
    d[0]["need"] = ("ms",)                          # (the wait goes in front of the glued s_nop)
    o += d
    o += [I(f"ds_swizzle_b32 v{u}, v{thr} offset:swizzle(SWAP,16)", lds="msw"),
          I(f"v_min_f32_e32 v{thr}, v{thr}, v{u}", need=("msw", "mr")),
          I(f"v_add_f32_e32 v{thr}, v{eps}, v{thr}"),                        # row minimum + eps sE
          I(f"v_cmp_le_f32_e64 s[64:65], v{m1}, v{thr}"), I(f"v_cmp_le_f32_e64 s[66:67], v{m2}, v{thr}"),
          I(f"v_cmp_ne_u32_e64 s[68:69], 0, v{flg}", glue=True),
          I("s_nop 3")]
    o += half_decision(64, 66, 68, 0, 70, 80, 82)
    o += half_decision(65, 67, 69, 1, 71, 81, 83)
    # winner lanes of decided rows write the entry: key = (0 : entry); exec-masked, no branch (exec = 0: nothing happens)
    o += [I(f"s_sub_u32 s72, s{S_T}, 1"), I("s_lshl_b32 s72, s72, 8"), I(f"s_add_u32 s72, s72, {L_RES - L_RS}"),   # (t - 1) * 256 + L_RES - L_RS
          I("s_mov_b64 exec, s[70:71]", glue=True),
          I(f"v_and_b32_e32 v{u}, 15, v{m1}", glue=True), I(f"v_lshlrev_b32_e32 v{val}, 1, v{u}", glue=True),
          I(f"v_and_b32_e32 v{val}, 24, v{val}", glue=True), I(f"v_and_b32_e32 v{u}, 3, v{u}", glue=True),
          I(f"v_or3_b32 v{ent}, v{u}, v{val}, v{V_ENTB}", glue=True), I(f"v_mov_b32_e32 v{zero}, 0", glue=True),
          I(f"v_add_u32_e32 v{a}, s72, v{V_RSA}", glue=True), I(f"ds_write_b64 v{a}, v[{ent}:{zero}]", glue=True),   # (not counted: exec may be 0)
          I("s_mov_b64 exec, -1")]
    return o


def record():
    """R (behind the scores; rare, with branches): a row the merge could not decide: {threshold, slow flag} -> L_REC[rowslot],
    rowslot -> the wave's list"""
    o = []
    thr, slw, a, val = V_THR, V_VAL, V_ADR, W[0]
    for half, (und, slowf, lane_exec) in enumerate(((80, 82, ("1", "0")), (81, 83, ("0", "1")))):
        nound = label("nound")
        o += [f"s_cmp_eq_u32 s{und}, 0", f"s_cbranch_scc1 {nound}",
              f"s_mov_b32 exec_lo, {lane_exec[0]}", f"s_mov_b32 exec_hi, {lane_exec[1]}",
              f"v_mov_b32_e32 v{slw}, s{slowf}",                                             # (v58, v59) = (threshold, slow flag)
              f"s_sub_u32 s72, s{S_T}, 1", "s_lshl_b32 s72, s72, 8", f"s_add_u32 s74, s72, {L_REC - L_RS}",
              f"v_add_u32_e32 v{a}, s74, v{V_RSA}",
              f"ds_write_b64 v{a}, v[{thr}:{slw}]",
              # rowslot = (t - 1) * 32 + 2 wave + half; list address = L_UND + (wave * UND_PER_WAVE + und) * 2
              f"s_sub_u32 s74, s{S_T}, 1", "s_lshl_b32 s74, s74, 5", f"s_lshl_b32 s75, {S_WAVE}, 1", "s_add_u32 s74, s74, s75",
              f"s_add_u32 s74, s74, {half}",
              f"s_mul_i32 s75, {S_WAVE}, {UND_PER_WAVE}", f"s_add_u32 s75, s75, s{S_UND}", "s_lshl_b32 s75, s75, 1",
              f"s_add_u32 s75, s75, {S_LDSB}", f"s_add_u32 s75, s75, {L_UND}",
              f"v_mov_b32_e32 v{a}, s75", f"v_mov_b32_e32 v{val}, s74", f"ds_write_b16 v{a}, v{val}",
              f"s_add_u32 s{S_UND}, s{S_UND}, 1",
              "s_mov_b64 exec, -1",
              f"{nound}:"]
    return o


def schedule(head, steps, filler):
    """head, then per step its instructions followed by a share of the filler; every `need` becomes a counted s_waitcnt lgkmcnt"""
    order = list(head)
    n = max(len(steps), 1)
    fi = 0
    for s, st in enumerate(steps):
        order += st
        stop = len(filler) * (s + 1) // n
        while fi < len(filler) and (fi < stop or filler[fi - 1]["glue"]):
            order.append(filler[fi])
            fi += 1
    order += filler[fi:]
    out, issued, done = [], [], -1                 # issued: tags of the LDS operations so far; done: index of the newest known complete
    for ins in order:
        if ins["need"]:
            idx = max(max(i for i, t in enumerate(issued) if t == tag) for tag in ins["need"])
            if idx > done:
                nn = min(len(issued) - 1 - idx, 15)
                out.append(f"s_waitcnt lgkmcnt({nn})")
                done = len(issued) - 1 - nn
        out.append(ins["t"])
        if ins["lds"] is not None:
            issued.append(ins["lds"])
    return out


STAMPS = [False]


def stamp(k):
    """diagnostics: shader-clock stamp k (0..7) of periods 3 and 4 -> L_DBG + 128 + wave * 64 + (t - 3) * 32 + 4 k"""
    if not STAMPS[0]:
        return []
    skip = label("nostamp")
    return [f"s_sub_u32 s64, s{S_T}, 3", "s_cmp_gt_u32 s64, 1", f"s_cbranch_scc1 {skip}",
            "s_memtime s[98:99]",
            "s_lshl_b32 s64, s64, 5", f"s_lshl_b32 s65, {S_WAVE}, 6", "s_add_u32 s64, s64, s65", f"s_add_u32 s64, s64, {S_LDSB}",
            f"s_add_u32 s64, s64, {L_DBG + 128 + 4 * k}",
            "s_waitcnt lgkmcnt(0)", f"v_mov_b32_e32 v{W[0]}, s98", f"v_mov_b32_e32 v{W[1]}, s64", f"ds_write_b32 v{W[1]}, v{W[0]}",
            f"{skip}:"]


def advance():
    return [f"s_add_i32 s{S_T}, s{S_T}, 1",
            f"s_add_u32 s{S_ZRD}, s{S_ZRD}, {Z16_BUF}", f"s_cmp_eq_u32 s{S_ZRD}, {NZB * Z16_BUF}", f"s_cselect_b32 s{S_ZRD}, 0, s{S_ZRD}",
            f"s_add_u32 s{S_ZWR}, s{S_ZWR}, {Z16_BUF}", f"s_cmp_eq_u32 s{S_ZWR}, {NZB * Z16_BUF}", f"s_cselect_b32 s{S_ZWR}, 0, s{S_ZWR}",
            f"s_add_i32 s{S_MS}, s{S_MS}, {MS_BUF}",
            f"s_add_u32 s{S_RSWR}, s{S_RSWR}, 256", f"s_and_b32 s{S_RSWR}, s{S_RSWR}, 0x3ff",
            f"s_add_u32 s{S_RSRD}, s{S_RSRD}, 256", f"s_and_b32 s{S_RSRD}, s{S_RSRD}, 0x3ff",
            f"s_add_i32 s{S_MROW}, s{S_MROW}, {S_ROWSTEP}"]


def period(st, abl, interleave=True):
    head, steps = products(mfma=not (abl & 32), reads=not (abl & 64))
    filler = (convert(st, on=not (abl & 4), wait=not (abl & 1) and not (abl & 128))
              + loads(st, on=not (abl & 1), nt=not (abl & 256), hot=bool(abl & 512))
              + merge(on=not (abl & 2)))
    o = ["s_waitcnt lgkmcnt(0)"]
    if not (abl & 16):
        o.append("s_barrier")
    o += stamp(0)
    if interleave:
        o += schedule(head, steps, filler)
    else:                                          # diagnostics: the vector work in front of the matrix phase instead of inside it
        o += schedule(filler, [], []) + stamp(1) + schedule(head, steps, [])
    o += stamp(2)
    o += [i["t"] for i in scores(on=not (abl & 8))]
    o += stamp(3)
    if not (abl & 2):
        o += record()
    o += stamp(4)
    return o


def program(abl=0, stamps=False, interleave=True):
    uid[0] = 0
    STAMPS[0] = stamps
    lane = W[0]
    o = [  # ---- constants of the block
        "s_cmp_eq_u32 %[evalid], 0", f"s_cselect_b64 s[{S_BADE}:{S_BADE + 1}], -1, 0",
        f"s_mov_b32 s{S_MASK}, 0xffffffe0", f"s_mov_b32 s{S_BIG}, 0x7f61b1e6", f"s_sub_u32 s{S_MM1}, {S_M}, 1",
        f"s_mov_b32 s{S_UND}, 0", "s_mov_b32 s80, 0", "s_mov_b32 s81, 0",
        # ---- rows of tiles 0 and 1 first (HBM latency), then the wave's 16 codebook fragments (L2)
        f"s_mov_b32 s{S_LDROW}, %[row0]"]
    if not (abl & 1):
        o += [i["t"] for i in loads(0) + loads(1)]
    else:
        o += [f"s_add_i32 s{S_LDROW}, s{S_LDROW}, {S_ROWSTEP}"] * 2
        o += [f"v_mov_b32_e32 v{r}, 0" for r in range(32, 48)]
    o += [f"v_mbcnt_lo_u32_b32 v{lane}, -1, 0", f"v_mbcnt_hi_u32_b32 v{lane}, -1, v{lane}", f"v_lshlrev_b32_e32 v{lane}, 4, v{lane}"]
    # fragment s is 1 KB: the 13-bit signed offset field reaches four of them per base register
    o += ["s_mov_b64 s[64:65], %[img]"]
    for blk in range(4):
        if blk:
            o += ["s_add_u32 s64, s64, 4096", "s_addc_u32 s65, s65, 0"]
        for s4 in range(4):
            s = 4 * blk + s4
            o.append(f"global_load_dwordx4 v[{64 + 4 * s}:{64 + 4 * s + 3}], v{lane}, s[64:65] offset:{1024 * s4}")
    o += ["s_waitcnt vmcnt(0)",
          # ---- conversions of tiles 0 and 1, loads of tiles 2 and 3: the C part at t = -2 and t = -1
          f"s_mov_b32 s{S_T}, -2",
          f"s_mov_b32 s{S_ZRD}, {Z16_BUF}", f"s_mov_b32 s{S_ZWR}, 0", f"s_mov_b32 s{S_MS}, {-2 * MS_BUF}",
          f"s_mov_b32 s{S_RSWR}, 0", f"s_mov_b32 s{S_RSRD}, 256",
          f"s_mul_i32 s64, {S_ROWSTEP}, 3", f"s_sub_u32 s{S_MROW}, %[row0], s64"]
    for st in (0, 1):
        o += schedule(convert(st, on=not (abl & 4), wait=False) + loads(st, on=not (abl & 1)), [], []) + advance()
    # ---- (prologue done: a 100 MHz stamp per wave into L_DBG, read by the diagnostics build only)
    o += ["s_memrealtime s[64:65]", f"s_lshl_b32 s66, {S_WAVE}, 3", f"s_add_u32 s66, s66, {S_LDSB}", f"s_add_u32 s66, s66, {L_DBG}",
          "s_mov_b32 exec_lo, 1", "s_mov_b32 exec_hi, 0", "s_waitcnt lgkmcnt(0)",
          f"v_mov_b32_e32 v{W[0]}, s64", f"v_mov_b32_e32 v{W[1]}, s65", f"v_mov_b32_e32 v{W[2]}, s66",
          f"ds_write_b64 v{W[2]}, v[{W[0]}:{W[1]}]", "s_mov_b64 exec, -1"]
    # ---- tile loop, two periods per trip (register sets A, B)
    loop, done = label("loop"), label("done")
    o += [f"{loop}:"]
    o += period(0, abl, interleave) + advance() + [f"s_cmp_ge_i32 s{S_T}, {S_NTL}", f"s_cbranch_scc1 {done}"]
    o += period(1, abl, interleave) + advance() + [f"s_cmp_lt_i32 s{S_T}, {S_NTL}", f"s_cbranch_scc1 {loop}"]
    o += [f"{done}:",
          "s_waitcnt vmcnt(0)",                                           # rows behind the last tile (never used) have landed
          # undecided-row count of this wave -> L_CNT[4 + wave]
          f"s_lshl_b32 s64, {S_WAVE}, 2", f"s_add_u32 s64, s64, {S_LDSB}", f"s_add_u32 s64, s64, {L_CNT + 16}",
          "s_mov_b32 exec_lo, 1", "s_mov_b32 exec_hi, 0",
          f"v_mov_b32_e32 v{W[0]}, s64", f"v_mov_b32_e32 v{W[1]}, s{S_UND}", f"ds_write_b32 v{W[0]}, v{W[1]}",
          "s_mov_b64 exec, -1",
          "s_waitcnt lgkmcnt(0)"]
    return o


def lit(lines):
    return " \\\n    ".join('"' + l + '\\n\\t"' for l in lines)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd", "csrc", "vq_pipe_loop.h")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_vq_pipe.py %d -- prologue and tile loop of vq_pipe.hip as one instruction block; do not edit\n"
                "#pragma once\n" % FD)
        for k, v in (("Z16_ROW", Z16_ROW), ("Z16_BUF", Z16_BUF), ("NZB", NZB), ("MS_ROW", MS_ROW), ("MS_BUF", MS_BUF),
                     ("PAIR_CAP", PAIR_CAP), ("L_Z16", L_Z16), ("L_MS", L_MS), ("L_RS", L_RS), ("L_EES", L_EES), ("L_RES", L_RES),
                     ("L_REC", L_REC), ("L_PAIR", L_PAIR), ("L_SLOW", L_SLOW), ("L_UND", L_UND), ("UND_PER_WAVE", UND_PER_WAVE),
                     ("L_CNT", L_CNT), ("L_DBG", L_DBG), ("LDS_BYTES", LDS_BYTES), ("FD", FD)):
            f.write(f"#define VQP_{k} {v}\n")
        f.write("#define VQP_ASM_DEFAULT \\\n    " + lit(program()) + "\n")
        f.write("#ifdef DVQ_DIAG\n")
        f.write("#define VQP_ASM_NOROT \\\n    " + lit(program(interleave=False)) + "\n")
        f.write("#define VQP_ASM_STAMPS \\\n    " + lit(program(stamps=True)) + "\n")
        for abl in (1, 2, 4, 8, 32, 96, 15, 130, 256, 512, 3):
            f.write(f"#define VQP_ASM_ABL{abl} \\\n    " + lit(program(abl=abl)) + "\n")
        f.write("#endif\n")
        clob = ([f'"v{i}"' for i in list(range(0, 48)) + [V_ZA, V_THR, V_VAL, V_ADR] + list(range(64, 128))]
                + [f'"s{i}"' for i in range(S_FIRST, S_LAST + 1)])
        f.write("#define VQP_ASM_CLOBBERS " + ", ".join(clob) + ', "vcc", "scc", "memory"\n')
    n = len(program())
    print("wrote", os.path.normpath(out), "FD =", FD, "instructions+labels in the default block:", n)
    if "--list" in sys.argv:
        print("\n".join(program()))


if __name__ == "__main__":
    main()
