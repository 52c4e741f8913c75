#!/usr/bin/env python3
"""Writes d-vqvae_amd/csrc/vq_pipe_loop.h: the prologue and tile loop of vq_pipe.hip as ONE hand-written instruction block.

Why by hand: the loop keeps 112 of a wave's 128 registers busy for its whole life (64 codebook fragment registers, 16 accumulators,
16 registers of tile fragments in flight, 16 registers of rows in flight from HBM).  Written in C++ (five attempts, round 6) the
register allocator either spilled codebook fragments, or moved registers whose loads were still in flight (an asm load's destination
counts as written at the end of the statement), or parked values in the accumulator registers the block had been given.  Here every
register has one owner:

  v0-v15    accumulators of the tile (16 MFMAs per tile, 32 rows x 32 entries)
  v16-v31   tile fragments in flight (FD x 4); outside the matrix phase v16-v23 are the vector phases' temporaries
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
  v57-v63   NOT used (the compiler needs a few registers of its own around the statement)
  v64-v127  the wave's 16 codebook fragments (A operands)
  s64-s99   the block's scalars (the operands, read in place, live below)

Per period t (between two workgroup barriers) a wave does, in an order that depends on its position q among the four waves of its
SIMD:  P  products of tile t (buffer t % 3) into the accumulators;  S  (min, second) of the lane's 16 scores -> slot of tile t;
C  conversion of its two rows of tile t + 2 from the register set t % 2, then the loads of tile t + 4 into that set;
M  merge of its two rows of tile t - 1.   q = 0: P S C M,  q = 1: C P S M,  q = 2: M C P S,  q = 3: C M P S.

usage: gen_vq_pipe.py [FD]   (FD = tile fragments in flight, 2..4; default 4)"""
import os
import sys

FD = int(sys.argv[1]) if len(sys.argv) > 1 else 4

# ---- LDS layout: must match vq_pipe.hip (static_assert'ed there through the VQP_L_* macros this file emits)
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
ACC = 0
FRAG = [f"v[{16 + 4 * i}:{16 + 4 * i + 3}]" for i in range(FD)]
XSET = {0: list(range(32, 40)), 1: list(range(40, 48))}
V_ZBASE, V_SLOTW, V_EESA, V_LOADOFF, V_CONVA, V_RSA, V_MRGA, V_ENTB = 48, 49, 50, 51, 52, 53, 54, 55
T = [16, 17, 18, 19, 20, 21, 22, 23]          # temporaries of the vector phases (the tile fragments' registers: dead outside the matrix phase)
V_ZA = 56                                      # fragment base address of the current tile buffer (live through the matrix phase)
AF0 = 64

# the block's own scalars: s64-s99 (36 registers; the operands and whatever the compiler keeps across the block live below)
S_T, S_ZRD, S_ZWR, S_MS, S_RSWR, S_RSRD, S_LDROW, S_MROW, S_UND = 84, 85, 86, 87, 88, 89, 90, 91, 92
S_MASK, S_BIG, S_MM1 = 93, 94, 95
S_BADE = 96                                    # s[96:97] all ones when the codebook image is invalid
S_FIRST, S_LAST = 64, 99
# s64-s83: temporaries of the phases (named s52.. in the text below, shifted by +12)
# operands read in place (never written)
S_M, S_NTL, S_Q, S_ROWSTEP, S_EMAX, S_DEMAX, S_SEF, S_LDSB, S_WAVE = ("%[M]", "%[ntl]", "%[q]", "%[rowstep]", "%[emax]", "%[demax]",
                                                                      "%[sef]", "%[ldsb]", "%[wave]")

uid = [0]


def label(name):
    uid[0] += 1
    return f".Lvqp_{name}_{uid[0]}_%="


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
            "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v26, s98", "v_mov_b32_e32 v27, s64", "ds_write_b32 v27, v26",
            f"{skip}:"]


def chain(mfma=True, reads=True):
    """matrix phase"""
    o = [f"v_add_u32_e32 v{V_ZA}, s{S_ZRD}, v{V_ZBASE}",
         f"ds_read_b128 v[0:3], v{V_EESA}", f"ds_read_b128 v[4:7], v{V_EESA} offset:32",
         f"ds_read_b128 v[8:11], v{V_EESA} offset:64", f"ds_read_b128 v[12:15], v{V_EESA} offset:96"]
    if not reads:
        return o + ["s_waitcnt lgkmcnt(0)"]
    for i in range(FD):
        o.append(f"ds_read_b128 {FRAG[i]}, v{V_ZA} offset:{32 * i}")
    for s in range(16):
        o.append(f"s_waitcnt lgkmcnt({min(15, s + FD - 1) - s})")
        if mfma:
            o.append(f"v_mfma_f32_32x32x16_f16 v[0:15], v[{AF0 + 4 * s}:{AF0 + 4 * s + 3}], {FRAG[s % FD]}, v[0:15]")
        if s + FD < 16:
            o.append(f"ds_read_b128 {FRAG[s % FD]}, v{V_ZA} offset:{32 * (s + FD)}")
    if mfma:
        o.append("s_nop 15")                   # MFMA result -> vector ALU read (8-pass XDL: 12 states; 16 here)
    return o


def scores(on=True):
    """(min, second) of the lane's 16 scores, id in the low mantissa bits -> slot of tile t"""
    m1, m2, a = T[0], T[1], T[2]                # (m1, m2) is stored as a pair: 64-bit register tuples start at even registers
    o = [f"v_and_or_b32 v{m1}, v0, s{S_MASK}, 0", f"v_mov_b32_e32 v{m2}, 0x7f800000"]
    for e in range(1, 16) if on else []:
        o += [f"v_and_or_b32 v{e}, v{e}, s{S_MASK}, {e}",
              f"v_med3_f32 v{m2}, v{m1}, v{m2}, v{e}",
              f"v_min_f32_e32 v{m1}, v{m1}, v{e}"]
    o += [f"v_add_u32_e32 v{a}, s{S_MS}, v{V_SLOTW}", f"ds_write_b64 v{a}, v[{m1}:{m2}]"]
    return o


def dpp_reduce(op, dst, src):
    """all-reduce over the 16 lanes of a DPP row: four steps, two wait states in front of each DPP read of a fresh register"""
    o = ["s_nop 1", f"{op} v{dst}, v{src}, v{src} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"]
    for ctl in ("quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
        o += ["s_nop 1", f"{op} v{dst}, v{dst}, v{dst} {ctl} row_mask:0xf bank_mask:0xf bound_ctrl:1"]
    return o


def convert_load(st, conv=True, loads=True, wait=True, nt=True, hot=False):
    """C: tile t + 2 from register set st -> fp16 image + {eps sE, flag}; then the rows of tile t + 4 into the set"""
    x = XSET[st]
    o = []
    if conv:
        skip = label("noconv")
        o += [f"s_add_i32 s64, s{S_T}, 2", f"s_cmp_ge_i32 s64, {S_NTL}", f"s_cbranch_scc1 {skip}"]
        if wait:
            o.append("s_waitcnt vmcnt(2)")      # the older set has arrived: at most the other set's two loads are outstanding
        lo0, hi0, lo1, hi1, hh, u, v, w = T
        o += [f"v_cvt_pk_f16_f32 v{lo0}, v{x[0]}, v{x[1]}", f"v_cvt_pk_f16_f32 v{hi0}, v{x[2]}, v{x[3]}",
              f"v_cvt_pk_f16_f32 v{lo1}, v{x[4]}, v{x[5]}", f"v_cvt_pk_f16_f32 v{hi1}, v{x[6]}, v{x[7]}",
              f"v_mov_b32_e32 v{hh}, 0",
              f"v_dot2c_f32_f16_e32 v{hh}, v{lo0}, v{lo0}", f"v_dot2c_f32_f16_e32 v{hh}, v{hi0}, v{hi0}",
              f"v_dot2c_f32_f16_e32 v{hh}, v{lo1}, v{lo1}", f"v_dot2c_f32_f16_e32 v{hh}, v{hi1}, v{hi1}",
              f"v_add_u32_e32 v{u}, s{S_ZWR}, v{V_CONVA}",
              f"ds_write_b64 v{u}, v[{lo0}:{hi0}]", f"ds_write_b64 v{u}, v[{lo1}:{hi1}] offset:256"]
        o += dpp_reduce("v_add_f32_dpp", hh, hh)
        o += [f"ds_swizzle_b32 v{u}, v{hh} offset:swizzle(SWAP,16)", "s_waitcnt lgkmcnt(0)",
              f"v_add_f32_e32 v{hh}, v{hh}, v{u}",                      # |h(z)|^2 of the row
              f"v_sqrt_f32_e32 v{u}, v{hh}", "s_nop 0",                  # hn
              # a-priori rounding bound (see convert() in the header of vq_pipe.hip): the same operations, in the same order,
              # as the C++ expression the compiler built for vq_stream16.hip with DVQ_MEASURE_DZ=0
              f"v_mul_f32_e32 v{v}, 0x3a0020cd, v{u}", f"v_add_f32_e32 v{v}, 0x3500d959, v{v}",     # dzn = hn * 4.8877e-4 + 4.8e-7
              f"v_add_f32_e32 v{w}, v{u}, v{v}", f"v_mul_f32_e32 v{w}, 0x3f800347, v{w}",           # zn = (hn + dzn) * 1.0001
              f"v_add_f32_e32 v{lo0}, {S_EMAX}, v{w}",                                              # u = zn + emax
              f"v_mul_f32_e32 v{hi0}, {S_EMAX}, v{v}", f"v_mul_f32_e32 v{lo1}, {S_DEMAX}, v{w}",  # dzn emax, zn demax
              f"v_add_f32_e32 v{hi0}, v{hi0}, v{lo1}", f"v_mul_f32_e32 v{lo1}, {S_DEMAX}, v{v}",   # + dzn demax
              f"v_add_f32_e32 v{hi0}, v{lo1}, v{hi0}",
              f"v_mul_f32_e32 v{lo1}, 0x38d3cff6, v{lo0}", f"v_mul_f32_e32 v{hi0}, 0x408020c5, v{hi0}",   # 1.01e-4 u, 4.004 (...)
              f"v_mul_f32_e32 v{lo1}, v{lo0}, v{lo1}", f"v_add_f32_e32 v{hi0}, v{lo1}, v{hi0}",           # eps
              f"v_mul_f32_e32 v{lo1}, {S_SEF}, v{hi0}",                                                   # eps sE  -> lo1
              f"v_cmp_nge_f32_e32 vcc, s{S_BIG}, v{hh}",                  # !(hh <= 3e38): NaN / Inf / fp16 overflow
              f"v_cmp_nge_f32_e64 s[64:65], s{S_BIG}, v{lo1}",            # !(eps sE <= 3e38)
              "s_nop 3",
              "s_or_b64 s[64:65], s[64:65], vcc", f"s_or_b64 s[64:65], s[64:65], s[{S_BADE}:{S_BADE + 1}]",
              "s_nop 1",
              f"v_cndmask_b32_e64 v{hi1}, 0, 1, s[64:65]",                # flag -> hi1 (= lo1 + 1: the pair {eps sE, flag})
              f"v_add_u32_e32 v{u}, s{S_RSWR}, v{V_RSA}",
              f"ds_write_b64 v{u}, v[{lo1}:{hi1}]",
              f"{skip}:"]
    if loads:
        # always issued (the counted wait above needs the same number of loads in flight in every period): tiles behind the
        # workgroup's last one and rows behind the end of the data read the last row again
        ntm = " nt" if nt else ""
        if hot:
            o += ["s_mov_b32 s64, %[row0]"]
        o += [(f"s_min_i32 s64, s{S_LDROW}, s{S_MM1}" if not hot else f"s_min_i32 s64, s64, s{S_MM1}"), "s_max_i32 s64, s64, 0",
              f"s_add_i32 s65, s{S_LDROW}, 1", f"s_cmp_lt_i32 s65, {S_M}", "s_cselect_b32 s65, -1, 0x3ff",
              f"v_and_b32_e32 v{T[0]}, s65, v{V_LOADOFF}",
              "s_mov_b32 s65, 0", "s_lshl_b64 s[64:65], s[64:65], 10",
              f"s_add_u32 s64, s64, %[zplo]", f"s_addc_u32 s65, s65, %[zphi]",
              f"global_load_dwordx4 v[{x[0]}:{x[3]}], v{T[0]}, s[64:65]{ntm}",
              f"global_load_dwordx4 v[{x[4]}:{x[7]}], v{T[0]}, s[64:65] offset:512{ntm}"]
    o.append(f"s_add_i32 s{S_LDROW}, s{S_LDROW}, {S_ROWSTEP}")
    return o


def half_decision(b1, b2, bad, rowadd, wm, und, slowf):
    """scalar decision for one of the wave's two rows: winner mask (b1 if the row is live and decided, else 0), undecided flag,
    slow flag (bad row or no score within eps)"""
    return [f"s_bcnt1_i32_b32 s74, s{b1}", f"s_bcnt1_i32_b32 s75, s{b2}",
            "s_cmp_eq_u32 s74, 1", "s_cselect_b32 s76, 1, 0",
            "s_cmp_eq_u32 s75, 0", "s_cselect_b32 s77, 1, 0", "s_and_b32 s76, s76, s77",
            f"s_cmp_eq_u32 s{bad}, 0", "s_cselect_b32 s77, 1, 0", "s_and_b32 s76, s76, s77",          # unique
            f"s_add_i32 s78, s{S_MROW}, {rowadd}", f"s_cmp_lt_i32 s78, {S_M}", "s_cselect_b32 s77, 1, 0",   # live
            "s_cmp_ge_i32 s78, 0", "s_cselect_b32 s78, 1, 0", "s_and_b32 s77, s77, s78",
            "s_and_b32 s78, s76, s77", "s_cmp_lg_u32 s78, 0", f"s_cselect_b32 s{wm}, s{b1}, 0",
            f"s_andn2_b32 s{und}, s77, s76",
            "s_cmp_eq_u32 s74, 0", "s_cselect_b32 s78, 1, 0",
            f"s_cmp_lg_u32 s{bad}, 0", "s_cselect_b32 s79, 1, 0", f"s_or_b32 s{slowf}, s78, s79"]


def merge(on=True):
    """M: merge of the wave's two rows of tile t - 1 (32 lanes per row, one slot per lane)"""
    if not on:
        return []
    skip, nowin = label("nomerge"), label("nowin")
    m1, m2, eps, flg, thr, slw, a, u = T         # pairs (m1, m2), (eps, flg), (thr, slw) start at even registers
    ent, zero, val = 24, 25, 26                  # more of the fragment registers
    o = [f"s_cmp_lt_i32 s{S_T}, 1", f"s_cbranch_scc1 {skip}",
         f"s_sub_u32 s64, s{S_MS}, {MS_BUF}",
         f"v_add_u32_e32 v{a}, s64, v{V_MRGA}", f"ds_read_b64 v[{m1}:{m2}], v{a}",
         f"v_add_u32_e32 v{a}, s{S_RSRD}, v{V_RSA}", f"ds_read_b64 v[{eps}:{flg}], v{a}",
         "s_waitcnt lgkmcnt(0)"]
    o += dpp_reduce("v_min_f32_dpp", thr, m1)
    o += [f"ds_swizzle_b32 v{u}, v{thr} offset:swizzle(SWAP,16)", "s_waitcnt lgkmcnt(0)",
          f"v_min_f32_e32 v{thr}, v{thr}, v{u}",
          f"v_add_f32_e32 v{thr}, v{eps}, v{thr}",                       # row minimum + eps sE
          f"v_cmp_le_f32_e64 s[64:65], v{m1}, v{thr}", f"v_cmp_le_f32_e64 s[66:67], v{m2}, v{thr}",
          f"v_cmp_ne_u32_e64 s[68:69], 0, v{flg}",
          "s_nop 3"]
    o += half_decision(64, 66, 68, 0, 70, 80, 82)
    o += half_decision(65, 67, 69, 1, 71, 81, 83)
    # winner lanes of decided rows write the entry: key = (0 : entry)
    o += [f"s_sub_u32 s72, s{S_T}, 1", "s_lshl_b32 s72, s72, 8", f"s_add_u32 s72, s72, {L_RES - L_RS}",   # (t - 1) * 256 + L_RES - L_RS
          "s_or_b64 s[74:75], s[70:71], s[70:71]", f"s_cbranch_scc0 {nowin}",
          "s_mov_b64 exec, s[70:71]",
          f"v_and_b32_e32 v{u}, 15, v{m1}", f"v_lshlrev_b32_e32 v{val}, 1, v{u}", f"v_and_b32_e32 v{val}, 24, v{val}",
          f"v_and_b32_e32 v{u}, 3, v{u}", f"v_or3_b32 v{ent}, v{u}, v{val}, v{V_ENTB}", f"v_mov_b32_e32 v{zero}, 0",
          f"v_add_u32_e32 v{a}, s72, v{V_RSA}", f"ds_write_b64 v{a}, v[{ent}:{zero}]",
          "s_mov_b64 exec, -1",
          f"{nowin}:"]
    # undecided rows: {threshold, slow flag} -> L_REC[rowslot], rowslot -> the wave's list
    for half, (und, slowf, lane_exec) in enumerate(((80, 82, ("1", "0")), (81, 83, ("0", "1")))):
        nound = label("nound")
        o += [f"s_cmp_eq_u32 s{und}, 0", f"s_cbranch_scc1 {nound}",
              f"s_mov_b32 exec_lo, {lane_exec[0]}", f"s_mov_b32 exec_hi, {lane_exec[1]}",
              f"v_mov_b32_e32 v{slw}, s{slowf}",
              f"s_add_u32 s74, s72, {L_REC - L_RES}", f"v_add_u32_e32 v{a}, s74, v{V_RSA}",
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
    o.append(f"{skip}:")
    return o


def advance():
    return [f"s_add_i32 s{S_T}, s{S_T}, 1",
            f"s_add_u32 s{S_ZRD}, s{S_ZRD}, {Z16_BUF}", f"s_cmp_eq_u32 s{S_ZRD}, {NZB * Z16_BUF}", f"s_cselect_b32 s{S_ZRD}, 0, s{S_ZRD}",
            f"s_add_u32 s{S_ZWR}, s{S_ZWR}, {Z16_BUF}", f"s_cmp_eq_u32 s{S_ZWR}, {NZB * Z16_BUF}", f"s_cselect_b32 s{S_ZWR}, 0, s{S_ZWR}",
            f"s_add_i32 s{S_MS}, s{S_MS}, {MS_BUF}",
            f"s_add_u32 s{S_RSWR}, s{S_RSWR}, 256", f"s_and_b32 s{S_RSWR}, s{S_RSWR}, 0x3ff",
            f"s_add_u32 s{S_RSRD}, s{S_RSRD}, 256", f"s_and_b32 s{S_RSRD}, s{S_RSRD}, 0x3ff",
            f"s_add_i32 s{S_MROW}, s{S_MROW}, {S_ROWSTEP}"]


def period(st, rot, abl):
    P = chain(mfma=not (abl & 32), reads=not (abl & 64))
    S = scores(on=not (abl & 8))
    C = convert_load(st, conv=not (abl & 4), loads=not (abl & 1), wait=not (abl & 1) and not (abl & 128), nt=not (abl & 256), hot=bool(abl & 512))
    Mg = merge(on=not (abl & 2))
    o = ["s_waitcnt lgkmcnt(0)"]
    if not (abl & 16):
        o.append("s_barrier")
    if not rot:
        return o + P + S + C + Mg
    l0, l1, l2, le = label("q0"), label("q1"), label("q2"), label("qe")
    o += [f"s_cmp_eq_u32 {S_Q}, 0", f"s_cbranch_scc1 {l0}", f"s_cmp_eq_u32 {S_Q}, 1", f"s_cbranch_scc1 {l1}",
          f"s_cmp_eq_u32 {S_Q}, 2", f"s_cbranch_scc1 {l2}"]
    # labels inside a phase are unique per instance: regenerate the phases for every copy
    def phases():
        return (chain(mfma=not (abl & 32), reads=not (abl & 64)), scores(on=not (abl & 8)),
                convert_load(st, conv=not (abl & 4), loads=not (abl & 1), wait=not (abl & 1) and not (abl & 128), nt=not (abl & 256), hot=bool(abl & 512)), merge(on=not (abl & 2)))
    def seq(order):
        p, s, c, m = phases()
        ph = {"P": stamp(1) + p + stamp(2), "S": s + stamp(3), "C": stamp(4) + c + stamp(5), "M": stamp(6) + m + stamp(7)}
        out = stamp(0)
        for x in order:
            out += ph[x]
        return out
    o += seq("CMPS") + [f"s_branch {le}", f"{l0}:"]                       # q = 3
    o += seq("PSCM") + [f"s_branch {le}", f"{l1}:"]                       # q = 0
    o += seq("CPSM") + [f"s_branch {le}", f"{l2}:"]                       # q = 1
    o += seq("MCPS") + [f"{le}:"]                                         # q = 2
    return o


def program(rot=True, abl=0, stamps=False):
    uid[0] = 0
    STAMPS[0] = stamps
    lane = T[0]
    o = [  # ---- constants of the block
        f"s_cmp_eq_u32 %[evalid], 0", f"s_cselect_b64 s[{S_BADE}:{S_BADE + 1}], -1, 0",
        f"s_mov_b32 s{S_MASK}, 0xffffffe0", f"s_mov_b32 s{S_BIG}, 0x7f61b1e6", f"s_sub_u32 s{S_MM1}, {S_M}, 1",
        f"s_mov_b32 s{S_UND}, 0",
        # ---- rows of tiles 0 and 1 first (HBM latency), then the wave's 16 codebook fragments (L2)
        f"s_mov_b32 s{S_LDROW}, %[row0]"]
    if not (abl & 1):
        o += convert_load(0, conv=False) + convert_load(1, conv=False)
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
            o.append(f"global_load_dwordx4 v[{AF0 + 4 * s}:{AF0 + 4 * s + 3}], v{lane}, s[64:65] offset:{1024 * s4}")
    o += ["s_waitcnt vmcnt(0)",
          # ---- conversions of tiles 0 and 1, loads of tiles 2 and 3: the C phase at t = -2 and t = -1
          f"s_mov_b32 s{S_T}, -2",
          f"s_mov_b32 s{S_ZRD}, {Z16_BUF}", f"s_mov_b32 s{S_ZWR}, 0", f"s_mov_b32 s{S_MS}, {-2 * MS_BUF}",
          f"s_mov_b32 s{S_RSWR}, 0", f"s_mov_b32 s{S_RSRD}, 256",
          f"s_mul_i32 s64, {S_ROWSTEP}, 3", f"s_sub_u32 s{S_MROW}, %[row0], s64"]
    o += convert_load(0, conv=not (abl & 4), loads=not (abl & 1), wait=False) + advance()
    o += convert_load(1, conv=not (abl & 4), loads=not (abl & 1), wait=False) + advance()
    # ---- (prologue done: a 100 MHz stamp per wave into L_DBG, read by the diagnostics build only)
    o += ["s_memrealtime s[64:65]", f"s_lshl_b32 s66, {S_WAVE}, 3", f"s_add_u32 s66, s66, {S_LDSB}", f"s_add_u32 s66, s66, {L_DBG}",
          "s_mov_b32 exec_lo, 1", "s_mov_b32 exec_hi, 0", "s_waitcnt lgkmcnt(0)",
          f"v_mov_b32_e32 v{T[0]}, s64", f"v_mov_b32_e32 v{T[1]}, s65", f"v_mov_b32_e32 v{T[2]}, s66",
          f"ds_write_b64 v{T[2]}, v[{T[0]}:{T[1]}]", "s_mov_b64 exec, -1"]
    # ---- tile loop, two periods per trip (register sets A, B)
    loop, done = label("loop"), label("done")
    o += [f"{loop}:"]
    o += period(0, rot, abl) + advance() + [f"s_cmp_ge_i32 s{S_T}, {S_NTL}", f"s_cbranch_scc1 {done}"]
    o += period(1, rot, abl) + advance() + [f"s_cmp_lt_i32 s{S_T}, {S_NTL}", f"s_cbranch_scc1 {loop}"]
    o += [f"{done}:",
          "s_waitcnt vmcnt(0)",                                           # rows behind the last tile (never used) have landed
          # undecided-row count of this wave -> L_CNT[4 + wave]
          f"s_lshl_b32 s64, {S_WAVE}, 2", f"s_add_u32 s64, s64, {S_LDSB}", f"s_add_u32 s64, s64, {L_CNT + 16}",
          "s_mov_b32 exec_lo, 1", "s_mov_b32 exec_hi, 0",
          f"v_mov_b32_e32 v{T[0]}, s64", f"v_mov_b32_e32 v{T[1]}, s{S_UND}", f"ds_write_b32 v{T[0]}, v{T[1]}",
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
        f.write("#define VQP_ASM_DEFAULT \\\n    " + lit(program(rot=True)) + "\n")
        f.write("#ifdef DVQ_DIAG\n")
        f.write("#define VQP_ASM_NOROT \\\n    " + lit(program(rot=False)) + "\n")
        f.write("#define VQP_ASM_STAMPS \\\n    " + lit(program(rot=True, stamps=True)) + "\n")
        for abl in (1, 2, 4, 8, 32, 96, 15, 130, 256, 512, 3):
            f.write(f"#define VQP_ASM_ABL{abl} \\\n    " + lit(program(rot=True, abl=abl)) + "\n")
        f.write("#endif\n")
        clob = [f'"v{i}"' for i in list(range(0, 48)) + [V_ZA] + list(range(64, 128))] + [f'"s{i}"' for i in range(S_FIRST, S_LAST + 1)]
        f.write("#define VQP_ASM_CLOBBERS " + ", ".join(clob) + ', "vcc", "scc", "memory"\n')
    n = len(program(rot=True))
    print("wrote", os.path.normpath(out), "FD =", FD, "instructions+labels in the default block:", n)
    if "--list" in sys.argv:
        print("\n".join(program(rot=True)))


if __name__ == "__main__":
    main()
