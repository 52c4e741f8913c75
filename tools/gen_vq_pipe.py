#!/usr/bin/env python3
"""Writes d-vqvae_amd/csrc/vq_pipe_loop.h: the prologue and the eight tile periods of vq_pipe.hip as ONE hand-written instruction
block.

Why by hand: the loop keeps 104 of a wave's 128 registers busy for its whole life (64 codebook fragment registers, 16 accumulators,
8 registers of tile fragments in flight, 16 registers of rows in flight from HBM).  Written in C++ (five attempts, round 6) the
register allocator either spilled codebook fragments, or moved registers whose loads were still in flight (an asm load's destination
counts as written at the end of the statement), or parked values in the accumulator registers the block had been given.  Here every
register has one owner:

  v0-v15    accumulators of the tile (16 MFMAs per tile, 32 rows x 32 entries)
  v16-v23   tile fragments in flight (two); behind the matrix phase: temporaries of the scores
  v24-v31   temporaries of the vector work that runs inside the matrix phase
  v32-v39   rows of the even tile in flight (two 16-byte pieces of each of the wave's two rows)       -- "set A"
  v40-v47   rows of the odd tile in flight                                                              -- "set B"
  v48-v55   lane constants (operands of the statement, pinned):
            v48 zbase   LDS address of the lane's fragment column in fp16 tile buffer 0:  L_Z16 + (lane % 32) * Z16_ROW + 16 * (lane / 32)
            v49 slotw   LDS address of the lane's merge slot, tile 0:   L_MS + (lane % 32) * MS_ROW + (2 wave + lane / 32) * 8
            v50 eesa    LDS address of the lane's accumulator start values:  L_EES + (32 wave + 4 (lane / 32)) * 4
            v51 loadoff byte offset of the lane's first 16 bytes in the wave's two rows:  (lane / 32) * 1024 + (lane % 32) * 16
            v52 conva   LDS address of the lane's 8 bytes in the fp16 image of its row, buffer 0:  L_Z16 + (2 wave + lane/32) * Z16_ROW + 8 (lane % 32)
            v53 rsa     LDS address of eps sE of the lane's row, tile 0:  L_RS + (2 wave + lane / 32) * 4
            v54 mrga    LDS address of the lane's slot to merge, tile 0:  L_MS + (2 wave + lane / 32) * MS_ROW + 8 (lane % 32)
            v55 code    0x80000000 | (lane % 32) << 4: what a decided row's winner lane stores (| id of the accumulator register)
  v57       resa: LDS address of the result of the lane's row, tile 0:  L_RES + (2 wave + lane / 32) * 8   (derived from v53)
  v58, v59  threshold of the merged rows (kept for the record), scratch
  v56, v60-v63   NOT used (the compiler needs a few registers of its own around the statement)
  v64-v127  the wave's 16 codebook fragments (A operands)
  s64-s99   the block's scalars (the operands, read in place, live below)

The eight periods are written out (a workgroup has at most eight tiles: every ring offset is an immediate, nothing is counted at run
time; behind period t the block leaves when the workgroup has no tile t + 1).  Per period t, between two workgroup barriers, every
wave runs the same program (round 6, third structure.  The first ran the pieces one after the other in an order rotated over the
four waves of a SIMD: per-wave stamps showed vector pieces of 500-1 300 cycles each and the matrix pipe idle 60 % of the period.
The second interleaved them with the MFMAs: the four waves of a SIMD still finished one after the other, 1 040 instructions per SIMD
and period at one instruction per ~4 cycles.  This one has 140 instructions per wave and period instead of 290):
  P  the 16 MFMAs of tile t (buffer t % 3), and IN THEIR GAPS, instruction by instruction,
  C  the conversion of the wave's two rows of tile t + 2 (register set t % 2; eps sE as a quadratic in |h(z)|, coefficients from
     the host side) and the loads of tile t + 4 into that set,
  M  the merge of the wave's two rows of tile t - 1 (decision on the scalar unit from two compare masks), without branches;
  S  (min, second) of the lane's 16 scores -> slot of tile t (groups of three: min3 + med3, merged pairwise: 40 instructions);
  R  the (rare) record of a row the merge could not decide.
Every LDS wait is COUNTED: the scheduler below merges the streams and derives each s_waitcnt lgkmcnt(N) from the program order of the
always-executed LDS operations behind the one that is needed (operations of a wave complete in order; a conditional one in between
only makes a wait longer).

usage: gen_vq_pipe.py [--list]"""
import os
import sys

FD = 2
DEFAULT_VAR = 1                                # structure switches of period(): see there; stamps = 8 (diagnostics)

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
L_RS = L_MS + MAX_TILES * MS_BUF                  # [MAX_TILES][32] f32 eps sE (inf / NaN: the row takes the all-entries path)
L_EES = L_RS + MAX_TILES * TILE * 4
L_RES = L_EES + K * 4
L_REC = L_RES + MAX_TILES * TILE * 8              # [MAX_TILES*32] 8-byte slots, low word: threshold of a row the loop left undecided
L_PAIR = L_REC + MAX_TILES * TILE * 8
L_SLOW = L_PAIR + PAIR_CAP * 4
L_UND = L_SLOW + MAX_TILES * TILE * 2
UND_PER_WAVE = 16
L_CNT = L_UND + NWV * UND_PER_WAVE * 2
L_DBG = L_CNT + 128
LDS_BYTES = L_DBG + 128 + NWV * 16 * 4            # [16] u64 prologue stamps, then [16 waves][2 tiles][8] u32 period stamps

# ---- registers
FRAG = [f"v[{16 + 4 * i}:{16 + 4 * i + 3}]" for i in range(FD)]
XSET = {0: list(range(32, 40)), 1: list(range(40, 48))}
V_ZBASE, V_SLOTW, V_EESA, V_LOADOFF, V_CONVA, V_RSA, V_MRGA, V_CODE = 48, 49, 50, 51, 52, 53, 54, 55
V_RESA, V_THR, V_VAL = 57, 58, 59
W = [24, 25, 26, 27, 28, 29, 30, 31]
TS = [16, 17, 18, 19]

S_UND, S_MASK, S_MM1 = 92, 93, 95
S_FIRST, S_LAST = 64, 99
# s64-s79 temporaries; s80 / s81: nonzero = nothing to record for the row (decided, or not live); s82:83 not-live masks
S_M, S_NTL, S_ROWSTEP, S_EA, S_EB, S_EC, S_LDSB, S_WAVE = ("%[M]", "%[ntl]", "%[rowstep]", "%[epsa]", "%[epsb]", "%[epsc]",
                                                           "%[ldsb]", "%[wave]")   # and %[q] = wave / 4, %[row0], %[zplo], %[zphi], %[img]

uid = [0]


def label(name):
    uid[0] += 1
    return f".Lvqp_{name}_{uid[0]}_%="


def I(text, lds=None, need=(), glue=False):
    """one instruction: lds = tag of the always-executed LDS operation it issues (None: not an LDS operation; "" an untagged one),
    need = tags of LDS operations whose results it uses, glue = stays with the next instruction (exec-masked groups, SCC pairs)"""
    return {"t": text, "lds": lds, "need": tuple(need), "glue": glue}


def off(n):
    assert 0 <= n < 65536, n
    return f" offset:{n}" if n else ""


def dpp_reduce(op, dst, src):
    """all-reduce over the 16 lanes of a DPP row: four steps, two wait states in front of each DPP read of a fresh register"""
    o = [I("s_nop 1", glue=True), I(f"{op} v{dst}, v{src}, v{src} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")]
    for ctl in ("quad_perm:[2,3,0,1]", "row_half_mirror", "row_mirror"):
        o += [I("s_nop 1", glue=True), I(f"{op} v{dst}, v{dst}, v{dst} {ctl} row_mask:0xf bank_mask:0xf bound_ctrl:1")]
    return o


def products(t, mfma=True, reads=True, prio=False):
    """matrix phase of tile t: head, then 16 steps (the instructions in front of each gap)"""
    zb = (t % NZB) * Z16_BUF
    head = [I(f"ds_read_b128 v[0:3], v{V_EESA}", lds="ci0"), I(f"ds_read_b128 v[4:7], v{V_EESA} offset:32", lds="ci1"),
            I(f"ds_read_b128 v[8:11], v{V_EESA} offset:64", lds="ci2"), I(f"ds_read_b128 v[12:15], v{V_EESA} offset:96", lds="ci3")]
    if not reads:
        return head, [[I("s_nop 0", need=("ci3",))]] + [[] for _ in range(15)]
    for i in range(FD):
        head.append(I(f"ds_read_b128 {FRAG[i]}, v{V_ZBASE}{off(zb + 32 * i)}", lds=f"f{i}"))
    steps = []
    for s in range(16):
        need = (f"f{s}",) + (("ci3",) if s == 0 else ())
        st = [I(f"v_mfma_f32_32x32x16_f16 v[0:15], v[{64 + 4 * s}:{64 + 4 * s + 3}], {FRAG[s % FD]}, v[0:15]", need=need) if mfma
              else I("s_nop 0", need=need)]
        if s + FD < 16:
            st.append(I(f"ds_read_b128 {FRAG[s % FD]}, v{V_ZBASE}{off(zb + 32 * (s + FD))}", lds=f"f{s + FD}"))
        if prio and s % 4 == 0:                    # the further behind a wave is, the higher its priority: the four waves of a SIMD
            st.insert(0, I(f"s_setprio {3 - s // 4}"))     # go through the period side by side instead of one after the other
        steps.append(st)
    if prio:
        steps[-1].append(I("s_setprio 0"))
    return head, steps


def scores(t, on=True):
    """(min, second) of the lane's 16 scores, id in the low mantissa bits -> slot of tile t.  Groups of three (min3 / med3 give a
    group's two smallest), merged into the running pair: second = med3(m1, g1, min(m2, g2))."""
    m1, m2, g1, g2 = TS
    o = [I("s_nop 15")]                         # MFMA result -> vector ALU read (8-pass XDL: 12 states; 16 here)
    if not on:
        # (timing-only ablation: two raw scores, but ORDERED -- the tail's pair count relies on second >= first)
        return o + [I(f"v_min_f32_e32 v{m1}, v0, v1"), I(f"v_max_f32_e32 v{m2}, v0, v1"), I(f"ds_write_b64 v{V_SLOTW}, v[{m1}:{m2}]{off(t * MS_BUF)}", lds="")]
    for e in range(16):
        o.append(I(f"v_and_or_b32 v{e}, v{e}, s{S_MASK}, {e}"))
    o += [I(f"v_min3_f32 v{m1}, v0, v1, v2"), I(f"v_med3_f32 v{m2}, v0, v1, v2")]
    for g in (3, 6, 9, 12):
        o += [I(f"v_min3_f32 v{g1}, v{g}, v{g + 1}, v{g + 2}"), I(f"v_med3_f32 v{g2}, v{g}, v{g + 1}, v{g + 2}"),
              I(f"v_min_f32_e32 v{g2}, v{m2}, v{g2}"), I(f"v_med3_f32 v{m2}, v{m1}, v{g1}, v{g2}"), I(f"v_min_f32_e32 v{m1}, v{m1}, v{g1}")]
    o += [I(f"v_med3_f32 v{m2}, v{m1}, v{m2}, v15"), I(f"v_min_f32_e32 v{m1}, v{m1}, v15"),
          I(f"ds_write_b64 v{V_SLOTW}, v[{m1}:{m2}]{off(t * MS_BUF)}", lds="")]
    return o


def convert(tile, st, on=True, wait=2):
    """C without branches: tile `tile` from register set st -> fp16 image + eps sE.  Behind the workgroup's last tile it converts the
    (valid, never used) rows the always-issued loads brought: the buffers it writes are free.
    eps sE = A hn^2 + B hn + C, hn = |h(z)|: the a-priori bound of vq_stream16.hip (DVQ_MEASURE_DZ=0) multiplied out on the host
    side (dvq_vq_pipe_eps in vq_pipe.hip), rounded up; C = inf when the codebook image is invalid."""
    if not on:
        return []
    x = XSET[st]
    lo0, hi0, lo1, hi1, hh, u = W[:6]
    o = []
    if wait is not None:
        o.append(I(f"s_waitcnt vmcnt({wait})"))  # the set has arrived (at most the other set's two loads are outstanding)
    zb = (tile % NZB) * Z16_BUF
    o += [I(f"v_cvt_pk_f16_f32 v{lo0}, v{x[0]}, v{x[1]}"), I(f"v_cvt_pk_f16_f32 v{hi0}, v{x[2]}, v{x[3]}"),
          I(f"v_cvt_pk_f16_f32 v{lo1}, v{x[4]}, v{x[5]}"), I(f"v_cvt_pk_f16_f32 v{hi1}, v{x[6]}, v{x[7]}"),
          I(f"v_dot2_f32_f16 v{hh}, v{lo0}, v{lo0}, 0"),
          I(f"v_dot2c_f32_f16_e32 v{hh}, v{hi0}, v{hi0}"),
          I(f"v_dot2c_f32_f16_e32 v{hh}, v{lo1}, v{lo1}"), I(f"v_dot2c_f32_f16_e32 v{hh}, v{hi1}, v{hi1}"),
          I(f"ds_write_b64 v{V_CONVA}, v[{lo0}:{hi0}]{off(zb)}", lds=""), I(f"ds_write_b64 v{V_CONVA}, v[{lo1}:{hi1}]{off(zb + 256)}", lds="")]
    o += dpp_reduce("v_add_f32_dpp", hh, hh)
    o += [I(f"ds_swizzle_b32 v{u}, v{hh} offset:swizzle(SWAP,16)", lds="csw"),
          I(f"v_add_f32_e32 v{hh}, v{hh}, v{u}", need=("csw",)),                 # |h(z)|^2 of the row
          I(f"v_sqrt_f32_e32 v{hh}, v{hh}", glue=True), I("s_nop 0"),             # hn
          I(f"v_mul_f32_e32 v{u}, {S_EA}, v{hh}"), I(f"v_add_f32_e32 v{u}, {S_EB}, v{u}"),
          I(f"v_mul_f32_e32 v{u}, v{u}, v{hh}"), I(f"v_add_f32_e32 v{u}, {S_EC}, v{u}"),
          I(f"ds_write_b32 v{V_RSA}, v{u}{off(tile * TILE * 4)}", lds="")]
    return o


def loads(tile, st, on=True, nt=True, hot=False):
    """the rows of tile `tile` into register set st.  Always issued for tiles 0..7 (the counted vmcnt wait needs the same number of
    loads in flight): tiles behind the workgroup's last one and rows behind the end of the data read the last row again."""
    if not on:
        return []
    x = XSET[st]
    ntm = " nt" if nt else ""
    u = W[5]
    o = [I(f"s_mul_i32 s64, {S_ROWSTEP}, {0 if hot else tile}"), I("s_add_i32 s64, s64, %[row0]"),
         I("s_add_i32 s65, s64, 1"), I(f"s_cmp_lt_i32 s65, {S_M}", glue=True), I("s_cselect_b32 s65, -1, 0x3ff"),
         I(f"v_and_b32_e32 v{u}, s65, v{V_LOADOFF}"),
         I(f"s_min_i32 s64, s64, s{S_MM1}"), I("s_max_i32 s64, s64, 0"),
         I("s_mov_b32 s65, 0"), I("s_lshl_b64 s[64:65], s[64:65], 10"),
         I("s_add_u32 s64, s64, %[zplo]", glue=True), I("s_addc_u32 s65, s65, %[zphi]"),
         I(f"global_load_dwordx4 v[{x[0]}:{x[3]}], v{u}, s[64:65]{ntm}"),
         I(f"global_load_dwordx4 v[{x[4]}:{x[7]}], v{u}, s[64:65] offset:512{ntm}")]
    return o


def merge(tm, on=True):
    """M without branches: merge of the wave's two rows of tile tm (32 lanes per row, one slot per lane).
    Leaves: s80 / s81 zero when the row must be recorded (live, no unique entry within eps), v58 the threshold."""
    if not on:
        return [I("s_mov_b32 s80, 1"), I("s_mov_b32 s81, 1")]
    m1, m2, eps, u = W[:4]
    thr = V_THR
    o = [I(f"ds_read_b64 v[{m1}:{m2}], v{V_MRGA}{off(tm * MS_BUF)}", lds="ms"),
         I(f"ds_read_b32 v{eps}, v{V_RSA}{off(tm * TILE * 4)}", lds="mr"),
         # rows of the tile that exist: not-live masks of the two rows (all ones: the row is behind the end of the data)
         I(f"s_mul_i32 s74, {S_ROWSTEP}, {tm}"), I("s_add_i32 s74, s74, %[row0]"),
         I(f"s_cmp_lt_i32 s74, {S_M}", glue=True), I("s_cselect_b32 s82, 0, -1"),
         I("s_add_i32 s74, s74, 1"), I(f"s_cmp_lt_i32 s74, {S_M}", glue=True), I("s_cselect_b32 s83, 0, -1")]
    d = dpp_reduce("v_min_f32_dpp", thr, m1)
    d[0]["need"] = ("ms",)                          # (the wait goes in front of the glued s_nop)
    o += d
    o += [I(f"ds_swizzle_b32 v{u}, v{thr} offset:swizzle(SWAP,16)", lds="msw"),
          I(f"v_min_f32_e32 v{thr}, v{thr}, v{u}", need=("msw", "mr")),
          I(f"v_add_f32_e32 v{thr}, v{eps}, v{thr}"),                        # row minimum + eps sE (inf / NaN: nothing is decided)
          I(f"v_cmp_le_f32_e64 s[64:65], v{m1}, v{thr}"), I(f"v_cmp_le_f32_e64 s[66:67], v{m2}, v{thr}", glue=True),
          I("s_nop 3"),
          # decided = exactly one first score within eps, no second score within eps, row live:  winner mask = the compare mask
          I("s_or_b64 s[66:67], s[66:67], s[82:83]"),
          I("s_bcnt1_i32_b32 s68, s64"), I("s_bcnt1_i32_b32 s69, s65"),
          I("s_cmp_eq_u32 s68, 1", glue=True), I("s_cselect_b32 s70, s66, -1"),
          I("s_cmp_eq_u32 s70, 0", glue=True), I("s_cselect_b32 s72, s64, 0"),
          I("s_cmp_eq_u32 s69, 1", glue=True), I("s_cselect_b32 s71, s67, -1"),
          I("s_cmp_eq_u32 s71, 0", glue=True), I("s_cselect_b32 s73, s65, 0"),
          I("s_or_b32 s80, s72, s82"), I("s_or_b32 s81, s73, s83"),
          # winner lanes store 0x80000000 | slot << 4 | register id (exec-masked, no branch; exec = 0: nothing happens)
          I("s_mov_b64 exec, s[72:73]", glue=True),
          I(f"v_and_or_b32 v{u}, v{m1}, 15, v{V_CODE}", glue=True),
          I(f"ds_write_b32 v{V_RESA}, v{u}{off(tm * TILE * 8)}", glue=True),      # (not counted: exec may be 0)
          I("s_mov_b64 exec, -1")]
    return o


def record(tm):
    """R (behind the scores; rare, with branches): a row the merge could not decide: threshold -> L_REC[rowslot], rowslot -> the
    wave's list"""
    o = []
    a, val = W[0], W[1]
    for half, (flag, lane_exec) in enumerate(((80, ("1", "0")), (81, ("0", "1")))):
        nound = label("nound")
        o += [f"s_cmp_lg_u32 s{flag}, 0", f"s_cbranch_scc1 {nound}",
              # the row's 1 KB -> L2, for the canonical chains behind the loop (one dword per 16 bytes touches its eight lines; the
              # destination is never read; one more load in flight only makes the counted vmcnt waits longer)
              f"s_mul_i32 s74, {S_ROWSTEP}, {tm}", "s_add_i32 s74, s74, %[row0]", f"s_add_i32 s74, s74, {half}",
              "s_mov_b32 s75, 0", "s_lshl_b64 s[74:75], s[74:75], 10", "s_add_u32 s74, s74, %[zplo]", "s_addc_u32 s75, s75, %[zphi]",
              f"v_mbcnt_lo_u32_b32 v{a}, -1, 0", f"v_mbcnt_hi_u32_b32 v{a}, -1, v{a}", f"v_lshlrev_b32_e32 v{a}, 4, v{a}",
              f"global_load_dword v{V_VAL}, v{a}, s[74:75]",
              f"s_mov_b32 exec_lo, {lane_exec[0]}", f"s_mov_b32 exec_hi, {lane_exec[1]}",
              f"ds_write_b32 v{V_RESA}, v{V_THR}{off(L_REC - L_RES + tm * TILE * 8)}",
              # rowslot = tm * 32 + 2 wave + half; list address = L_UND + (wave * UND_PER_WAVE + und) * 2
              f"s_lshl_b32 s74, {S_WAVE}, 1", f"s_add_u32 s74, s74, {tm * TILE + half}",
              f"s_mul_i32 s75, {S_WAVE}, {UND_PER_WAVE}", f"s_add_u32 s75, s75, s{S_UND}", "s_lshl_b32 s75, s75, 1",
              f"s_add_u32 s75, s75, {S_LDSB}", f"s_add_u32 s75, s75, {L_UND}",
              f"v_mov_b32_e32 v{a}, s75", f"v_mov_b32_e32 v{val}, s74", f"ds_write_b16 v{a}, v{val}",
              f"s_add_u32 s{S_UND}, s{S_UND}, 1",
              "s_mov_b64 exec, -1",
              f"{nound}:"]
    return o


def schedule(head, steps, filler, gaps=None):
    """head, then per step its instructions followed by a share of the filler (spread over the first `gaps` steps; default: all);
    every `need` becomes a counted s_waitcnt lgkmcnt"""
    order = list(head)
    n = max(min(len(steps), gaps or len(steps)), 1)
    fi = 0
    for s, st in enumerate(steps):
        order += st
        stop = len(filler) * min(s + 1, n) // n
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


def stamp(t, k):
    """diagnostics: shader-clock stamp k (0..7) of periods 3 and 4 -> L_DBG + 128 + wave * 64 + (t - 3) * 32 + 4 k"""
    if not STAMPS[0] or t not in (3, 4):
        return []
    return ["s_memtime s[98:99]",
            f"s_lshl_b32 s64, {S_WAVE}, 6", f"s_add_u32 s64, s64, {S_LDSB}", f"s_add_u32 s64, s64, {L_DBG + 128 + (t - 3) * 32 + 4 * k}",
            "s_waitcnt lgkmcnt(0)", f"v_mov_b32_e32 v{W[0]}, s98", f"v_mov_b32_e32 v{W[1]}, s64", f"ds_write_b32 v{W[1]}, v{W[0]}"]


def period(t, abl, var):
    """var bits: 1 = the waves w >= 8 (the two younger of a SIMD's four) store the scores of tile t at the START of period t + 1, so that
    their vector phase runs beside the older waves' matrix phase instead of behind everybody's (merges then lag two tiles);
    2 = the vector work fills the first ten MFMA gaps only; 4 = static wave priority (block start); 16 = priority by progress:
    3 for MFMAs 0-3, 2 for 4-7, 1 for 8-11, 0 behind (a wave that is behind overtakes)"""
    lag = 2 if var & 1 else 1
    st = t % 2
    head, steps = products(t, mfma=not (abl & 32), reads=not (abl & 64), prio=bool(var & 16))
    filler = []
    if t + 2 < MAX_TILES:                          # tile t + 2 from its register set (behind it nothing loads into the other set:
        wait = None if (abl & 1) or (abl & 128) else (2 if t + 3 < MAX_TILES else 0)       # the last set waits for everything)
        filler += convert(t + 2, st, on=not (abl & 4), wait=wait)
    if t + 4 < MAX_TILES:
        filler += loads(t + 4, st, on=not (abl & 1), nt=not (abl & 256), hot=bool(abl & 512))
    if t >= lag:
        filler += merge(t - lag, on=not (abl & 2))
    o = ["s_waitcnt lgkmcnt(0)"]
    if not (abl & 16):
        o.append("s_barrier")
    o += stamp(t, 0)
    if (var & 1) and t >= 1:
        skip = label("nodefer")
        o += ["s_cmp_lt_u32 %[q], 2", f"s_cbranch_scc1 {skip}"] + [i["t"] for i in scores(t - 1, on=not (abl & 8))[1:]] + [f"{skip}:"]
    o += stamp(t, 1)
    o += schedule(head, steps, filler, gaps=10 if var & 2 else None)
    o += stamp(t, 2)
    sc = [i["t"] for i in scores(t, on=not (abl & 8))]
    if var & 1:
        skip = label("deferred")
        o += ["s_cmp_ge_u32 %[q], 2", f"s_cbranch_scc1 {skip}"] + sc + [f"{skip}:"]
    else:
        o += sc
    o += stamp(t, 3)
    if t >= lag and not (abl & 2):
        o += record(t - lag)
    o += stamp(t, 4)
    return o


def leave(t, abl, var):
    """behind the workgroup's last period t: the deferred scores of that tile"""
    if not (var & 1):
        return []
    skip = label("left")
    return ["s_cmp_lt_u32 %[q], 2", f"s_cbranch_scc1 {skip}"] + [i["t"] for i in scores(t, on=not (abl & 8))] + [f"{skip}:"]


def program(abl=0, var=0, stamps=False):
    uid[0] = 0
    STAMPS[0] = stamps
    lane = W[0]
    o = []
    if var & 4:                                    # static priority: the younger a wave of a SIMD, the higher (age breaks ties the other way)
        l1, l2, l3, le = label("p1"), label("p2"), label("p3"), label("pe")
        o += ["s_cmp_eq_u32 %[q], 1", f"s_cbranch_scc1 {l1}", "s_cmp_eq_u32 %[q], 2", f"s_cbranch_scc1 {l2}", "s_cmp_eq_u32 %[q], 3",
              f"s_cbranch_scc1 {l3}", f"s_branch {le}", f"{l1}:", "s_setprio 1", f"s_branch {le}", f"{l2}:", "s_setprio 2", f"s_branch {le}",
              f"{l3}:", "s_setprio 3", f"{le}:"]
    o += [f"s_mov_b32 s{S_MASK}, 0xffffffe0", f"s_sub_u32 s{S_MM1}, {S_M}, 1", f"s_mov_b32 s{S_UND}, 0",
         # resa = L_RES + (2 wave + lane / 32) * 8 from rsa = L_RS + (2 wave + lane / 32) * 4
         f"s_add_u32 s64, {S_LDSB}, {L_RS}", f"v_subrev_u32_e32 v{V_RESA}, s64, v{V_RSA}", f"v_lshlrev_b32_e32 v{V_RESA}, 1, v{V_RESA}",
         f"s_add_u32 s64, {S_LDSB}, {L_RES}", f"v_add_u32_e32 v{V_RESA}, s64, v{V_RESA}"]
    # ---- rows of tiles 0 and 1 first (HBM latency), then the wave's 16 codebook fragments (L2)
    if not (abl & 1):
        o += [i["t"] for i in loads(0, 0) + loads(1, 1)]
    else:
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
    o += ["s_waitcnt vmcnt(0)"]
    # ---- conversions of tiles 0 and 1, loads of tiles 2 and 3
    for tile in (0, 1):
        o += schedule(convert(tile, tile, on=not (abl & 4), wait=None) + loads(tile + 2, tile, on=not (abl & 1)), [], [])
    # ---- (prologue done: a 100 MHz stamp per wave into L_DBG, read by the diagnostics build only)
    o += ["s_memrealtime s[64:65]", f"s_lshl_b32 s66, {S_WAVE}, 3", f"s_add_u32 s66, s66, {S_LDSB}", f"s_add_u32 s66, s66, {L_DBG}",
          "s_mov_b32 exec_lo, 1", "s_mov_b32 exec_hi, 0", "s_waitcnt lgkmcnt(0)",
          f"v_mov_b32_e32 v{W[0]}, s64", f"v_mov_b32_e32 v{W[1]}, s65", f"v_mov_b32_e32 v{W[2]}, s66",
          f"ds_write_b64 v{W[2]}, v[{W[0]}:{W[1]}]", "s_mov_b64 exec, -1"]
    # ---- the eight periods; behind period t the block leaves when the workgroup has no tile t + 1
    done = label("done")
    exits = []
    for t in range(MAX_TILES):
        o += period(t, abl, var)
        if t + 1 < MAX_TILES:
            ex = label(f"exit{t}")
            o += [f"s_cmp_le_i32 {S_NTL}, {t + 1}", f"s_cbranch_scc1 {ex}"]
            exits += [f"{ex}:"] + leave(t, abl, var) + [f"s_branch {done}"]
    o += leave(MAX_TILES - 1, abl, var) + [f"s_branch {done}"] + exits
    o += [f"{done}:",
          "s_waitcnt vmcnt(0)",                                           # rows behind the last tile (never used) have landed
          # undecided-row count of this wave -> L_CNT[4 + wave]
          f"s_lshl_b32 s64, {S_WAVE}, 2", f"s_add_u32 s64, s64, {S_LDSB}", f"s_add_u32 s64, s64, {L_CNT + 16}",
          "s_mov_b32 exec_lo, 1", "s_mov_b32 exec_hi, 0",
          f"v_mov_b32_e32 v{W[0]}, s64", f"v_mov_b32_e32 v{W[1]}, s{S_UND}", f"ds_write_b32 v{W[0]}, v{W[1]}",
          "s_mov_b64 exec, -1",
          "s_waitcnt lgkmcnt(0)"]
    if var & 4:
        o.append("s_setprio 0")
    return o


def lit(lines):
    return " \\\n    ".join('"' + l + '\\n\\t"' for l in lines)


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "d-vqvae_amd", "csrc", "vq_pipe_loop.h")
    with open(out, "w") as f:
        f.write("// GENERATED by tools/gen_vq_pipe.py -- prologue and tile periods of vq_pipe.hip as one instruction block; do not edit\n"
                "#pragma once\n")
        for k, v in (("Z16_ROW", Z16_ROW), ("Z16_BUF", Z16_BUF), ("NZB", NZB), ("MS_ROW", MS_ROW), ("MS_BUF", MS_BUF),
                     ("PAIR_CAP", PAIR_CAP), ("L_Z16", L_Z16), ("L_MS", L_MS), ("L_RS", L_RS), ("L_EES", L_EES), ("L_RES", L_RES),
                     ("L_REC", L_REC), ("L_PAIR", L_PAIR), ("L_SLOW", L_SLOW), ("L_UND", L_UND), ("UND_PER_WAVE", UND_PER_WAVE),
                     ("L_CNT", L_CNT), ("L_DBG", L_DBG), ("LDS_BYTES", LDS_BYTES), ("FD", FD)):
            f.write(f"#define VQP_{k} {v}\n")
        f.write(f"#define VQP_DEFAULT_VAR {DEFAULT_VAR}\n")
        f.write(f"#define VQP_ASM_0_{DEFAULT_VAR} \\\n    " + lit(program(var=DEFAULT_VAR)) + "\n")
        combos = [(0, DEFAULT_VAR)]
        f.write("#ifdef DVQ_DIAG\n")
        for var in (0, 1, 5, 16, 17, 18, 19):
            for stamps in (0, 8):
                if (0, var | stamps) in combos:
                    continue
                combos.append((0, var | stamps))
                f.write(f"#define VQP_ASM_0_{var | stamps} \\\n    " + lit(program(var=var, stamps=bool(stamps))) + "\n")
        for abl in (1, 2, 4, 8, 32, 96, 15, 256, 512):
            combos.append((abl, DEFAULT_VAR))
            f.write(f"#define VQP_ASM_{abl}_{DEFAULT_VAR} \\\n    " + lit(program(abl=abl, var=DEFAULT_VAR)) + "\n")
        f.write("#define VQP_VARIANTS(X) " + " ".join(f"X({a}, {v})" for a, v in combos) + "\n")
        f.write("#else\n")
        f.write(f"#define VQP_VARIANTS(X) X(0, {DEFAULT_VAR})\n")
        f.write("#endif\n")
        clob = ([f'"v{i}"' for i in list(range(0, 48)) + [V_RESA, V_THR, V_VAL] + list(range(64, 128))]
                + [f'"s{i}"' for i in range(S_FIRST, S_LAST + 1)])
        f.write("#define VQP_ASM_CLOBBERS " + ", ".join(clob) + ', "vcc", "scc", "memory"\n')
    prog = program(var=DEFAULT_VAR)
    per = [i for i, l in enumerate(prog) if l == "s_barrier"]
    print("wrote", os.path.normpath(out), "- instructions + labels in the default block:", len(prog),
          "; period 3:", per[4] - per[3] if len(per) > 4 else "?")
    if "--list" in sys.argv:
        print("\n".join(prog))


if __name__ == "__main__":
    main()
