// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256), second structure (DVQ_VQ_KERNEL=16): SIXTEEN waves per
// workgroup, four per SIMD, each holding 32 codebook entries; plain straight-line code, no generated gap tables.  Same result
// as vq_stream.hip / vq.hip / oracle/vq_canonical.c, bit for bit.  Reference: VectorQuantizer.forward(z, istrain=False),
// network/vqvae/quantizer.py:46-49.
//
// Why it exists (round 3; every number: MI355X, M = 65 536, six rotating inputs, tools/vq_kernel_ab.py, tools/vq16_phase_stamps.py,
// tools/vq_pmc.sh).  Three structures were built to get the 36 us of vq_stream.hip (eight waves, two per SIMD) to the 21.3 us the
// task asks for:
//   (a) FOUR waves, one per SIMD, the codebook as the MFMA A operand in a[0:255] (inline asm; microbenchmark
//       tools/microbench/mfma_agpr_operand.hip: 32.1 cycles per dependent MFMA, 4-5 placed vector instructions per gap free),
//       64-gap generated body: 37-43 us.  A single in-order wave per SIMD pays the dependent-issue latency of every vector
//       chain (SQ counters: 29 % of the wave cycles issuing vector instructions, 44 % waiting, MFMA busy 23 %); compile-time
//       ablations were ADDITIVE (no MFMA -8.5 us of 22.3, no scoring -2.4, no merge decisions -3.7, no barrier -1.3, no row
//       loads 0): nothing overlapped.  Removed from the tree (history: "VQ: four-wave streaming kernel").
//   (b) THIS kernel: 36.4 us = vq_stream.hip.  Ablations of its loop (21-24 us): no conversion -7.4, no MFMA -9.9, no merge -2.7,
//       no scoring -2.5, no barrier -2.1, no row loads -2.0: additive again although a matrix-only wave and a vector-only wave on one
//       SIMD do run concurrently (tools/microbench/mfma_valu_two_waves.hip: 520 vs 512 cycles per 16 MFMAs, vector chain +9 %).
//       SQ counters: 8.2 M vector instructions per launch at 4.0 cycles of SIMD issue each = 15 us of pure vector issue per SIMD;
//       dealing the vector work by role (below) took 4 000 -> 2 200 instructions per tile and CU and the loop 24.4 -> 21.3 us.
//   (c) 64-row tiles (half the barriers): 24.4 us, 17 spilled registers: not kept.
// What the three have in common is ~6 000 cycles per 32-row tile and CU against 2 048 of matrix work and ~2 200 of vector issue:
// the per-tile dependency chain (barrier -> fragment reads -> 16 dependent MFMAs -> 16-step top-2 chain -> slot -> barrier, and
// beside it load -> convert -> row reduction -> LDS) is executed by too few independent instruction streams to fill either pipe.
//
//   * wave w keeps entries [32w, 32w+32) -- fp16 of -2 sE e_k, all 256 dims -- as the MFMA A operand in 64 VGPRs (<= 128 VGPRs:
//     four waves per SIMD); the accumulator start values sE |e_k|^2 come from the LDS each tile;
//   * per tile: one s_barrier (the fp16 image of tile t and the slots of tile t-1 are complete), 16 fragment reads + 16 MFMAs,
//     scores -> the wave's slot (min, second) per row; waves 0-7 convert 4 rows of tile t+1 each (16 lanes per row, loaded a
//     tile earlier) and load tile t+2; waves 8-11 merge 8 rows of tile t-1 each (8 lanes per row, 4 slots per lane); waves 4-11
//     do that vector work BEFORE their MFMAs, the others after (the four waves of a SIMD leave the barrier together);
//   * decided rows write their entry; every row leaves a record {threshold, flags}; undecided rows go to the merging wave's own
//     list (no atomics, no branch in the loop) and are expanded into candidate pairs after the loop (all slots stay in the LDS:
//     9 x 8.7 KB), then the canonical fp32 refine of vq_stream.hip (see its header for the error bound), eight lanes per chain.
#include "dvq_internal.h"
#include "vq_pack.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int K = VQ_K, D = VQ_D;
constexpr int NWV = 16;                            // waves per workgroup = 32-entry slices of the codebook, four per SIMD
constexpr int NT = 64 * NWV;                       // 1024 threads, one workgroup per CU
constexpr int TILE = 32;
constexpr int MAX_TILES = 8;

constexpr int Z16_ROW = 528;                       // padded fp16 row: 512 B + 16 B
constexpr int Z16_BUF = TILE * Z16_ROW;
constexpr int MS_ROW = 272;                        // merge slots of a row: 32 (wave, lane half) x 8 B + 16 B pad
constexpr int MS_BUF = TILE * MS_ROW;
constexpr int PAIR_CAP = 2048;

constexpr int L_Z16 = 0;                                   // 2 x fp16 tile
constexpr int L_MS = L_Z16 + 2 * Z16_BUF;                  // (MAX_TILES + 1) x merge slots: local tiles -1 .. 7, kept until the tail
constexpr int L_RS = L_MS + (MAX_TILES + 1) * MS_BUF;      // 4 x [32] {eps sE, flag}
constexpr int L_EES = L_RS + 4 * TILE * 8;                 // [K] f32: sE |e_k|^2 (accumulator start values)
constexpr int L_RES = L_EES + K * 4;                       // [MAX_TILES*32] u64 row results (ordered distance bits : entry)
constexpr int L_PAIR = L_RES + MAX_TILES * TILE * 8;       // [PAIR_CAP] u32 (rowslot << 16 | entry)
constexpr int L_SLOW = L_PAIR + PAIR_CAP * 4;              // [MAX_TILES*32] u16 rowslots for the all-entries path
constexpr int L_CNT = L_SLOW + MAX_TILES * TILE * 2;       // [0] pairs, [1] rows for the all-entries path, [2] rows with >= 32 pairs,
                                                           // [4 + m] undecided rows of merging wave m
constexpr int L_REC = L_CNT + 128;                         // [MAX_TILES*32] {threshold, flags} of every row (merge -> tail)
constexpr int L_UND = L_REC + MAX_TILES * TILE * 8;        // [4 merging waves][64] u16 rowslots of the rows the merge could not decide
constexpr int LDS_BYTES = L_UND + 4 * 64 * 2;
static_assert(LDS_BYTES <= 160 * 1024 && L_MS % 16 == 0 && L_RS % 16 == 0 && L_RES % 8 == 0 && L_REC % 8 == 0, "LDS layout");

// ------------------------------------------------------------------------------------------------ LDS access by hand
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF, class V>
__device__ __forceinline__ void ds_rd128(V& d, unsigned a) {
    static_assert(sizeof(V) == 16, "16-byte destination");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
template <int OFF, class V>
__device__ __forceinline__ void ds_rd64(V& d, unsigned a) {
    static_assert(sizeof(V) == 8, "8-byte destination");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
template <int OFF, class V>
__device__ __forceinline__ void ds_wr64(unsigned a, const V& v) {
    static_assert(sizeof(V) == 8, "8-byte source");
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void ds_wr32(unsigned a, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void ds_wr16(unsigned a, unsigned v) { asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ds_add_rtn(unsigned a, unsigned v) {
    unsigned r;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(v) : "memory");
    return r;
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float row16_sum(float v) {           // all-reduce over the 16 lanes of a DPP row
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}
// exchange with lane ^ 16 (the other 16-lane DPP row of the same 32-lane half): ds_swizzle, bit mode xor 0x10
__device__ __forceinline__ float swz16(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F)); }
__device__ __forceinline__ float row32_sum(float v) { v = row16_sum(v); return v + swz16(v); }
__device__ __forceinline__ float min_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -3.0e38f); }

// entry of accumulator register i (0..15), lane half h, wave w (v_mfma_f32_32x32x16 D layout)
__device__ __forceinline__ int entry_of(int w, int i, int h) { return 32 * w + 8 * (i >> 2) + 4 * h + (i & 3); }

__device__ __forceinline__ unsigned long long order_key(float d, int k) {
    const unsigned b = __float_as_uint(d);
    const unsigned u = (d != d) ? 0u : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    return ((unsigned long long)u << 32) | (unsigned)k;
}

// ------------------------------------------------------------------------------------------------ refine chains (as vq_stream.hip)
// Canonical chains threaded through 8 lanes: lane q of a group holds floats [32q, 32q+32) of its z row and of its candidate's
// codebook row (all loads issued up front: ONE memory latency; 64 registers: this kernel runs at 128), then the k-ordered fmaf
// chain runs as eight 32-step rounds, round q continuing from the accumulator lane q-1 produced.  Bit-identical to one 256-step chain.
__device__ __forceinline__ void chain_pair_x8(const float* __restrict__ zr, const float* __restrict__ er, int q, bool active,
                                              float& zz, float& dot) {
    f32x4 x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {                               // idle lanes load nothing
        x[u] = active ? *reinterpret_cast<const f32x4*>(zr + 32 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
        y[u] = active ? *reinterpret_cast<const f32x4*>(er + 32 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // The accumulator travels lane q - 1 -> lane q by DPP (row_shr:1: one vector instruction; the ds_bpermute broadcasts this replaced
    // were sixteen LDS round trips per pair, about a microsecond of the kernel's tail).  Lane q's value counts in round q only, so
    // what lane 0 of a group receives from the neighbouring group is never used.  The result is LANE 7's (q == 7): the caller reads it there.
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int round = 0; round < 8; ++round) {
        const float a_in = round ? dpp_f<0x111>(a) : 0.f;
        const float b_in = round ? dpp_f<0x111>(b) : 0.f;
        float ta = a_in, tb = b_in;
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                ta = fmaf(x[u][cc], x[u][cc], ta);
                tb = fmaf(x[u][cc], y[u][cc], tb);
            }
        if (q == round) { a = ta; b = tb; }
    }
    zz = a;                                                     // valid in lane q == 7
    dot = b;
}
__device__ __forceinline__ void chain_pair(const float* __restrict__ zr, const float* __restrict__ er, float& zz, float& dot) {
    float a = 0.f, b = 0.f;
    for (int j0 = 0; j0 < D; j0 += 32) {
        f32x4 x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = *reinterpret_cast<const f32x4*>(zr + j0 + 4 * u);
            y[u] = *reinterpret_cast<const f32x4*>(er + j0 + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a = fmaf(x[u][c], x[u][c], a);
                b = fmaf(x[u][c], y[u][c], b);
            }
    }
    zz = a;
    dot = b;
}

// ------------------------------------------------------------------------------------------------ the kernel
#ifndef DVQ_MEASURE_DZ
#define DVQ_MEASURE_DZ 1   // 1: |z - h(z)| measured element by element; 0: half-ulp bound (fewer vector instructions, ~1.5x the pairs)
#endif
template <int HI>
__device__ __forceinline__ float mix_diff(float hp, float x) {           // h - x with h one fp16 half of hp: exact (one v_fma_mix_f32)
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    return d;
}
__device__ __forceinline__ unsigned ms_buf(int t) { return (unsigned)((t + 1) * MS_BUF); }
__device__ __forceinline__ unsigned rs_tab(int t) { return (unsigned)((t & 3) * (TILE * 8)); }

struct Ctx {
    const float* z;
    long M, tile0;
    int G, ntl, wave, lane;
    unsigned lds0;
    float emax, demax, sEf;
    bool e_valid;
};
// rows of local tile j that exist (wave-uniform): 32 inside the data, fewer in the last tile, <= 0 for j < 0
__device__ __forceinline__ int rows_of(const Ctx& c, int j) {
    if (j < 0) return 0;
    const long left = c.M - (c.tile0 + (long)blockIdx.x + (long)j * c.G) * TILE;
    return left >= TILE ? TILE : (int)(left > 0 ? left : 0);
}
// Vector work is dealt to the waves by role so that no per-row overhead is replicated more than necessary (the loop is bound by
// the SIMDs' vector issue: ~4 cycles per wave instruction whatever the number of waves): waves 0-7 convert 4 rows of the next
// tile each (16 lanes per row), waves 8-11 merge 8 rows of the previous tile each (8 lanes per row, 4 slots per lane), waves
// 12-15 only multiply and score: every SIMD (waves w, w+4, w+8, w+12) has two converters, one merger and one plain wave.
constexpr int N_CONV = 8, FIRST_MERGE = 8, N_MERGE = 4;

// rows 4 cw + g (g = lane / 16) of local tile j, HBM -> registers: lane i = lane % 16 takes floats 4 (i + 16 q) .. +3, q = 0..3
// (per instruction four 256-byte segments); rows behind the end of the data repeat the last row
__device__ __forceinline__ void load_rows(const Ctx& c, int j, f32x4 (&x)[4]) {
    long gr = (c.tile0 + (long)blockIdx.x + (long)j * c.G) * TILE + c.wave * 4 + (c.lane >> 4);
    if (gr >= c.M) gr = c.M - 1;
    const f32x4* p = reinterpret_cast<const f32x4*>(c.z + gr * D) + (c.lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = __builtin_nontemporal_load(p + 16 * q);
}
// fp32 -> fp16 of the wave's four rows of tile j: image, |h(z)|^2, measured rounding error, eps_row
__device__ __forceinline__ void convert(const Ctx& c, int j, const f32x4 (&x)[4]) {
    const int g = c.lane >> 4, i = c.lane & 15, row = 4 * c.wave + g;
    const unsigned za = c.lds0 + L_Z16 + (unsigned)((j & 1) * Z16_BUF) + row * Z16_ROW + 8 * i;
    float hh = 0.f, dsq = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2 a, b, pk;
        a[0] = x[q][0]; a[1] = x[q][1]; b[0] = x[q][2]; b[1] = x[q][3];
        const f16x2 lo = __builtin_convertvector(a, f16x2), hi = __builtin_convertvector(b, f16x2);
        hh = __builtin_amdgcn_fdot2(lo, lo, hh, false);
        hh = __builtin_amdgcn_fdot2(hi, hi, hh, false);
        pk[0] = __builtin_bit_cast(float, lo);
        pk[1] = __builtin_bit_cast(float, hi);
        if (q == 0) ds_wr64<0>(za, pk);
        else if (q == 1) ds_wr64<128>(za, pk);
        else if (q == 2) ds_wr64<256>(za, pk);
        else ds_wr64<384>(za, pk);
        if (DVQ_MEASURE_DZ) {
            const float d0 = mix_diff<0>(pk[0], x[q][0]), d1 = mix_diff<1>(pk[0], x[q][1]);
            const float d2 = mix_diff<0>(pk[1], x[q][2]), d3 = mix_diff<1>(pk[1], x[q][3]);
            dsq = fmaf(d0, d0, dsq); dsq = fmaf(d1, d1, dsq); dsq = fmaf(d2, d2, dsq); dsq = fmaf(d3, d3, dsq);
        }
    }
    hh = row16_sum(hh);
    if (DVQ_MEASURE_DZ) dsq = row16_sum(dsq);
    const float hn = __builtin_amdgcn_sqrtf(hh);
    // |z - h(z)|: measured, or a priori (half an ulp of a normal fp16 is at most 2^-11 |h|, of a subnormal one 2^-25)
    const float dzn = DVQ_MEASURE_DZ ? __builtin_amdgcn_sqrtf(dsq) * 1.0001f : hn * 4.8877e-4f + 4.8e-7f;
    const float zn = (hn + dzn) * 1.0001f;                              // |z| <= |h(z)| + |z - h(z)|
    const float u = zn + c.emax;
    const float eps = 4.004f * (dzn * c.emax + zn * c.demax + dzn * c.demax) + 1.01e-4f * u * u;
    const float epsS = eps * c.sEf;
    const bool bad = !c.e_valid || !(hh <= 3.0e38f) || !(dsq <= 3.0e38f) || !(epsS <= 3.0e38f);   // NaN/Inf, fp16 overflow
    if (i == 0) {
        f32x2 rs;
        rs[0] = epsS;
        rs[1] = __uint_as_float(bad ? 1u : 0u);
        ds_wr64<0>(c.lds0 + L_RS + rs_tab(j) + row * 8, rs);
    }
}

__device__ __forceinline__ int row8_sum(int v) {                // all-reduce over 8 consecutive lanes
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    return v;
}
// merge of rows 8 mw + g (g = lane / 8; mw = wave - FIRST_MERGE) of tile tm: 8 lanes per row, lane i looks at slots 4i .. 4i+3,
// slot s = (source wave s / 2, lane half s % 2); `rows` = rows of tile tm that exist; `und` counts the rows this wave left
// undecided (wave-uniform).  `s0`, `s1`, thr are reused by expand() in the tail.
struct Merge {
    f32x4 s0, s1;                                                       // (m1, m2) of slots 4i, 4i+1 | 4i+2, 4i+3
    float thr;
    int c1, c2, n1, n2;
    __device__ __forceinline__ void counts() {
        const bool a0 = s0[0] <= thr, b0 = s0[1] <= thr, a1 = s0[2] <= thr, b1 = s0[3] <= thr;
        const bool a2 = s1[0] <= thr, b2 = s1[1] <= thr, a3 = s1[2] <= thr, b3 = s1[3] <= thr;
        c1 = (int)a0 + (int)a1 + (int)a2 + (int)a3;
        c2 = (int)b0 + (int)b1 + (int)b2 + (int)b3;
        n1 = row8_sum(c1);
        n2 = row8_sum(c2);
    }
    __device__ __forceinline__ void run(const Ctx& c, int tm, int rows, int& und) {
        const int g = c.lane >> 3, i = c.lane & 7, r = 8 * (c.wave - FIRST_MERGE) + g;
        f32x2 rs;
        const unsigned a = c.lds0 + L_MS + ms_buf(tm) + r * MS_ROW + i * 32;
        ds_rd128<0>(s0, a);
        ds_rd128<16>(s1, a);
        ds_rd64<0>(rs, c.lds0 + L_RS + rs_tab(tm) + r * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(rs));
        thr = min_nc(min_nc(s0[0], s0[2]), min_nc(s1[0], s1[2]));
        thr = min_nc(thr, dpp_f<0xB1>(thr));
        thr = min_nc(thr, dpp_f<0x4E>(thr));
        thr = min_nc(thr, dpp_f<0x141>(thr));
        thr = thr + rs[0];
        counts();
        const bool bad = __float_as_uint(rs[1]) != 0u;
        const bool live = r < rows;
        const bool slow = bad || n1 == 0;
        const bool unique = !slow && n1 == 1 && n2 == 0;
        const bool amb = live && !slow && !unique;
        const unsigned flags = (live ? 1u : 0u) | (slow ? 2u : 0u) | (unique ? 4u : 0u) | (amb ? 8u : 0u);
        const unsigned rowslot = (unsigned)(tm * TILE + r);
        if (live && unique && c1) {                                     // the winner lane of a decided row writes the entry
            const bool a0 = s0[0] <= thr, a1 = s0[2] <= thr, a2 = s1[0] <= thr;
            const int j = a0 ? 0 : (a1 ? 1 : (a2 ? 2 : 3));
            const float best = a0 ? s0[0] : (a1 ? s0[2] : (a2 ? s1[0] : s1[2]));
            const int sidx = 4 * i + j;
            f32x2 kv;
            kv[0] = __uint_as_float((unsigned)entry_of(sidx >> 1, (int)(__float_as_uint(best) & 15u), sidx & 1));
            kv[1] = __uint_as_float(0u);
            ds_wr64<0>(c.lds0 + L_RES + rowslot * 8, kv);
        }
        const bool lead = i == 0;
        if (lead && live) {                                             // every live row: its threshold and flags, for the tail
            f32x2 rec;
            rec[0] = thr;
            rec[1] = __uint_as_float(flags);
            ds_wr64<0>(c.lds0 + L_REC + rowslot * 8, rec);
        }
        const bool put = lead && live && !unique;                       // undecided: ambiguous or all-entries
        const unsigned long long m = __ballot(put);
        const int pos = und + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (put && pos < 64) ds_wr16(c.lds0 + L_UND + ((c.wave - FIRST_MERGE) * 64 + pos) * 2, rowslot);
        und += __popcll(m);
    }
    // tail: the undecided row `rowslot` (8 lanes per row, as in the merge) -> candidate pairs / the all-entries list
    __device__ __forceinline__ void expand(const Ctx& c, unsigned rowslot, bool valid) {
        const int i = c.lane & 7;
        const int tm = (int)(rowslot >> 5), r = (int)(rowslot & 31);
        const unsigned a = c.lds0 + L_MS + ms_buf(tm) + r * MS_ROW + i * 32;
        f32x2 rec;
        ds_rd128<0>(s0, a);
        ds_rd128<16>(s1, a);
        ds_rd64<0>(rec, c.lds0 + L_REC + rowslot * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(rec));
        thr = rec[0];
        const unsigned flags = valid ? __float_as_uint(rec[1]) : 0u;
        const bool live = flags & 1u, slow = flags & 2u, amb = flags & 8u;
        counts();
        const bool a0 = s0[0] <= thr, b0 = s0[1] <= thr, a1 = s0[2] <= thr, b1 = s0[3] <= thr;
        const bool a2 = s1[0] <= thr, b2 = s1[1] <= thr, a3 = s1[2] <= thr, b3 = s1[3] <= thr;
        // one reservation per ambiguous row (leader lane i == 0): a slot with one score within eps takes one pair, a slot whose
        // second score is within eps too (it may hide a third) takes all its 16 entries
        const int need = (n1 - n2) + 16 * n2;
        unsigned pos = 0;
        if (amb && i == 0) pos = ds_add_rtn(c.lds0 + L_CNT, (unsigned)need);
        pos = (unsigned)__shfl((int)pos, c.lane & ~7);
        const bool fits = pos + (unsigned)need <= (unsigned)PAIR_CAP;
        if (amb && i == 0 && fits && need >= 32) asm volatile("ds_add_u32 %0, %1" ::"v"(c.lds0 + L_CNT + 8), "v"(1u) : "memory");
        const int mine = (c1 - c2) + 16 * c2;                            // exclusive prefix of the lanes' pair counts within the row
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (i >= d) incl += up;
        }
        if (amb && fits && c1) {
            unsigned off = pos + (unsigned)(incl - mine);
            auto push = [&](int j, bool one, bool all, float best) {
                const int sidx = 4 * i + j;
                if (all) {
                    for (int e = 0; e < 16; ++e) ds_wr32(c.lds0 + L_PAIR + (off + e) * 4, (rowslot << 16) | (unsigned)entry_of(sidx >> 1, e, sidx & 1));
                    off += 16;
                } else if (one) {
                    ds_wr32(c.lds0 + L_PAIR + off * 4, (rowslot << 16) | (unsigned)entry_of(sidx >> 1, (int)(__float_as_uint(best) & 15u), sidx & 1));
                    off += 1;
                }
            };
            push(0, a0, b0, s0[0]);
            push(1, a1, b1, s0[2]);
            push(2, a2, b2, s1[0]);
            push(3, a3, b3, s1[2]);
        }
        if (live && i == 0 && (slow || (amb && !fits))) {
            const unsigned sp = ds_add_rtn(c.lds0 + L_CNT + 4, 1u);
            ds_wr16(c.lds0 + L_SLOW + sp * 2, rowslot);
        }
    }
};

template <int ABL>    // diagnostics build only (DVQ_VQ16_ABL; results INVALID unless 0): 1 no row loads, 2 no merge, 4 no conversion, 8 no scoring,
                      // 16 no barrier, 32 no MFMA, 64 no fragment reads
__global__ __launch_bounds__(NT) void vq_stream16_kernel(const float* __restrict__ z, const float* __restrict__ E, long M, long tile0,
                                                         long n_tiles, const char* __restrict__ packed, int64_t* __restrict__ idx,
                                                         unsigned long long* __restrict__ slow_rows, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
#if DVQ_DIAG_ON
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
#define DVQ_T(VAR) const unsigned long long VAR = __builtin_amdgcn_s_memrealtime()
#else
#define DVQ_T(VAR)
#endif
    Ctx c;
    c.z = z;
    c.M = M;
    c.tile0 = tile0;
    c.G = (int)gridDim.x;
    c.ntl = (int)((n_tiles - (long)blockIdx.x + c.G - 1) / c.G);
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.lds0 = lds_addr(lds);
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    __builtin_amdgcn_s_setreg((2 - 1) << 11 | 6 << 6 | 1, 3);            // MODE.FP_DENORM[3:2] = 3: fp16 subnormals kept

    // ---- prologue: the first rows (needed first), the wave's 16 codebook fragments, its accumulator start values
    const bool conv = c.wave < N_CONV, mrg = c.wave >= FIRST_MERGE && c.wave < FIRST_MERGE + N_MERGE;
    f32x4 x[4];
    if (conv) load_rows(c, 0, x);
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    c.emax = hdr->emax;
    c.demax = hdr->demax;
    c.e_valid = hdr->valid != 0;
    c.sEf = c.e_valid ? pow2f(hdr->sexp) : 1.0f;
    const int r_l = c.lane & 31, h_l = c.lane >> 5;
    f16x8 af[16];
    {
        const f16x8* img = reinterpret_cast<const f16x8*>(packed + PK_OFF_IMG) + (size_t)c.wave * (16 * 64) + c.lane;
#pragma unroll
        for (int s = 0; s < 16; ++s) af[s] = img[s * 64];
    }
    const int Kr = hdr->K;                                 // real entries (a multiple of 32); the rest of the image is padding (vq_pack_norm_kernel)
    if (tid < K) reinterpret_cast<float*>(lds + L_EES)[tid] = tid < Kr ? ee_g[tid] * c.sEf : 3.0e38f;   // accumulator start values (read back per
                                                                                       // tile: 16 resident registers would spill fragments)
    for (int u = tid; u < MAX_TILES * TILE; u += NT) reinterpret_cast<unsigned long long*>(lds + L_RES)[u] = ~0ull;
    for (int u = tid; u < PAIR_CAP; u += NT) reinterpret_cast<unsigned*>(lds + L_PAIR)[u] = ~0u;
    if (tid < 32) reinterpret_cast<unsigned*>(lds + L_CNT)[tid] = 0u;
    if (conv) {
        convert(c, 0, x);
        if (c.ntl > 1) load_rows(c, 1, x);
    }
    DVQ_T(t_pro);

    // ---- tile loop
    const unsigned zbase = c.lds0 + L_Z16 + r_l * Z16_ROW + 16 * h_l;
    const unsigned slotw = c.lds0 + L_MS + r_l * MS_ROW + (2 * c.wave + h_l) * 8;
    int und = 0;
    Merge mg;
    // The four waves of a SIMD (w, w+4, w+8, w+12) leave the barrier together; if all ran the same program they would all want
    // the matrix pipe first and the vector port afterwards.  Waves 4-7 (converters) and 8-11 (mergers) therefore do their vector
    // work (conversion of tile t+1, merge of tile t-1: independent of tile t's products) BEFORE their MFMAs, the others after.
    const bool late = c.wave >= 4 && c.wave < 12;
    auto products = [&](int t) __attribute__((always_inline)) {
        const char* zt = lds + (zbase - c.lds0) + (t & 1) * Z16_BUF;
        f32x16 acc;
        {
            const f32x4* ci = reinterpret_cast<const f32x4*>(lds + L_EES + (32 * c.wave + 4 * h_l) * 4);   // registers 4q..4q+3 <-> entries 8q + 4h ..
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = ci[2 * q];
                acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
            }
        }
        const f16x8 bfix = af[3];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f16x8 bfr = (ABL & 64) ? bfix : *reinterpret_cast<const f16x8*>(zt + 32 * s);
            if (!(ABL & 32)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], bfr, acc, 0, 0, 0);
            else if (!(ABL & 64)) asm volatile("" :: "v"(bfr));
        }
        float m1 = INFINITY, m2 = INFINITY;                              // (min, second) of the lane's 16 scores, id in the low 5 mantissa bits
#pragma unroll
        for (int e = (ABL & 8) ? 15 : 0; e < 16; ++e) {
            const float p = __uint_as_float((__float_as_uint(acc[e]) & ~31u) | (unsigned)e);
            m2 = __builtin_amdgcn_fmed3f(m1, m2, p);
            m1 = min_nc(m1, p);
        }
        f32x2 v;
        v[0] = m1;
        v[1] = m2;
        ds_wr64<0>(slotw + ms_buf(t), v);
    };
    for (int t = 0; t < c.ntl; ++t) {
        if (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else wg_barrier();                                              // the fp16 image of tile t and the slots of tile t-1 are complete
        if (!late) products(t);
        if (conv) {
            if (!(ABL & 4)) convert(c, t + 1, x);                       // (behind the last tile: stale registers, never used)
            if (t + 2 < c.ntl && !(ABL & 1)) load_rows(c, t + 2, x);
        }
        if (mrg && !(ABL & 2)) mg.run(c, t - 1, rows_of(c, t - 1), und);
        if (late) products(t);
    }
    wg_barrier();
    DVQ_T(t_loop);
    if (mrg) {
        mg.run(c, c.ntl - 1, rows_of(c, c.ntl - 1), und);
        if (c.lane == 0) reinterpret_cast<unsigned*>(lds + L_CNT)[4 + c.wave - FIRST_MERGE] = (unsigned)und;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- the undecided rows of the four lists -> candidate pairs / the all-entries list: 8 lanes per row, 128 rows per pass
    {
        const unsigned* cnt4 = reinterpret_cast<const unsigned*>(lds + L_CNT) + 4;
        const int n0 = (int)min(cnt4[0], 64u), n1_ = (int)min(cnt4[1], 64u), n2_ = (int)min(cnt4[2], 64u), n3_ = (int)min(cnt4[3], 64u);
        const int n_und = n0 + n1_ + n2_ + n3_;
        const uint16_t* ul = reinterpret_cast<const uint16_t*>(lds + L_UND);
        for (int base = 0; base < n_und; base += NT / 8) {
            if (base + c.wave * 8 < n_und) {                            // wave-uniform: this wave has at least one row
                const int k = base + (tid >> 3);
                const bool valid = k < n_und;
                int w = 0, kk = valid ? k : 0;
                if (kk >= n0) { kk -= n0; w = 1; if (kk >= n1_) { kk -= n1_; w = 2; if (kk >= n2_) { kk -= n2_; w = 3; } } }
                mg.expand(c, (unsigned)ul[w * 64 + kk], valid);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    DVQ_T(t_merge);

    // ---- refine: canonical distances of the listed pairs, eight lanes per chain; key minimum per row
    unsigned long long* s_res = reinterpret_cast<unsigned long long*>(lds + L_RES);
    const unsigned* s_pair = reinterpret_cast<const unsigned*>(lds + L_PAIR);
    const uint16_t* s_slow = reinterpret_cast<const uint16_t*>(lds + L_SLOW);
    const unsigned* s_cnt = reinterpret_cast<const unsigned*>(lds + L_CNT);
    const int total = (int)min(s_cnt[0], (unsigned)PAIR_CAP);
    const int n_slow = (int)s_cnt[1];
    auto grow_of = [&](int rowslot) { return (c.tile0 + (long)blockIdx.x + (long)(rowslot >> 5) * c.G) * TILE + (rowslot & 31); };
    for (int s0 = 0; s0 < total; s0 += NT / 8) {
        if (s0 + c.wave * 8 < total) {                            // wave-uniform: this wave has at least one pair
            const int slot = s0 + (tid >> 3), q = tid & 7;
            const unsigned pr = slot < total ? s_pair[slot] : ~0u;
            const bool act = pr != ~0u;
            const int rowslot = (int)(pr >> 16), k = (int)(pr & 0xffffu);
            float zz, dot;
            chain_pair_x8(z + grow_of(rowslot) * D, E + (long)k * D, q, act, zz, dot);
            if (act && q == 7) {                                  // the chain ends in the group's last lane
                const float tsum = zz + ee_g[k];
                atomicMin(&s_res[rowslot], order_key(tsum - 2.0f * dot, k));
            }
        }
    }
    // what is left (NaN/Inf, fp16 overflow, invalid codebook image, overflowing list): all K entries canonically
    for (int o = 0; o < n_slow; ++o) {
        const int rowslot = s_slow[o];
        if (tid < Kr) {
            float zz2, dot2;
            chain_pair(z + grow_of(rowslot) * D, E + (long)tid * D, zz2, dot2);
            const float tsum = zz2 + ee_g[tid];
            atomicMin(&s_res[rowslot], order_key(tsum - 2.0f * dot2, tid));
        }
    }
    if (slow_rows && tid == 0 && n_slow + (int)s_cnt[2] > 0) atomicAdd(slow_rows, (unsigned long long)(n_slow + (int)s_cnt[2]));
    __syncthreads();
    if (tid < c.ntl * TILE) {
        const long gr = grow_of(tid);
        if (gr < M) idx[gr] = (int64_t)(unsigned)(s_res[tid] & 0xffffffffull);
    }
#if DVQ_DIAG_ON
    if (dbg && tid == 0) {                                              // diagnostics build only: phase stamps (100 MHz) per workgroup
        unsigned long long* o = dbg + (size_t)blockIdx.x * 8;
        o[0] = t_start; o[1] = t_pro; o[2] = t_loop; o[3] = t_merge; o[4] = __builtin_amdgcn_s_memrealtime();
        o[5] = (unsigned long long)total; o[6] = (unsigned long long)n_slow;
    }
#endif
}

}  // namespace

// called by dvq_vq_argmin_fast (vq_stream.hip) after argument validation
int dvq_launch_vq_stream16(const float* z, const float* E, const void* packed, long M, int64_t* idx, unsigned long long* slow_rows,
                           unsigned long long* dbg, hipStream_t st) {
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            hipError_t e = hipFuncSetAttribute((const void*)&vq_stream16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#ifdef DVQ_DIAG
            for (const void* fn : {(const void*)&vq_stream16_kernel<1>, (const void*)&vq_stream16_kernel<2>, (const void*)&vq_stream16_kernel<4>,
                                   (const void*)&vq_stream16_kernel<8>, (const void*)&vq_stream16_kernel<16>, (const void*)&vq_stream16_kernel<32>,
                                   (const void*)&vq_stream16_kernel<96>, (const void*)&vq_stream16_kernel<15>, (const void*)&vq_stream16_kernel<47>,
                                   (const void*)&vq_stream16_kernel<111>, (const void*)&vq_stream16_kernel<127>}) {
                const hipError_t e1 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
                if (e1 != hipSuccess) e = e1;
            }
#endif
            return e;
        });
        if (e != hipSuccess) {
            dvq_set_error("vq_argmin_fast: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    int cus = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long tiles = (M + TILE - 1) / TILE;
    const long per_launch = (long)cus * MAX_TILES;
    for (long t0 = 0; t0 < tiles; t0 += per_launch) {
        const long nt = (tiles - t0 < per_launch) ? tiles - t0 : per_launch;
        const unsigned grid = (unsigned)(nt < cus ? nt : cus);
        unsigned long long* dp = t0 == 0 ? dbg : (unsigned long long*)nullptr;
#define DVQ_GO(A) DVQ_LAUNCH(vq_stream16_kernel<A>, dim3(grid), dim3(NT), LDS_BYTES, st, z, E, M, t0, nt, (const char*)packed, idx, slow_rows, dp)
#ifdef DVQ_DIAG
        const char* ae = getenv("DVQ_VQ16_ABL");
        switch (ae ? atoi(ae) : 0) {
            case 1: DVQ_GO(1); break;
            case 2: DVQ_GO(2); break;
            case 4: DVQ_GO(4); break;
            case 8: DVQ_GO(8); break;
            case 16: DVQ_GO(16); break;
            case 32: DVQ_GO(32); break;
            case 96: DVQ_GO(96); break;
            case 15: DVQ_GO(15); break;
            case 47: DVQ_GO(47); break;
            case 111: DVQ_GO(111); break;
            case 127: DVQ_GO(127); break;
            default: DVQ_GO(0); break;
        }
#else
        DVQ_GO(0);
#endif
#undef DVQ_GO
        DVQ_CHECK_LAUNCH("vq_stream16");
    }
    return DVQ_OK;
}
