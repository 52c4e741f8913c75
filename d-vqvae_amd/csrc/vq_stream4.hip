// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256), second structure: FOUR waves per workgroup, one
// per SIMD, the codebook in the ACCUMULATOR half of the register file.  Same result as vq_stream.hip / vq.hip /
// oracle/vq_canonical.c, bit for bit.  Reference: VectorQuantizer.forward(z, istrain=False), network/vqvae/quantizer.py:46-49.
//
// Why a second structure.  vq_stream.hip puts two waves on every SIMD (eight waves x 64 entries in 128 VGPRs each).  Its tile
// loop took 2.9 us per tile against 0.85 us of matrix work: the two waves of a SIMD share the vector issue port and the matrix
// pipe and wait for each other at the tile barrier for 47 % of their cycles (SQ counters).  Here a SIMD runs ONE
// wave that owns the whole 512-register file:
//   * wave w keeps codebook entries [128w, 128w+128) -- fp16 of -2 sE e_k, all 256 dims: 64 fragments = a[0:255], loaded once
//     by global_load straight into the accumulator registers and named literally as the A operand of
//     v_mfma_f32_32x32x16_f16 (inline asm: the compiler would put the accumulators there instead);
//   * the architectural VGPRs hold everything that vector instructions touch: two 32x32 accumulators (ping-pong over the
//     wave's four 32-entry blocks), the tile's 16 B-operand fragments (read from the fp16 LDS tile once per tile, used by
//     all four blocks), the fp32 rows in flight, scoring / conversion / merge state;
//   * per tile a wave issues 64 MFMAs; the in-order wave overlaps vector work with the matrix pipe only where that work is
//     PLACED between two MFMAs (tools/microbench/mfma_agpr_operand.hip: four to five dependent vector instructions per gap
//     are free, each further one costs ~4.5 cycles), so the tile body is generated (tools/gen_vq4_tile.py) from a gap table:
//     scores of block b-1 under the MFMAs of block b, merge of tile t-2, conversion of tile t+1, loads of tile t+2;
//   * each wave converts 8 of the tile's 32 rows (two passes of 4 rows, 16 lanes per row), merges 8 rows (8 lanes per row,
//     4 slots per lane), and one s_barrier per tile separates the fp16 images; filter, error bound, candidate lists and the
//     canonical refine are those of vq_stream.hip (see its header for the derivation of eps_row and the rules of hand-issued
//     LDS reads).
#include "dvq_internal.h"
#include "vq_pack.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int K = VQ_K, D = VQ_D;
constexpr int NWV = 4;                             // waves per workgroup = 128-entry slices of the codebook, one per SIMD
constexpr int NT = 64 * NWV;                       // 256 threads, one workgroup per CU
constexpr int TILE = 32;
constexpr int MAX_TILES = 8;

constexpr int Z16_ROW = 528;                       // padded fp16 row: 512 B + 16 B
constexpr int Z16_BUF = TILE * Z16_ROW;
constexpr int MS_ROW = 272;                        // merge slots of a row: 32 (wave, lane half, block) x 8 B + 16 B pad
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
                                                           // [4 + w] undecided rows of wave w
constexpr int L_REC = L_CNT + 64;                          // [MAX_TILES*32] {threshold, flags} of every row (merge -> tail)
constexpr int L_UND = L_REC + MAX_TILES * TILE * 8;        // [NWV][64] u16 rowslots of the rows the merge could not decide
constexpr int LDS_BYTES = L_UND + NWV * 64 * 2;
static_assert(LDS_BYTES <= 160 * 1024 && L_MS % 16 == 0 && L_RS % 16 == 0 && L_EES % 16 == 0, "LDS layout");

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
__device__ __forceinline__ int row8_sum(int v) {                // all-reduce over 8 consecutive lanes
    v += dpp_i<0xB1>(v);
    v += dpp_i<0x4E>(v);
    v += dpp_i<0x141>(v);
    return v;
}
__device__ __forceinline__ float min_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -3.0e38f); }

// entry of accumulator register i (0..15) of block b (0..3), lane half h, wave w (v_mfma_f32_32x32x16 D layout)
__device__ __forceinline__ int entry_of(int w, int b, int i, int h) { return 128 * w + 32 * b + 8 * (i >> 2) + 4 * h + (i & 3); }

__device__ __forceinline__ unsigned long long order_key(float d, int k) {
    const unsigned b = __float_as_uint(d);
    const unsigned u = (d != d) ? 0u : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    return ((unsigned long long)u << 32) | (unsigned)k;
}

// ------------------------------------------------------------------------------------------------ the codebook in a[0:255]
// Every statement that writes accumulator registers names all of them as clobbers: the compiler then allocates the file for
// the kernel and keeps its own values out of it.  AUDIT after every edit (-save-temps): no v_accvgpr_* and no a[...] outside
// ASMSTART/ASMEND, no scratch.
#define DVQ_A16(B) "a" #B "0", "a" #B "1", "a" #B "2", "a" #B "3", "a" #B "4", "a" #B "5", "a" #B "6", "a" #B "7", "a" #B "8", "a" #B "9"
#define DVQ_ACLOB                                                                                                               \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", DVQ_A16(1), DVQ_A16(2), DVQ_A16(3), DVQ_A16(4), DVQ_A16(5),     \
        DVQ_A16(6), DVQ_A16(7), DVQ_A16(8), DVQ_A16(9), DVQ_A16(10), DVQ_A16(11), DVQ_A16(12), DVQ_A16(13), DVQ_A16(14),         \
        DVQ_A16(15), DVQ_A16(16), DVQ_A16(17), DVQ_A16(18), DVQ_A16(19), DVQ_A16(20), DVQ_A16(21), DVQ_A16(22), DVQ_A16(23),     \
        DVQ_A16(24), "a250", "a251", "a252", "a253", "a254", "a255"
template <int F>
__device__ __forceinline__ void load_afrag(const f16x8* p) {
    asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off" ::"v"(p), "i"(4 * F), "i"(4 * F + 3) : "memory", DVQ_ACLOB);
}
template <int F0, int N>
struct LoadFrags {
    static __device__ __forceinline__ void run(const f16x8* p) {
        load_afrag<F0>(p + F0 * 64);
        LoadFrags<F0 + 1, N - 1>::run(p);
    }
};
template <int F0>
struct LoadFrags<F0, 0> {
    static __device__ __forceinline__ void run(const f16x8*) {}
};
// acc += A(fragment F = 16 * block + k-step, in a[4F:4F+3]) x bf
// (s_nop 1 first: the operands may have been written by a vector instruction just before; the compiler pads nothing for asm)
template <int F>
__device__ __forceinline__ void mfma_a(f32x16& acc, const f16x8& bf) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(bf), "i"(4 * F), "i"(4 * F + 3));
}
// acc = A x bf + cin (the block's first k-step: the accumulator starts at sE |e_k|^2)
template <int F>
__device__ __forceinline__ void mfma_ac(f32x16& acc, const f16x8& bf, const f32x16& cin) {
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, a[%c3:%c4], %1, %2" : "=&v"(acc) : "v"(bf), "v"(cin), "i"(4 * F), "i"(4 * F + 3));
}

// ------------------------------------------------------------------------------------------------ refine chains (as vq_stream.hip)
__device__ __forceinline__ void chain_pair_x4(const float* __restrict__ zr, const float* __restrict__ er, int q, bool active,
                                              float& zz, float& dot) {
    f32x4 x[16], y[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        x[u] = active ? *reinterpret_cast<const f32x4*>(zr + 64 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
        y[u] = active ? *reinterpret_cast<const f32x4*>(er + 64 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float a = 0.f, b = 0.f;
    const int lane = threadIdx.x & 63, base = lane & ~3;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const float a_in = round ? __shfl(a, base + round - 1) : 0.f;
        const float b_in = round ? __shfl(b, base + round - 1) : 0.f;
        float ta = a_in, tb = b_in;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ta = fmaf(x[u][c], x[u][c], ta);
                tb = fmaf(x[u][c], y[u][c], tb);
            }
        if (q == round) { a = ta; b = tb; }
    }
    zz = __shfl(a, base + 3);
    dot = __shfl(b, base + 3);
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

// ------------------------------------------------------------------------------------------------ the kernel's pieces
// Everything below is cut into pieces of a few vector instructions: the generated tile body puts ONE piece into an MFMA gap
// (a gap hides ~6 vector instructions; a 40-instruction piece in one gap is 150 exposed cycles).  Lane-constant parts of every
// address are computed once before the loop.
struct Ctx {
    const float* z;
    long M;
    long tile0;
    int G, ntl;
    int wave, lane;
    unsigned lds0;
    float emax, demax, sEf;
    bool e_valid;
    // lane-constant address parts
    unsigned zw[2];        // fp16 image, buffer 0: this lane's 8 bytes of row 8 wave + 4 pass + g   (+ 128 Q, + buffer)
    unsigned rsw[2];       // row statistics of that row, table 0                                      (+ table)
    unsigned slotw;        // merge slots of row lane % 32, slot group (wave, lane half), buffer 0   (+ 8 block, + buffer)
    unsigned mrd, rsr;     // merge: the lane's four slots of row 8 wave + lane / 8, buffer 0; that row's statistics, table 0
    const float* rp[2];    // next rows to load (pass 0 / 1): advanced by one tile stride per load
    long rstride;          // floats between a workgroup's consecutive tiles
};

// scalar (wave-uniform) per-tile offsets
__device__ __forceinline__ unsigned ms_buf(int t) { return (unsigned)((t + 1) * MS_BUF); }
__device__ __forceinline__ unsigned rs_tab(int t) { return (unsigned)((t & 3) * (TILE * 8)); }

// rows 8 wave + 4 pass + g of the workgroup's NEXT tile to load, HBM -> registers (lane (g, i): floats 4 (i + 16 q) .. +3);
// `full`: the tile lies inside the data (wave-uniform); otherwise rows behind the end repeat the last row
__device__ __forceinline__ void load_rows(Ctx& c, int pass, int j, f32x4 (&x)[4]) {
    const long tile = c.tile0 + (long)blockIdx.x + (long)j * c.G;
    const bool full = (tile + 1) * TILE <= c.M;
    const f32x4* p = reinterpret_cast<const f32x4*>(c.rp[pass]);
    if (!full) {
        long gr = tile * TILE + c.wave * 8 + pass * 4 + (c.lane >> 4);
        if (gr >= c.M) gr = c.M - 1;
        p = reinterpret_cast<const f32x4*>(c.z + gr * D) + (c.lane & 15);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = __builtin_nontemporal_load(p + 16 * q);
    c.rp[pass] += c.rstride;
}

// The same inside the tile loop, hand-issued: the compiler's own wait for a row loaded a tile earlier is vmcnt(3) / vmcnt(0) --
// it also waits for the OTHER pass's younger loads, issued a third of a tile ago -- so these loads are asm and the generated
// body waits with the counts of the loads' program order.  Rule: the destination registers are pinned ("+v") by the wait.
template <int OFF>
__device__ __forceinline__ void gld128(f32x4& d, const f32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2 nt" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void load_rows_asm(Ctx& c, int pass, int j, f32x4 (&x)[4]) {
    const long tile = c.tile0 + (long)blockIdx.x + (long)j * c.G;
    const bool full = (tile + 1) * TILE <= c.M;
    const f32x4* p = reinterpret_cast<const f32x4*>(c.rp[pass]);
    if (!full) {
        long gr = tile * TILE + c.wave * 8 + pass * 4 + (c.lane >> 4);
        if (gr >= c.M) gr = c.M - 1;
        p = reinterpret_cast<const f32x4*>(c.z + gr * D) + (c.lane & 15);
    }
    gld128<0>(x[0], p);
    gld128<256>(x[1], p);
    gld128<512>(x[2], p);
    gld128<768>(x[3], p);
    c.rp[pass] += c.rstride;
}

// scores of accumulator registers [B, E): register index in the low 5 mantissa bits, lane-local top-2 update
template <int B, int E>
__device__ __forceinline__ void score(const f32x16& a, float& m1, float& m2) {
#pragma unroll
    for (int i = B; i < E; ++i) {
        const float p = __uint_as_float((__float_as_uint(a[i]) & ~31u) | (unsigned)i);
        m2 = __builtin_amdgcn_fmed3f(m1, m2, p);
        m1 = min_nc(m1, p);
    }
}
// (min, second) of block BLK's 16 scores -> slot (wave, lane half, block) of the lane's row, tile t
template <int BLK>
__device__ __forceinline__ void write_slot(const Ctx& c, int t, float m1, float m2) {
    f32x2 v;
    v[0] = m1;
    v[1] = m2;
    ds_wr64<8 * BLK>(c.slotw + ms_buf(t), v);
}

template <int HI>
__device__ __forceinline__ float mix_diff(float hp, float x) {
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    return d;
}

#ifndef DVQ_MEASURE_DZ
#define DVQ_MEASURE_DZ 1   // 1: |z - h(z)| measured element by element; 0: half-ulp bound (fewer vector instructions, ~1.5x the pairs)
#endif
// fp32 -> fp16 conversion of 4 of the wave's rows (one pass), staged
struct Convert {
    float hh, dsq, hn, dzn, eps;
    f32x2 pk;
    unsigned za;                                                          // image address of this pass in the tile being written
    __device__ __forceinline__ void start(const Ctx& c, int j, int pass) {
        hh = 0.f;
        dsq = 0.f;
        za = c.zw[pass] + (unsigned)((j & 1) * Z16_BUF);
    }
    template <int Q>
    __device__ __forceinline__ void cvt(const f32x4 (&x)[4]) {
        f32x2 a, b;
        a[0] = x[Q][0]; a[1] = x[Q][1]; b[0] = x[Q][2]; b[1] = x[Q][3];
        const f16x2 lo = __builtin_convertvector(a, f16x2), hi = __builtin_convertvector(b, f16x2);
        hh = __builtin_amdgcn_fdot2(lo, lo, hh, false);
        hh = __builtin_amdgcn_fdot2(hi, hi, hh, false);
        pk[0] = __builtin_bit_cast(float, lo);
        pk[1] = __builtin_bit_cast(float, hi);
        ds_wr64<128 * Q>(za, pk);
    }
    template <int Q, int HALF>
    __device__ __forceinline__ void err(const f32x2& p, const f32x4 (&x)[4]) {     // rounding error of two of quad Q's four elements
        if (DVQ_MEASURE_DZ) {
            const float d0 = mix_diff<0>(p[HALF], x[Q][2 * HALF]), d1 = mix_diff<1>(p[HALF], x[Q][2 * HALF + 1]);
            dsq = fmaf(d0, d0, dsq);
            dsq = fmaf(d1, d1, dsq);
        }
    }
    __device__ __forceinline__ void sum_hh() { hh = row16_sum(hh); }
    __device__ __forceinline__ void sum_dsq() { if (DVQ_MEASURE_DZ) dsq = row16_sum(dsq); }
    __device__ __forceinline__ void fin_a() {
        hn = __builtin_amdgcn_sqrtf(hh);
        // |z - h(z)|: measured, or a priori (half an ulp of a normal fp16 is at most 2^-11 |h|, of a subnormal one 2^-25)
        dzn = DVQ_MEASURE_DZ ? __builtin_amdgcn_sqrtf(dsq) * 1.0001f : hn * 4.8877e-4f + 4.8e-7f;
    }
    __device__ __forceinline__ void fin_b(const Ctx& c) {
        const float zn = (hn + dzn) * 1.0001f;                          // |z| <= |h(z)| + |z - h(z)|
        const float u = zn + c.emax;
        eps = 4.004f * (dzn * c.emax + zn * c.demax + dzn * c.demax) + 1.01e-4f * u * u;
    }
    __device__ __forceinline__ void fin_c(const Ctx& c, int j, int pass) {
        const float epsS = eps * c.sEf;
        const bool bad = !c.e_valid || !(hh <= 3.0e38f) || !(dsq <= 3.0e38f) || !(epsS <= 3.0e38f);   // NaN/Inf, fp16 overflow
        if ((c.lane & 15) == 0) {
            f32x2 rs;
            rs[0] = epsS;
            rs[1] = __uint_as_float(bad ? 1u : 0u);
            ds_wr64<0>(c.rsw[pass] + rs_tab(j), rs);
        }
    }
};

// merge of a tile, staged: this wave's rows 8 wave + g (g = lane / 8), 8 lanes per row; lane i looks at the four slots (blocks
// 0..3) of (source wave i / 2, lane half i % 2), i.e. at 4 x 16 codebook entries
struct Merge {
    f32x4 s0, s1;                                                       // (m1, m2) of blocks 0, 1 | blocks 2, 3
    f32x2 rs;
    float thr;
    int c1, c2, n1, n2;
    __device__ __forceinline__ void read(const Ctx& c, int tm) {
        const unsigned a = c.mrd + ms_buf(tm);
        ds_rd128<0>(s0, a);
        ds_rd128<16>(s1, a);
        ds_rd64<0>(rs, c.rsr + rs_tab(tm));
    }
    __device__ __forceinline__ void min_a() {                            // after the wait that pins s0, s1, rs
        thr = min_nc(min_nc(s0[0], s0[2]), min_nc(s1[0], s1[2]));
        thr = min_nc(thr, dpp_f<0xB1>(thr));
    }
    __device__ __forceinline__ void min_b() {
        thr = min_nc(thr, dpp_f<0x4E>(thr));
        thr = min_nc(thr, dpp_f<0x141>(thr));
        thr = thr + rs[0];
    }
    __device__ __forceinline__ void count_a() { c1 = (int)(s0[0] <= thr) + (int)(s0[2] <= thr) + (int)(s1[0] <= thr) + (int)(s1[2] <= thr); }
    __device__ __forceinline__ void count_b() { c2 = (int)(s0[1] <= thr) + (int)(s0[3] <= thr) + (int)(s1[1] <= thr) + (int)(s1[3] <= thr); }
    __device__ __forceinline__ void sum_a() { n1 = row8_sum(c1); }
    __device__ __forceinline__ void sum_b() { n2 = row8_sum(c2); }
    // decisions; `rows` = rows of tile tm that exist (wave-uniform: 32 except behind the end of the data, <= 0 for "tiles" -2, -1)
    unsigned flags;                                                     // bit 0 live, 1 slow, 2 unique, 3 ambiguous
    unsigned rowslot;
    __device__ __forceinline__ void act_a(const Ctx& c, int tm, int rows) {
        const int r = 8 * c.wave + (c.lane >> 3);
        const bool bad = __float_as_uint(rs[1]) != 0u;
        const bool live = r < rows;
        const bool slow = bad || n1 == 0;
        const bool unique = !slow && n1 == 1 && n2 == 0;
        const bool amb = live && !slow && !unique;
        flags = (live ? 1u : 0u) | (slow ? 2u : 0u) | (unique ? 4u : 0u) | (amb ? 8u : 0u);
        rowslot = (unsigned)(tm * TILE + r);
    }
    __device__ __forceinline__ void act_b(const Ctx& c) {               // the decided rows: the winner lane writes the entry
        const int i = c.lane & 7;
        if ((flags & 5u) == 5u && c1) {
            const bool a0 = s0[0] <= thr, a1 = s0[2] <= thr, a2 = s1[0] <= thr;
            const int j = a0 ? 0 : (a1 ? 1 : (a2 ? 2 : 3));
            const float best = a0 ? s0[0] : (a1 ? s0[2] : (a2 ? s1[0] : s1[2]));
            f32x2 kv;
            kv[0] = __uint_as_float((unsigned)entry_of(i >> 1, j, (int)(__float_as_uint(best) & 15u), i & 1));
            kv[1] = __uint_as_float(0u);
            ds_wr64<0>(c.lds0 + L_RES + rowslot * 8, kv);
        }
    }
    // every row's threshold and flags go to the row record; the rows the merge could not decide (ambiguous, or all-entries) are
    // appended to the WAVE's own list (no atomics, no branch: the tail expands them into candidate pairs)
    __device__ __forceinline__ void act_c(const Ctx& c, int& und) {
        const bool lead = (c.lane & 7) == 0;
        if (lead && (flags & 1u)) {                                      // live rows only: "tiles" -2, -1 have no record
            f32x2 rec;
            rec[0] = thr;
            rec[1] = __uint_as_float(flags);
            ds_wr64<0>(c.lds0 + L_REC + rowslot * 8, rec);
        }
        const bool put = lead && (flags & 1u) && !(flags & 4u);          // live and not decided
        const unsigned long long m = __ballot(put);
        const int pos = und + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (put && pos < 64) ds_wr16(c.lds0 + L_UND + (c.wave * 64 + pos) * 2, rowslot);
        und += __popcll(m);
    }
    // tail: the undecided row `rowslot_` (8 lanes per row, as in the merge): candidate pairs / all-entries list
    __device__ __forceinline__ void expand(const Ctx& c, unsigned rowslot_, bool valid) {
        const int i = c.lane & 7;
        const int tm = (int)(rowslot_ >> 5), r = (int)(rowslot_ & 31);
        const unsigned a = c.lds0 + L_MS + ms_buf(tm) + r * MS_ROW + i * 32;
        f32x2 rec;
        ds_rd128<0>(s0, a);
        ds_rd128<16>(s1, a);
        ds_rd64<0>(rec, c.lds0 + L_REC + rowslot_ * 8);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(rec));
        thr = rec[0];
        flags = valid ? __float_as_uint(rec[1]) : 0u;
        rowslot = rowslot_;
        const bool live = flags & 1u, slow = flags & 2u, amb = flags & 8u;
        const int w_src = i >> 1, h_src = i & 1;
        const bool a0 = s0[0] <= thr, b0 = s0[1] <= thr, a1 = s0[2] <= thr, b1 = s0[3] <= thr;
        const bool a2 = s1[0] <= thr, b2 = s1[1] <= thr, a3 = s1[2] <= thr, b3 = s1[3] <= thr;
        c1 = (int)a0 + (int)a1 + (int)a2 + (int)a3;
        c2 = (int)b0 + (int)b1 + (int)b2 + (int)b3;
        n1 = row8_sum(c1);
        n2 = row8_sum(c2);
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
                if (all) {
                    for (int e = 0; e < 16; ++e) ds_wr32(c.lds0 + L_PAIR + (off + e) * 4, (rowslot << 16) | (unsigned)entry_of(w_src, j, e, h_src));
                    off += 16;
                } else if (one) {
                    ds_wr32(c.lds0 + L_PAIR + off * 4, (rowslot << 16) | (unsigned)entry_of(w_src, j, (int)(__float_as_uint(best) & 15u), h_src));
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
    __device__ __forceinline__ void all(const Ctx& c, int tm, int rows, int& und) {   // unstaged (after the loop)
        min_a(); min_b(); count_a(); count_b(); sum_a(); sum_b(); act_a(c, tm, rows); act_b(c); act_c(c, und);
    }
};

// rows of local tile j that exist (wave-uniform): 32 inside the data, fewer in the last tile, <= 0 for j < 0
__device__ __forceinline__ int rows_of(const Ctx& c, int j) {
    if (j < 0) return 0;
    const long first = (c.tile0 + (long)blockIdx.x + (long)j * c.G) * TILE;
    const long left = c.M - first;
    return left >= TILE ? TILE : (int)(left > 0 ? left : 0);
}

template <int ABL>
__global__ __launch_bounds__(NT, 1) void vq_stream4_kernel(const float* __restrict__ z, const float* __restrict__ E, long M, long tile0,
                                                           long n_tiles, const char* __restrict__ packed, int64_t* __restrict__ idx,
                                                           unsigned long long* __restrict__ slow_rows, unsigned long long* __restrict__ dbg) {
    constexpr int abl = ABL;      // diagnostics build only (DVQ_VQ4_ABL; results INVALID unless 0): 1 no row loads in the loop, 2 no merge
                                  // decisions, 4 no rounding-error measurement, 8 no scoring, 16 no barrier, 32 no MFMA, 1024 segment stamps
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

    // ---- prologue: the codebook slice (64 fragments of 1 KiB per wave) straight into a[0:255], then the first rows
    LoadFrags<0, 64>::run(reinterpret_cast<const f16x8*>(packed + PK_OFF_IMG) + (size_t)c.wave * (64 * 64) + c.lane);
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    c.emax = hdr->emax;
    c.demax = hdr->demax;
    c.e_valid = hdr->valid != 0;
    c.sEf = c.e_valid ? pow2f(hdr->sexp) : 1.0f;
    {
        const int g4 = c.lane >> 4, i16 = c.lane & 15, g8 = c.lane >> 3, i8 = c.lane & 7;
        for (int p = 0; p < 2; ++p) {
            const int row = c.wave * 8 + p * 4 + g4;
            c.zw[p] = c.lds0 + L_Z16 + row * Z16_ROW + 8 * i16;
            c.rsw[p] = c.lds0 + L_RS + row * 8;
            c.rp[p] = z + ((tile0 + (long)blockIdx.x) * TILE + row) * D + 4 * i16;
        }
        c.slotw = c.lds0 + L_MS + (c.lane & 31) * MS_ROW + (2 * c.wave + (c.lane >> 5)) * 32;
        c.mrd = c.lds0 + L_MS + (8 * c.wave + g8) * MS_ROW + i8 * 32;
        c.rsr = c.lds0 + L_RS + (8 * c.wave + g8) * 8;
        c.rstride = (long)c.G * TILE * D;
    }
    f32x4 x0[4], x1[4];                                                  // fp32 rows in flight: pass 0 (rows 8w..8w+3), pass 1 (8w+4..8w+7)
    load_rows(c, 0, 0, x0);
    load_rows(c, 1, 0, x1);
    for (int u = tid; u < MAX_TILES * TILE; u += NT) reinterpret_cast<unsigned long long*>(lds + L_RES)[u] = ~0ull;
    for (int u = tid; u < PAIR_CAP; u += NT) reinterpret_cast<unsigned*>(lds + L_PAIR)[u] = ~0u;
    if (tid < 3) reinterpret_cast<unsigned*>(lds + L_CNT)[tid] = 0u;
    reinterpret_cast<float*>(lds + L_EES)[tid] = ee_g[tid] * c.sEf;
    reinterpret_cast<float*>(lds + L_EES)[tid + NT] = ee_g[tid + NT] * c.sEf;
    Convert cv;
    {
        f32x2 p0, p1, p2, p3;
        cv.start(c, 0, 0);
        cv.cvt<0>(x0); p0 = cv.pk; cv.cvt<1>(x0); p1 = cv.pk; cv.cvt<2>(x0); p2 = cv.pk; cv.cvt<3>(x0); p3 = cv.pk;
        cv.err<0, 0>(p0, x0); cv.err<0, 1>(p0, x0); cv.err<1, 0>(p1, x0); cv.err<1, 1>(p1, x0);
        cv.err<2, 0>(p2, x0); cv.err<2, 1>(p2, x0); cv.err<3, 0>(p3, x0); cv.err<3, 1>(p3, x0);
        cv.sum_hh(); cv.sum_dsq(); cv.fin_a(); cv.fin_b(c); cv.fin_c(c, 0, 0);
        cv.start(c, 0, 1);
        cv.cvt<0>(x1); p0 = cv.pk; cv.cvt<1>(x1); p1 = cv.pk; cv.cvt<2>(x1); p2 = cv.pk; cv.cvt<3>(x1); p3 = cv.pk;
        cv.err<0, 0>(p0, x1); cv.err<0, 1>(p0, x1); cv.err<1, 0>(p1, x1); cv.err<1, 1>(p1, x1);
        cv.err<2, 0>(p2, x1); cv.err<2, 1>(p2, x1); cv.err<3, 0>(p3, x1); cv.err<3, 1>(p3, x1);
        cv.sum_hh(); cv.sum_dsq(); cv.fin_a(); cv.fin_b(c); cv.fin_c(c, 0, 1);
        // tile 1's rows, hand-issued as in the loop: pass 0 first, then pass 1 (the loop's waits count on this order)
        if (c.ntl > 1) { load_rows_asm(c, 0, 1, x0); load_rows_asm(c, 1, 1, x1); }
    }
    // the rows of tile 0 are YOUNGER than the codebook loads and loads return in order: once they have been used, a[0:255] is
    // complete.  (The wait the compiler placed before the first use of x0 therefore covered the codebook slice as well.)
    wg_barrier();
    DVQ_T(t_pro);

    // ---- tile loop: 64 MFMA gaps per tile, generated (tools/gen_vq4_tile.py)
    f32x16 accP = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, accQ = accP;
    f16x8 bf[16];
    f32x4 ci[4];
    const int r_l = c.lane & 31, h_l = c.lane >> 5;
    const unsigned ea = c.lds0 + L_EES + (128 * c.wave + 4 * h_l) * 4;
    const unsigned zbase = c.lds0 + L_Z16 + r_l * Z16_ROW + 16 * h_l;
    Merge mg;
    int und = 0;                                                         // rows this wave's merges left undecided (wave-uniform)
#if DVQ_DIAG_ON
    // per-wave sums of the shader-clock time between stamps (tile top -> barrier passed -> end of block 0, 1, 2, 3 -> next tile top)
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0}, last = __builtin_amdgcn_s_memtime();
#define DVQ_STAMP(I) if (abl & 1024) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); seg[I] += now_ - last; last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define DVQ_STAMP(I)
#endif
#if DVQ_DIAG_ON
    // one shader-clock stamp behind every MFMA gap (workgroup 0 only): where a tile's cycles go, gap by gap
#define DVQ_GSTAMP(G) if ((abl & 2048) && dbg && blockIdx.x == 0) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); if (c.lane == 0) reinterpret_cast<unsigned*>(dbg + 16384)[(c.wave * 8 + t) * 64 + (G)] = (unsigned)now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define DVQ_GSTAMP(G)
#endif
#define DVQ_SB() __builtin_amdgcn_sched_barrier(0)
#define DVQ_PIN2(A, B) asm volatile("" : "+v"(A), "+v"(B))
#define DVQ_RDF(S) ds_rd128<32 * (S)>(bf[S], za)
#define DVQ_RDCI(BLK) { ds_rd128<128 * (BLK)>(ci[0], ea); ds_rd128<128 * (BLK) + 32>(ci[1], ea); ds_rd128<128 * (BLK) + 64>(ci[2], ea); ds_rd128<128 * (BLK) + 96>(ci[3], ea); }
#define DVQ_START(ACC) { _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) ACC[4 * q_ + e_] = ci[q_][e_]; }
#define DVQ_SCORE(ACC, B, E) { score<B, E>(ACC, m1, m2); DVQ_PIN2(m1, m2); }
    auto tile = [&](const int t) __attribute__((always_inline)) {
    // <<< GENERATED by tools/gen_vq4_tile.py
        const bool do_load = t + 2 < c.ntl && !(abl & 1);
        const unsigned za = zbase + (t & 1) * Z16_BUF;
        const int rows_m2 = rows_of(c, t - 2);
        DVQ_STAMP(0);
        if (abl & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DVQ_STAMP(1);
        DVQ_RDCI(0)
        DVQ_RDF(0);
        DVQ_RDF(1);
        DVQ_RDF(2);
        float m1 = INFINITY, m2 = INFINITY;
        f32x2 pk0, pk1, pk2, pk3;
        // gap 0: block 0, k-step 0
        m1 = INFINITY; m2 = INFINITY;
        mg.read(c, t - 2);
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ci[0]), "+v"(ci[1]), "+v"(ci[2]), "+v"(ci[3]), "+v"(bf[0]));
        { f32x16 st; DVQ_START(st) if (abl & 32) accP = st; else mfma_ac<0>(accP, bf[0], st); }
        DVQ_RDF(3);
        DVQ_SB();
        DVQ_GSTAMP(0);
        // gap 1: block 0, k-step 1
        asm volatile("s_waitcnt vmcnt(7)" : "+v"(x0[0]));
        { cv.start(c, t + 1, 0); cv.template cvt<0>(x0); pk0 = cv.pk; DVQ_PIN2(cv.hh, pk0); }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bf[1]));
        if (!(abl & 32)) mfma_a<1>(accP, bf[1]);
        DVQ_RDF(4);
        DVQ_SB();
        DVQ_GSTAMP(1);
        // gap 2: block 0, k-step 2
        if (!(abl & 8)) DVQ_SCORE(accQ, 0, 2)
        if (!(abl & 4)) { cv.template err<0, 0>(pk0, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bf[2]));
        if (!(abl & 32)) mfma_a<2>(accP, bf[2]);
        DVQ_RDF(5);
        DVQ_SB();
        DVQ_GSTAMP(2);
        // gap 3: block 0, k-step 3
        if (!(abl & 8)) DVQ_SCORE(accQ, 2, 4)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(mg.s0), "+v"(mg.s1), "+v"(mg.rs));
        if (!(abl & 2)) { mg.min_a(); asm volatile("" : "+v"(mg.thr)); }
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[3]));
        if (!(abl & 32)) mfma_a<3>(accP, bf[3]);
        DVQ_RDF(6);
        DVQ_SB();
        DVQ_GSTAMP(3);
        // gap 4: block 0, k-step 4
        if (!(abl & 8)) DVQ_SCORE(accQ, 4, 6)
        if (!(abl & 2)) { mg.min_b(); asm volatile("" : "+v"(mg.thr)); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[4]));
        if (!(abl & 32)) mfma_a<4>(accP, bf[4]);
        DVQ_RDF(7);
        DVQ_SB();
        DVQ_GSTAMP(4);
        // gap 5: block 0, k-step 5
        if (!(abl & 8)) DVQ_SCORE(accQ, 6, 8)
        if (!(abl & 2)) { mg.count_a(); asm volatile("" : "+v"(mg.c1)); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[5]));
        if (!(abl & 32)) mfma_a<5>(accP, bf[5]);
        DVQ_RDF(8);
        DVQ_SB();
        DVQ_GSTAMP(5);
        // gap 6: block 0, k-step 6
        if (!(abl & 8)) DVQ_SCORE(accQ, 8, 10)
        if (!(abl & 2)) { mg.count_b(); asm volatile("" : "+v"(mg.c2)); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[6]));
        if (!(abl & 32)) mfma_a<6>(accP, bf[6]);
        DVQ_RDF(9);
        DVQ_SB();
        DVQ_GSTAMP(6);
        // gap 7: block 0, k-step 7
        if (!(abl & 8)) DVQ_SCORE(accQ, 10, 12)
        if (!(abl & 2)) { mg.sum_a(); asm volatile("" : "+v"(mg.n1)); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[7]));
        if (!(abl & 32)) mfma_a<7>(accP, bf[7]);
        DVQ_RDF(10);
        DVQ_SB();
        DVQ_GSTAMP(7);
        // gap 8: block 0, k-step 8
        if (!(abl & 8)) DVQ_SCORE(accQ, 12, 14)
        if (!(abl & 2)) { mg.sum_b(); asm volatile("" : "+v"(mg.n2)); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[8]));
        if (!(abl & 32)) mfma_a<8>(accP, bf[8]);
        DVQ_RDF(11);
        DVQ_SB();
        DVQ_GSTAMP(8);
        // gap 9: block 0, k-step 9
        if (!(abl & 8)) DVQ_SCORE(accQ, 14, 16)
        if (!(abl & 4)) { cv.template err<0, 1>(pk0, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[9]));
        if (!(abl & 32)) mfma_a<9>(accP, bf[9]);
        DVQ_RDF(12);
        DVQ_SB();
        DVQ_GSTAMP(9);
        // gap 10: block 0, k-step 10
        write_slot<3>(c, t - 1, m1, m2);
        if (!(abl & 2)) { mg.act_a(c, t - 2, rows_m2); asm volatile("" : "+v"(mg.flags), "+v"(mg.rowslot)); }
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[10]));
        if (!(abl & 32)) mfma_a<10>(accP, bf[10]);
        DVQ_RDF(13);
        DVQ_SB();
        DVQ_GSTAMP(10);
        // gap 11: block 0, k-step 11
        if (!(abl & 2)) mg.act_b(c);
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bf[11]));
        if (!(abl & 32)) mfma_a<11>(accP, bf[11]);
        DVQ_RDF(14);
        DVQ_RDCI(1)
        DVQ_SB();
        DVQ_GSTAMP(11);
        // gap 12: block 0, k-step 12
        if (!(abl & 2)) mg.act_c(c, und);
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bf[12]));
        if (!(abl & 32)) mfma_a<12>(accP, bf[12]);
        DVQ_RDF(15);
        DVQ_SB();
        DVQ_GSTAMP(12);
        // gap 13: block 0, k-step 13
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(x0[1]));
        { cv.template cvt<1>(x0); pk1 = cv.pk; DVQ_PIN2(cv.hh, pk1); }
        if (!(abl & 4)) { cv.template err<1, 0>(pk1, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bf[13]));
        if (!(abl & 32)) mfma_a<13>(accP, bf[13]);
        DVQ_SB();
        DVQ_GSTAMP(13);
        // gap 14: block 0, k-step 14
        if (!(abl & 4)) { cv.template err<1, 1>(pk1, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt vmcnt(5)" : "+v"(x0[2]));
        { cv.template cvt<2>(x0); pk2 = cv.pk; DVQ_PIN2(cv.hh, pk2); }
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bf[14]));
        if (!(abl & 32)) mfma_a<14>(accP, bf[14]);
        DVQ_SB();
        DVQ_GSTAMP(14);
        // gap 15: block 0, k-step 15
        if (!(abl & 4)) { cv.template err<2, 0>(pk2, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 4)) { cv.template err<2, 1>(pk2, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[15]));
        if (!(abl & 32)) mfma_a<15>(accP, bf[15]);
        DVQ_SB();
        DVQ_GSTAMP(15);
        DVQ_STAMP(2);
        // gap 16: block 1, k-step 0
        m1 = INFINITY; m2 = INFINITY;
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(x0[3]));
        { cv.template cvt<3>(x0); pk3 = cv.pk; DVQ_PIN2(cv.hh, pk3); }
        if (!(abl & 4)) { cv.template err<3, 0>(pk3, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ci[0]), "+v"(ci[1]), "+v"(ci[2]), "+v"(ci[3]));
        { f32x16 st; DVQ_START(st) if (abl & 32) accQ = st; else mfma_ac<16>(accQ, bf[0], st); }
        DVQ_SB();
        DVQ_GSTAMP(16);
        // gap 17: block 1, k-step 1
        if (!(abl & 4)) { cv.template err<3, 1>(pk3, x0); DVQ_PIN2(cv.hh, cv.dsq); }
        { cv.sum_hh(); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<17>(accQ, bf[1]);
        DVQ_SB();
        DVQ_GSTAMP(17);
        // gap 18: block 1, k-step 2
        if (!(abl & 8)) DVQ_SCORE(accP, 0, 2)
        { cv.sum_dsq(); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<18>(accQ, bf[2]);
        DVQ_SB();
        DVQ_GSTAMP(18);
        // gap 19: block 1, k-step 3
        if (!(abl & 8)) DVQ_SCORE(accP, 2, 4)
        { cv.fin_a(); DVQ_PIN2(cv.hn, cv.dzn); }
        if (!(abl & 32)) mfma_a<19>(accQ, bf[3]);
        DVQ_SB();
        DVQ_GSTAMP(19);
        // gap 20: block 1, k-step 4
        if (!(abl & 8)) DVQ_SCORE(accP, 4, 6)
        { cv.fin_b(c); asm volatile("" : "+v"(cv.eps)); }
        if (!(abl & 32)) mfma_a<20>(accQ, bf[4]);
        DVQ_SB();
        DVQ_GSTAMP(20);
        // gap 21: block 1, k-step 5
        if (!(abl & 8)) DVQ_SCORE(accP, 6, 8)
        if (!(abl & 32)) mfma_a<21>(accQ, bf[5]);
        DVQ_SB();
        DVQ_GSTAMP(21);
        // gap 22: block 1, k-step 6
        if (!(abl & 8)) DVQ_SCORE(accP, 8, 10)
        if (!(abl & 32)) mfma_a<22>(accQ, bf[6]);
        DVQ_SB();
        DVQ_GSTAMP(22);
        // gap 23: block 1, k-step 7
        if (!(abl & 8)) DVQ_SCORE(accP, 10, 12)
        if (!(abl & 32)) mfma_a<23>(accQ, bf[7]);
        DVQ_SB();
        DVQ_GSTAMP(23);
        // gap 24: block 1, k-step 8
        if (!(abl & 8)) DVQ_SCORE(accP, 12, 14)
        if (!(abl & 32)) mfma_a<24>(accQ, bf[8]);
        DVQ_SB();
        DVQ_GSTAMP(24);
        // gap 25: block 1, k-step 9
        if (!(abl & 8)) DVQ_SCORE(accP, 14, 16)
        if (!(abl & 32)) mfma_a<25>(accQ, bf[9]);
        DVQ_SB();
        DVQ_GSTAMP(25);
        // gap 26: block 1, k-step 10
        write_slot<0>(c, t, m1, m2);
        cv.fin_c(c, t + 1, 0);
        if (!(abl & 32)) mfma_a<26>(accQ, bf[10]);
        DVQ_SB();
        DVQ_GSTAMP(26);
        // gap 27: block 1, k-step 11
        if (do_load) load_rows_asm(c, 0, t + 2, x0);
        if (!(abl & 32)) mfma_a<27>(accQ, bf[11]);
        DVQ_RDCI(2)
        DVQ_SB();
        DVQ_GSTAMP(27);
        // gap 28: block 1, k-step 12
        asm volatile("s_waitcnt vmcnt(7)" : "+v"(x1[0]));
        { cv.start(c, t + 1, 1); cv.template cvt<0>(x1); pk0 = cv.pk; DVQ_PIN2(cv.hh, pk0); }
        if (!(abl & 32)) mfma_a<28>(accQ, bf[12]);
        DVQ_SB();
        DVQ_GSTAMP(28);
        // gap 29: block 1, k-step 13
        if (!(abl & 4)) { cv.template err<0, 0>(pk0, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 4)) { cv.template err<0, 1>(pk0, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<29>(accQ, bf[13]);
        DVQ_SB();
        DVQ_GSTAMP(29);
        // gap 30: block 1, k-step 14
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(x1[1]));
        { cv.template cvt<1>(x1); pk1 = cv.pk; DVQ_PIN2(cv.hh, pk1); }
        if (!(abl & 4)) { cv.template err<1, 0>(pk1, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<30>(accQ, bf[14]);
        DVQ_SB();
        DVQ_GSTAMP(30);
        // gap 31: block 1, k-step 15
        if (!(abl & 4)) { cv.template err<1, 1>(pk1, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt vmcnt(5)" : "+v"(x1[2]));
        { cv.template cvt<2>(x1); pk2 = cv.pk; DVQ_PIN2(cv.hh, pk2); }
        if (!(abl & 32)) mfma_a<31>(accQ, bf[15]);
        DVQ_SB();
        DVQ_GSTAMP(31);
        DVQ_STAMP(3);
        // gap 32: block 2, k-step 0
        m1 = INFINITY; m2 = INFINITY;
        if (!(abl & 4)) { cv.template err<2, 0>(pk2, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 4)) { cv.template err<2, 1>(pk2, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ci[0]), "+v"(ci[1]), "+v"(ci[2]), "+v"(ci[3]));
        { f32x16 st; DVQ_START(st) if (abl & 32) accP = st; else mfma_ac<32>(accP, bf[0], st); }
        DVQ_SB();
        DVQ_GSTAMP(32);
        // gap 33: block 2, k-step 1
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(x1[3]));
        { cv.template cvt<3>(x1); pk3 = cv.pk; DVQ_PIN2(cv.hh, pk3); }
        if (!(abl & 4)) { cv.template err<3, 0>(pk3, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<33>(accP, bf[1]);
        DVQ_SB();
        DVQ_GSTAMP(33);
        // gap 34: block 2, k-step 2
        if (!(abl & 8)) DVQ_SCORE(accQ, 0, 2)
        if (!(abl & 4)) { cv.template err<3, 1>(pk3, x1); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<34>(accP, bf[2]);
        DVQ_SB();
        DVQ_GSTAMP(34);
        // gap 35: block 2, k-step 3
        if (!(abl & 8)) DVQ_SCORE(accQ, 2, 4)
        { cv.sum_hh(); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<35>(accP, bf[3]);
        DVQ_SB();
        DVQ_GSTAMP(35);
        // gap 36: block 2, k-step 4
        if (!(abl & 8)) DVQ_SCORE(accQ, 4, 6)
        { cv.sum_dsq(); DVQ_PIN2(cv.hh, cv.dsq); }
        if (!(abl & 32)) mfma_a<36>(accP, bf[4]);
        DVQ_SB();
        DVQ_GSTAMP(36);
        // gap 37: block 2, k-step 5
        if (!(abl & 8)) DVQ_SCORE(accQ, 6, 8)
        { cv.fin_a(); DVQ_PIN2(cv.hn, cv.dzn); }
        if (!(abl & 32)) mfma_a<37>(accP, bf[5]);
        DVQ_SB();
        DVQ_GSTAMP(37);
        // gap 38: block 2, k-step 6
        if (!(abl & 8)) DVQ_SCORE(accQ, 8, 10)
        { cv.fin_b(c); asm volatile("" : "+v"(cv.eps)); }
        if (!(abl & 32)) mfma_a<38>(accP, bf[6]);
        DVQ_SB();
        DVQ_GSTAMP(38);
        // gap 39: block 2, k-step 7
        if (!(abl & 8)) DVQ_SCORE(accQ, 10, 12)
        if (!(abl & 32)) mfma_a<39>(accP, bf[7]);
        DVQ_SB();
        DVQ_GSTAMP(39);
        // gap 40: block 2, k-step 8
        if (!(abl & 8)) DVQ_SCORE(accQ, 12, 14)
        if (!(abl & 32)) mfma_a<40>(accP, bf[8]);
        DVQ_SB();
        DVQ_GSTAMP(40);
        // gap 41: block 2, k-step 9
        if (!(abl & 8)) DVQ_SCORE(accQ, 14, 16)
        if (!(abl & 32)) mfma_a<41>(accP, bf[9]);
        DVQ_SB();
        DVQ_GSTAMP(41);
        // gap 42: block 2, k-step 10
        write_slot<1>(c, t, m1, m2);
        cv.fin_c(c, t + 1, 1);
        if (!(abl & 32)) mfma_a<42>(accP, bf[10]);
        DVQ_SB();
        DVQ_GSTAMP(42);
        // gap 43: block 2, k-step 11
        if (do_load) load_rows_asm(c, 1, t + 2, x1);
        if (!(abl & 32)) mfma_a<43>(accP, bf[11]);
        DVQ_RDCI(3)
        DVQ_SB();
        DVQ_GSTAMP(43);
        // gap 44: block 2, k-step 12
        if (!(abl & 32)) mfma_a<44>(accP, bf[12]);
        DVQ_SB();
        DVQ_GSTAMP(44);
        // gap 45: block 2, k-step 13
        if (!(abl & 32)) mfma_a<45>(accP, bf[13]);
        DVQ_SB();
        DVQ_GSTAMP(45);
        // gap 46: block 2, k-step 14
        if (!(abl & 32)) mfma_a<46>(accP, bf[14]);
        DVQ_SB();
        DVQ_GSTAMP(46);
        // gap 47: block 2, k-step 15
        if (!(abl & 32)) mfma_a<47>(accP, bf[15]);
        DVQ_SB();
        DVQ_GSTAMP(47);
        DVQ_STAMP(4);
        // gap 48: block 3, k-step 0
        m1 = INFINITY; m2 = INFINITY;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ci[0]), "+v"(ci[1]), "+v"(ci[2]), "+v"(ci[3]));
        { f32x16 st; DVQ_START(st) if (abl & 32) accQ = st; else mfma_ac<48>(accQ, bf[0], st); }
        DVQ_SB();
        DVQ_GSTAMP(48);
        // gap 49: block 3, k-step 1
        if (!(abl & 32)) mfma_a<49>(accQ, bf[1]);
        DVQ_SB();
        DVQ_GSTAMP(49);
        // gap 50: block 3, k-step 2
        if (!(abl & 8)) DVQ_SCORE(accP, 0, 2)
        if (!(abl & 32)) mfma_a<50>(accQ, bf[2]);
        DVQ_SB();
        DVQ_GSTAMP(50);
        // gap 51: block 3, k-step 3
        if (!(abl & 8)) DVQ_SCORE(accP, 2, 4)
        if (!(abl & 32)) mfma_a<51>(accQ, bf[3]);
        DVQ_SB();
        DVQ_GSTAMP(51);
        // gap 52: block 3, k-step 4
        if (!(abl & 8)) DVQ_SCORE(accP, 4, 6)
        if (!(abl & 32)) mfma_a<52>(accQ, bf[4]);
        DVQ_SB();
        DVQ_GSTAMP(52);
        // gap 53: block 3, k-step 5
        if (!(abl & 8)) DVQ_SCORE(accP, 6, 8)
        if (!(abl & 32)) mfma_a<53>(accQ, bf[5]);
        DVQ_SB();
        DVQ_GSTAMP(53);
        // gap 54: block 3, k-step 6
        if (!(abl & 8)) DVQ_SCORE(accP, 8, 10)
        if (!(abl & 32)) mfma_a<54>(accQ, bf[6]);
        DVQ_SB();
        DVQ_GSTAMP(54);
        // gap 55: block 3, k-step 7
        if (!(abl & 8)) DVQ_SCORE(accP, 10, 12)
        if (!(abl & 32)) mfma_a<55>(accQ, bf[7]);
        DVQ_SB();
        DVQ_GSTAMP(55);
        // gap 56: block 3, k-step 8
        if (!(abl & 8)) DVQ_SCORE(accP, 12, 14)
        if (!(abl & 32)) mfma_a<56>(accQ, bf[8]);
        DVQ_SB();
        DVQ_GSTAMP(56);
        // gap 57: block 3, k-step 9
        if (!(abl & 8)) DVQ_SCORE(accP, 14, 16)
        if (!(abl & 32)) mfma_a<57>(accQ, bf[9]);
        DVQ_SB();
        DVQ_GSTAMP(57);
        // gap 58: block 3, k-step 10
        write_slot<2>(c, t, m1, m2);
        if (!(abl & 32)) mfma_a<58>(accQ, bf[10]);
        DVQ_SB();
        DVQ_GSTAMP(58);
        // gap 59: block 3, k-step 11
        if (!(abl & 32)) mfma_a<59>(accQ, bf[11]);
        DVQ_SB();
        DVQ_GSTAMP(59);
        // gap 60: block 3, k-step 12
        if (!(abl & 32)) mfma_a<60>(accQ, bf[12]);
        DVQ_SB();
        DVQ_GSTAMP(60);
        // gap 61: block 3, k-step 13
        if (!(abl & 32)) mfma_a<61>(accQ, bf[13]);
        DVQ_SB();
        DVQ_GSTAMP(61);
        // gap 62: block 3, k-step 14
        if (!(abl & 32)) mfma_a<62>(accQ, bf[14]);
        DVQ_SB();
        DVQ_GSTAMP(62);
        // gap 63: block 3, k-step 15
        if (!(abl & 32)) mfma_a<63>(accQ, bf[15]);
        DVQ_SB();
        DVQ_GSTAMP(63);
        DVQ_STAMP(5);
    // >>> GENERATED
    };
    for (int t = 0; t < c.ntl; ++t) tile(t);
    wg_barrier();
    DVQ_T(t_loop);
    // ---- the last block's scores, the last two merges
    {
        float m1 = INFINITY, m2 = INFINITY;
        score<0, 16>(accQ, m1, m2);
        write_slot<3>(c, c.ntl - 1, m1, m2);
    }
    if (c.ntl >= 2) {
        mg.read(c, c.ntl - 2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mg.s0), "+v"(mg.s1), "+v"(mg.rs));
        mg.all(c, c.ntl - 2, rows_of(c, c.ntl - 2), und);
    }
    wg_barrier();
    mg.read(c, c.ntl - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mg.s0), "+v"(mg.s1), "+v"(mg.rs));
    mg.all(c, c.ntl - 1, rows_of(c, c.ntl - 1), und);
    if (c.lane == 0) reinterpret_cast<unsigned*>(lds + L_CNT)[4 + c.wave] = (unsigned)und;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- the undecided rows of all four lists -> candidate pairs / the all-entries list: 8 lanes per row, 32 rows per pass
    {
        const unsigned* cnt4 = reinterpret_cast<const unsigned*>(lds + L_CNT) + 4;
        const int n0 = (int)min(cnt4[0], 64u), n1_ = (int)min(cnt4[1], 64u), n2_ = (int)min(cnt4[2], 64u), n3_ = (int)min(cnt4[3], 64u);
        const int n_und = n0 + n1_ + n2_ + n3_;
        const bool overflow = cnt4[0] > 64u || cnt4[1] > 64u || cnt4[2] > 64u || cnt4[3] > 64u;   // a wave with > 64 undecided rows of its 64:
        (void)overflow;                                                                            // impossible (8 rows x 8 tiles), kept as a bound
        const uint16_t* ul = reinterpret_cast<const uint16_t*>(lds + L_UND);
        for (int base = 0; base < n_und; base += NT / 8) {
            const int k = base + (tid >> 3);
            const bool valid = k < n_und;
            int w = 0, kk = valid ? k : 0;
            if (kk >= n0) { kk -= n0; w = 1; if (kk >= n1_) { kk -= n1_; w = 2; if (kk >= n2_) { kk -= n2_; w = 3; } } }
            mg.expand(c, (unsigned)ul[w * 64 + kk], valid);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    DVQ_T(t_merge);

    // ---- refine: canonical distances of the listed pairs, four lanes per chain; key minimum per row
    unsigned long long* s_res = reinterpret_cast<unsigned long long*>(lds + L_RES);
    const unsigned* s_pair = reinterpret_cast<const unsigned*>(lds + L_PAIR);
    const uint16_t* s_slow = reinterpret_cast<const uint16_t*>(lds + L_SLOW);
    const unsigned* s_cnt = reinterpret_cast<const unsigned*>(lds + L_CNT);
    const int total = (int)min(s_cnt[0], (unsigned)PAIR_CAP);
    const int n_slow = (int)s_cnt[1];
    auto grow_of = [&](int rowslot) { return (c.tile0 + (long)blockIdx.x + (long)(rowslot >> 5) * c.G) * TILE + (rowslot & 31); };
    for (int s0 = 0; s0 < total; s0 += NT / 4) {
        if (s0 + c.wave * 16 < total) {                           // wave-uniform: this wave has at least one pair
            const int slot = s0 + (tid >> 2), q = tid & 3;
            const unsigned pr = slot < total ? s_pair[slot] : ~0u;
            const bool act = pr != ~0u;
            const int rowslot = (int)(pr >> 16), k = (int)(pr & 0xffffu);
            float zz, dot;
            chain_pair_x4(z + grow_of(rowslot) * D, E + (long)k * D, q, act, zz, dot);
            if (act && q == 0) {
                const float tsum = zz + ee_g[k];
                atomicMin(&s_res[rowslot], order_key(tsum - 2.0f * dot, k));
            }
        }
    }
    // what is left (NaN/Inf, fp16 overflow, invalid codebook image, overflowing list): all K entries canonically
    for (int o = 0; o < n_slow; ++o) {
        const int rowslot = s_slow[o];
        for (int k = tid; k < K; k += NT) {
            float zz2, dot2;
            chain_pair(z + grow_of(rowslot) * D, E + (long)k * D, zz2, dot2);
            const float tsum = zz2 + ee_g[k];
            atomicMin(&s_res[rowslot], order_key(tsum - 2.0f * dot2, k));
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
    if (dbg && c.lane == 0) {
        unsigned long long* o = dbg + 4096 + ((size_t)blockIdx.x * NWV + c.wave) * 8;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = seg[i];
    }
#endif
}

}  // namespace

// called by dvq_vq_argmin_fast (vq_stream.hip) after argument validation
int dvq_launch_vq_stream4(const float* z, const float* E, const void* packed, long M, int64_t* idx, unsigned long long* slow_rows,
                          unsigned long long* dbg, hipStream_t st) {
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            hipError_t e = hipFuncSetAttribute((const void*)&vq_stream4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
#ifdef DVQ_DIAG
            for (const void* fn : {(const void*)&vq_stream4_kernel<1>, (const void*)&vq_stream4_kernel<2>, (const void*)&vq_stream4_kernel<4>,
                                   (const void*)&vq_stream4_kernel<8>, (const void*)&vq_stream4_kernel<16>, (const void*)&vq_stream4_kernel<32>,
                                   (const void*)&vq_stream4_kernel<15>, (const void*)&vq_stream4_kernel<47>, (const void*)&vq_stream4_kernel<63>,
                                   (const void*)&vq_stream4_kernel<1024>, (const void*)&vq_stream4_kernel<2048>}) {
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
    int abl = 0;
#ifdef DVQ_DIAG
    if (const char* e = getenv("DVQ_VQ4_ABL")) abl = atoi(e);
#endif
    for (long t0 = 0; t0 < tiles; t0 += per_launch) {
        const long nt = (tiles - t0 < per_launch) ? tiles - t0 : per_launch;
        const unsigned grid = (unsigned)(nt < cus ? nt : cus);
        unsigned long long* dp = t0 == 0 ? dbg : (unsigned long long*)nullptr;
#define DVQ_GO(A) DVQ_LAUNCH(vq_stream4_kernel<A>, dim3(grid), dim3(NT), LDS_BYTES, st, z, E, M, t0, nt, (const char*)packed, idx, slow_rows, dp)
#ifdef DVQ_DIAG
        switch (abl) {
            case 1: DVQ_GO(1); break;
            case 2: DVQ_GO(2); break;
            case 4: DVQ_GO(4); break;
            case 8: DVQ_GO(8); break;
            case 16: DVQ_GO(16); break;
            case 32: DVQ_GO(32); break;
            case 15: DVQ_GO(15); break;
            case 47: DVQ_GO(47); break;
            case 63: DVQ_GO(63); break;
            case 1024: DVQ_GO(1024); break;
            case 2048: DVQ_GO(2048); break;
            default: DVQ_GO(0); break;
        }
#else
        (void)abl;
        DVQ_GO(0);
#endif
#undef DVQ_GO
        DVQ_CHECK_LAUNCH("vq_stream4");
    }
    return DVQ_OK;
}
