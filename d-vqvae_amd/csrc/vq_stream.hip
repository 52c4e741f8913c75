// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256): same indices as the exact kernel
// (vq.hip / oracle/vq_canonical.c), bit for bit, with z streamed from HBM exactly once UNDER the matrix work.
// Reference: VectorQuantizer.forward(z, istrain=False), network/vqvae/quantizer.py:46-49.
//
// Structure (persistent, codebook in registers, z through registers and one fp16 LDS tile):
//   * one 512-thread workgroup per CU walks over up to 8 tiles of 32 rows (tile = blockIdx + j * gridDim);
//   * wave w keeps codebook entries [64w, 64w+64) -- fp16 of -2 sE e_k, all 256 dims -- as the MFMA A operand in
//     128 VGPRs for the whole kernel (8 waves x 64 entries = the whole codebook, 256 KB of the CU's 512 KB of VGPRs);
//   * each wave owns 4 rows of every tile: it loads them HBM -> registers one tile ahead (16 lanes per row, 16 floats
//     per lane: 256-byte segments; the registers are refilled as soon as a tile is converted), converts them to fp16 ONCE (|h(z)|^2, the measured
//     rounding error |z - h(z)|, eps_row) and writes them to a padded fp16 tile in LDS (528-byte rows: conflict-free
//     ds_read_b128 fragments with one address register and immediate offsets);
//   * every wave multiplies the tile with its 64 entries: 16 fragment reads feed 32 v_mfma_f32_32x32x16_f16, the
//     accumulators START at sE |e_k|^2 (read from LDS straight into the accumulator registers), so they END as the
//     scores sE (|e_k|^2 - 2 z.e_k): no per-score arithmetic beyond id packing and a lane-local top-2 (3 VALU per score);
//   * the loop is software-pipelined by hand: the 32 MFMA gaps of tile t carry the scores of tile t-1 / t, the merge of
//     tile t-2, the conversion of tile t+1 and the loads of tile t+2; one s_barrier per tile;
//   * per tile, 16 (wave, lane-half) slots per row hold (min, second) packed scores; the merge finds the row minimum,
//     the slots within eps_row of it, and either decides the row or appends (row, entry) pairs to a list;
//   * after the last tile: the canonical fp32 evaluation d_k = (zz + ee_k) - 2 dot_k (k-ordered fmaf chains, four lanes
//     per chain) of the listed pairs, ds_min_u64 of (ordered distance bits, entry) per row = torch.argmin's order
//     (first minimum, NaN first).  Rows with NaN/Inf, fp16 overflow or an overflowing list: all K entries canonically.
//
// Error bound (filter keeps the exact winner).  No per-row scaling: h(.) = round to nearest fp16, subnormals kept (by the
// conversion and, measured on gfx950, by the matrix core: tests/test_gpu_parity.py::test_mfma_keeps_f16_subnormals).
// With dz_j = z_j - h(z_j), de_kj = e_kj - h(-2 sE e_kj)/(-2 sE) (exact in fp32), products h.h exact in the MFMA's fp32,
// T_k = sE (true_k - |z|^2), gamma_n = n 2^-24 (gamma'_n = n 2^-23 allows a truncating accumulator):
//   |S_k - T_k|         <= sE [2 (|dz||e_k| + |z||de_k| + |dz||de_k|) + gamma'_259 (|e_k|^2 + 2|z||e_k|)]          (filter)
//   |d_k - true_k|      <= gamma_260 (|z| + |e_k|)^2                                                            (exact side)
//   |packed(S_k) - S_k| <= 2^-18 |S_k| <= 2^-18 sE (|z| + |e_k|)^2                                 (5-bit id in the mantissa)
// so for the canonical winner k* and every k:  packed(S_k*) <= packed(S_k) + sE eps_row,
//   eps_row = 4 (|dz| Emax + |z| dEmax + |dz| dEmax) + (2 gamma'_259 + 2^-17 + 2 gamma_260 = 9.97e-5) (|z| + Emax)^2,
// |dz| measured by the kernel's own conversion (-DDVQ_MEASURE_DZ=0: half-ulp bound 2^-11 |h(z)| + 2^-21), |z| <= |h(z)| + |dz|, dEmax measured by
// dvq_vq_pack, norms rounded up.  A slot whose
// SECOND score is within eps may hide a third: all 32 entries of that slot are listed, so the list always contains k*.
// Algorithmic HBM bytes per row: D*4 (z) + 8 (int64 index); the codebook (K*D*4) is read once.
//
// Hand-issued LDS reads (inline asm + counted s_waitcnt) carry three rules, each learnt from a wrong result:
//   (1) the destination of a read is "defined" for the compiler at the asm statement: it must not be copied or reused before
//       the wait -- so no such read is in flight across a control-flow join (the loop is rotated: joins sit behind the
//       barrier's lgkmcnt(0)), and a read whose value is never used is still pinned by the wait that completes it;
//   (2) vector work placed in an MFMA gap is anchored there by an empty asm that uses its result (else the compiler
//       sinks it to the end of the loop body, behind every MFMA);
//   (3) __builtin_bit_cast applied directly to a vector-element expression reads element 0: copy the element out first.
#include "dvq_internal.h"
#include "vq_pack.h"
#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int K = 512, D = 256;
constexpr int NWV = 8;                             // waves per workgroup = 64-entry slices of the codebook
constexpr int NT = 64 * NWV;                       // 512 threads, one workgroup per CU
constexpr int TILE = 32;                           // rows per tile
constexpr int MAX_TILES = 8;                       // tiles per workgroup per launch (row-result table, slow list)
constexpr int EXP_LIMIT = 40;                      // |log2(max codebook magnitude)| beyond this -> exact fallback

constexpr int Z16_ROW = 528;                       // padded fp16 row: 512 B + 16 B (shifts consecutive rows by one bank group)
constexpr int Z16_BUF = TILE * Z16_ROW;            // 16 896
constexpr int MS_ROW = 264;                        // merge slots of a row: 16 (wave, lane half) x 2 (accumulator) x 8 B + 8 B pad
constexpr int MS_BUF = TILE * MS_ROW;              // 8 448
constexpr int PAIR_CAP = 2048;

constexpr int L_Z16 = 0;                                   // 2 x fp16 tile
constexpr int L_MS = L_Z16 + 2 * Z16_BUF;                  // 3 x merge slots (tile % 3)
constexpr int L_RS = L_MS + 3 * MS_BUF;                    // 4 x [32] {eps sE, flag}
constexpr int L_EES = L_RS + 4 * TILE * 8;                 // [K] f32: sE |e_k|^2 (accumulator start values)
constexpr int L_RES = L_EES + K * 4;                       // [MAX_TILES*32] u64 row results (ordered distance bits : entry)
constexpr int L_PAIR = L_RES + MAX_TILES * TILE * 8;       // [PAIR_CAP] u32 (rowslot << 16 | entry)
constexpr int L_SLOW = L_PAIR + PAIR_CAP * 4;              // [MAX_TILES*32] u16 rowslots for the all-entries path
constexpr int L_CNT = L_SLOW + MAX_TILES * TILE * 2;       // [0] pairs, [1] rows for the all-entries path, [2] rows with >= 32 pairs
constexpr int L_DBG = L_CNT + 64;                          // [8 waves][32] u32 phase stamps (DVQ_VQ_DBG only)
constexpr int LDS_BYTES = L_DBG + NWV * 32 * 4;
constexpr int DBG_WG_BYTES = 64 + NWV * 32 * 4;            // per-workgroup debug record in the workspace
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// ------------------------------------------------------------------------------------------------ pack
// A codebook of Kr < K entries (Kr a multiple of 32; round 6: the model's K = 128 codebooks) is PADDED to the kernels' K = 512: the
// padding entries have a zero image and 3e38 as |e|^2 (their filter scores are 3e38, never within eps of a real one; the canonical
// paths stop at Kr); hdr->K = Kr.
__global__ void vq_pack_norm_kernel(const float* __restrict__ E, int Kr, float* __restrict__ ee, PackHeader* hdr) {
    __shared__ float red[K];
    __shared__ float redm[K];
    const int k = threadIdx.x;                                   // blockDim = K
    const float* p = E + (k < Kr ? k : 0) * D;
    float acc = 0.f, mx = 0.f;
    bool finite = true;
    for (int j = 0; j < D; ++j) {
        acc = fmaf(p[j], p[j], acc);                             // canonical chain (same as rownorm_kernel)
        finite = finite && (fabsf(p[j]) <= 3.0e38f);
        mx = fmaxf(mx, fabsf(p[j]));
    }
    ee[k] = k < Kr ? acc : 3.0e38f;
    red[k] = k >= Kr ? 0.f : (finite && acc <= 3.0e38f) ? acc : INFINITY;
    redm[k] = k >= Kr ? 0.f : finite ? mx : INFINITY;
    __syncthreads();
    for (int o = K / 2; o > 0; o >>= 1) {
        if (k < o) {
            red[k] = fmaxf(red[k], red[k + o]);
            redm[k] = fmaxf(redm[k], redm[k + o]);
        }
        __syncthreads();
    }
    if (k == 0) {
        hdr->emax = sqrtf(red[0]) * 1.00001f;
        const float m2 = 2.0f * redm[0];                         // the image holds -2 e
        const int e = (int)((__float_as_uint(m2) >> 23) & 0xff) - 127;
        const bool ok = redm[0] > 0.f && e >= -EXP_LIMIT && e <= EXP_LIMIT && red[0] <= 3.0e38f;
        hdr->sexp = ok ? 13 - e : 0;
        hdr->valid = ok ? 1 : 0;
        hdr->K = Kr;
        hdr->D = D;
        hdr->layout = 3;
    }
}

// image: fp16 of -2 sE e in fragment order
__global__ void vq_pack_img_kernel(const float* __restrict__ E, int Kr, const PackHeader* __restrict__ hdr, _Float16* __restrict__ img) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= K * D) return;
    img[img_pos(gid / D, gid % D)] = gid < Kr * D ? (_Float16)(-2.0f * pow2f(hdr->sexp) * E[gid]) : (_Float16)0.0f;   // (subnormals kept; the error kernel measures what is stored)
}

// measured rounding error of the image, per entry, as a 2-norm in codebook units; its maximum goes into the header
__global__ void vq_pack_err_kernel(const float* __restrict__ E, int Kr, const _Float16* __restrict__ img, PackHeader* hdr) {
    __shared__ float red[K];
    const int k = threadIdx.x;                                   // blockDim = K
    const float m2s = -2.0f * pow2f(hdr->sexp);
    float acc = 0.f;
    for (int j = 0; j < D && k < Kr; ++j) {
        const float sv = m2s * E[k * D + j];                     // exact (power-of-two scale, range checked by `valid`)
        const float d = sv - (float)img[img_pos(k, j)];          // exact: fp32 values at most 11 significant bits apart (or sv itself)
        acc = fmaf(d, d, acc);
    }
    red[k] = acc;
    __syncthreads();
    for (int o = K / 2; o > 0; o >>= 1) {
        if (k < o) red[k] = fmaxf(red[k], red[k + o]);
        __syncthreads();
    }
    if (k == 0) hdr->demax = hdr->valid ? sqrtf(red[0]) / fabsf(m2s) * 1.0001f : INFINITY;
}

// ------------------------------------------------------------------------------------------------ LDS access by hand
// The tile loop's LDS reads are inline asm with counted waits (the compiler's own waits are lgkmcnt(0) at every use: it
// cannot keep four fragment reads in flight behind the one an MFMA needs).  The "+v" operands of a wait pin the consumers
// below it.  See the three rules in the file header.
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
__device__ __forceinline__ void wg_barrier() {                    // LDS operations of this wave done, then the workgroup barrier; VMEM stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16-lane all-reduce by DPP (xor 1, xor 2, half mirror, mirror): every lane of a row of 16 gets the result
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}
// one instruction (fminf adds a canonicalising v_max); in-range scores are far above -3e38
__device__ __forceinline__ float min_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -3.0e38f); }

// entry of accumulator register i (0..15) of entry tile jn, lane half h, wave w (v_mfma_f32_32x32x16 D layout)
__device__ __forceinline__ int entry_of(int w, int jn, int i, int h) { return 64 * w + 32 * jn + 8 * (i >> 2) + 4 * h + (i & 3); }

// torch.argmin order as one unsigned key: smaller distance first, equal distances -> lower entry, NaN before everything
__device__ __forceinline__ unsigned long long order_key(float d, int k) {
    const unsigned b = __float_as_uint(d);
    const unsigned u = (d != d) ? 0u : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    return ((unsigned long long)u << 32) | (unsigned)k;
}

// ------------------------------------------------------------------------------------------------ refine chains
// Canonical chains threaded through 4 lanes: lane q of a group holds floats [64q, 64q+64) of its z row and of its
// candidate's codebook row (all loads issued up front: ONE memory latency), then the k-ordered fmaf chain runs as four
// 64-step rounds, round q continuing from the accumulator lane q-1 produced.  Bit-identical to a single 256-step chain.
__device__ __forceinline__ void chain_pair_x4(const float* __restrict__ zr, const float* __restrict__ er, int q, bool active,
                                              float& zz, float& dot) {
    f32x4 x[16], y[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {                              // idle lanes load nothing (a shared dummy row would hot-spot one L2 channel)
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

// full-row single-lane form (all-entries fallback)
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
struct Ctx {
    const float* z;
    long M;
    long tile0;          // first tile of this launch
    int G;               // workgroups (tile stride)
    int ntl;             // tiles of this workgroup
    int wave, lane;
    unsigned lds0;       // LDS byte address of the dynamic segment
    float emax, demax, sEf;
    bool e_valid;
};

// shader-clock stamp `slot` of this wave (tiles 2 and 5 only: slots 0-15 / 16-31); DVQ_VQ_DBG instantiation only
__device__ __forceinline__ void stamp(const Ctx& c, int t, int slot) {
    if (t == 2 || t == 5) {
        const unsigned v = (unsigned)__builtin_amdgcn_s_memtime();
        if (c.lane == 0) ds_wr32(c.lds0 + L_DBG + (c.wave * 32 + (t == 5 ? 16 : 0) + slot) * 4, v);
    }
}

// this wave's 4 rows of local tile j, HBM -> registers: lane (g = lane / 16, i = lane % 16) takes floats 4 (i + 16 q) .. +3 of
// row 4 wave + g, q = 0..3 (per instruction four 256-byte segments); rows behind the end of the data repeat the last row
__device__ __forceinline__ void load_rows(const Ctx& c, int j, f32x4 (&x)[4]) {
    const long tile = c.tile0 + (long)blockIdx.x + (long)j * c.G;
    long gr = tile * TILE + c.wave * 4 + (c.lane >> 4);
    if (gr >= c.M) gr = c.M - 1;
    const f32x4* p = reinterpret_cast<const f32x4*>(c.z + gr * D) + (c.lane & 15);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = __builtin_nontemporal_load(p + 16 * q);
}

// scores of accumulator registers [B, E): id = idbase + register index in the low 5 mantissa bits, top-2 update
template <int B, int E>
__device__ __forceinline__ void score(const f32x16& a, int idbase, float& m1, float& m2) {
#pragma unroll
    for (int i = B; i < E; ++i) {
        const float p = __uint_as_float((__float_as_uint(a[i]) & ~31u) | (unsigned)(idbase + i));
        m2 = __builtin_amdgcn_fmed3f(m1, m2, p);
        m1 = min_nc(m1, p);
    }
}

// (min, second) of one accumulator's 16 scores -> slot (wave, lane half, JN) of the lane's row, tile t
template <int JN>
__device__ __forceinline__ void write_slot(const Ctx& c, int t, float m1, float m2) {
    f32x2 v;
    v[0] = m1;
    v[1] = m2;
    ds_wr64<8 * JN>(c.lds0 + L_MS + ((t + 3) % 3) * MS_BUF + (c.lane & 31) * MS_ROW + (2 * c.wave + (c.lane >> 5)) * 16, v);
}

// d = h - x with h the low (HI = 0) or high (HI = 1) fp16 half of `hp`: one v_fma_mix_f32 (exact; the sign does not matter)
template <int HI>
__device__ __forceinline__ float mix_diff(float hp, float x) {
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    return d;
}

// fp32 -> fp16 conversion of the wave's 4 rows of a tile in pieces (the pipelined loop puts one piece into each MFMA gap)
#ifndef DVQ_MEASURE_DZ
#define DVQ_MEASURE_DZ 1   // 1: |z - h(z)| measured element by element (31 vector instructions per wave and tile; eps_row 0.65x, 0.65x the pairs in the refine tail: measured 36.0 vs 37.3 us per call); 0: half-ulp bound
#endif
struct Convert {
    float hh, dsq;                                                     // sum h(z)^2, sum (z - h(z))^2
    f32x2 pk;
    __device__ __forceinline__ void start() { hh = 0.f; dsq = 0.f; }
    template <int Q>
    __device__ __forceinline__ void cvt(const Ctx& c, int j, const f32x4 (&x)[4]) {   // quad Q: convert, store into the fp16 image of tile j
        const int g = c.lane >> 4, i = c.lane & 15;
        f32x2 a, b;
        a[0] = x[Q][0]; a[1] = x[Q][1]; b[0] = x[Q][2]; b[1] = x[Q][3];
        const f16x2 lo = __builtin_convertvector(a, f16x2), hi = __builtin_convertvector(b, f16x2);
        hh = __builtin_amdgcn_fdot2(lo, lo, hh, false);
        hh = __builtin_amdgcn_fdot2(hi, hi, hh, false);
        pk[0] = __builtin_bit_cast(float, lo);
        pk[1] = __builtin_bit_cast(float, hi);
        ds_wr64<128 * Q>(c.lds0 + L_Z16 + (j & 1) * Z16_BUF + (c.wave * 4 + g) * Z16_ROW + 8 * i, pk);
    }
    template <int Q>
    __device__ __forceinline__ void err(float lo_hi0, float lo_hi1, const f32x4 (&x)[4]) {   // rounding error of quad Q (pk of that quad)
        if (DVQ_MEASURE_DZ) {
            const float d0 = mix_diff<0>(lo_hi0, x[Q][0]), d1 = mix_diff<1>(lo_hi0, x[Q][1]);
            const float d2 = mix_diff<0>(lo_hi1, x[Q][2]), d3 = mix_diff<1>(lo_hi1, x[Q][3]);
            dsq = fmaf(d0, d0, dsq); dsq = fmaf(d1, d1, dsq); dsq = fmaf(d2, d2, dsq); dsq = fmaf(d3, d3, dsq);
        }
    }
    __device__ __forceinline__ void finish(const Ctx& c, int j) {        // hh, dsq already reduced over the row's 16 lanes
        const int g = c.lane >> 4, i = c.lane & 15;
        const float hn = __builtin_amdgcn_sqrtf(hh);
        // |z - h(z)|: measured, or a priori: half an ulp of a normal fp16 is at most 2^-11 |h|, of a subnormal one 2^-25
        // (16 * 2^-25 over a row); fp16 subnormals are KEPT by the conversion (MODE) and by the matrix core (measured on gfx950:
        // tools/microbench/f16_denorm_probe.hip, asserted by tests/test_gpu_parity.py::test_mfma_keeps_f16_subnormals)
        const float dzn = DVQ_MEASURE_DZ ? __builtin_amdgcn_sqrtf(dsq) * 1.0001f : hn * 4.8877e-4f + 4.8e-7f;
        const float zn = (hn + dzn) * 1.0001f;                          // |z| <= |h(z)| + |z - h(z)|
        const float eps = 4.004f * (dzn * c.emax + zn * c.demax + dzn * c.demax) + 1.01e-4f * (zn + c.emax) * (zn + c.emax);
        const float epsS = eps * c.sEf;
        const bool bad = !c.e_valid || !(hh <= 3.0e38f) || !(dsq <= 3.0e38f) || !(epsS <= 3.0e38f);   // NaN/Inf, fp16 overflow
        if (i == 0) {
            f32x2 rs;
            rs[0] = epsS;
            rs[1] = __uint_as_float(bad ? 1u : 0u);
            ds_wr64<0>(c.lds0 + L_RS + (j & 3) * (TILE * 8) + (c.wave * 4 + g) * 8, rs);
        }
    }
};

// merge of a tile in stages: this wave's rows 4 wave + g, 16 lanes per row, lane i looks at the two slots (accumulator 0 / 1)
// of (source wave i / 2, lane half i % 2), i.e. at 2 x 16 codebook entries
struct Merge {
    f32x4 sl;                                                           // (min, second) of accumulator 0, (min, second) of accumulator 1
    f32x2 rs;
    float rmin;                                                         // row minimum, then the row's threshold
    unsigned long long bA1, bA2, bB1, bB2;                              // wave ballots (uniform: SGPRs) of "within eps of the row minimum"
    __device__ __forceinline__ void read(const Ctx& c, int tm) {
        const int g = c.lane >> 4, i = c.lane & 15;
        const int r = 4 * c.wave + g;
        ds_rd128<0>(sl, c.lds0 + L_MS + ((tm + 3) % 3) * MS_BUF + r * MS_ROW + i * 16);
        ds_rd64<0>(rs, c.lds0 + L_RS + (tm & 3) * (TILE * 8) + r * 8);
    }
    __device__ __forceinline__ void min_a() {                            // after the wait that pins sl, rs
        rmin = min_nc(sl[0], sl[2]);
        rmin = min_nc(rmin, dpp_f<0xB1>(rmin));
        rmin = min_nc(rmin, dpp_f<0x4E>(rmin));
    }
    __device__ __forceinline__ void min_b(const Ctx& c) {
        rmin = min_nc(rmin, dpp_f<0x141>(rmin));
        rmin = min_nc(rmin, dpp_f<0x140>(rmin));
        rmin = rmin + rs[0];
        bA1 = __ballot(sl[0] <= rmin);
        bA2 = __ballot(sl[1] <= rmin);
        bB1 = __ballot(sl[2] <= rmin);
        bB2 = __ballot(sl[3] <= rmin);
    }
    __device__ __forceinline__ void act(const Ctx& c, int tm) {
        const int g = c.lane >> 4, i = c.lane & 15;
        const int r = 4 * c.wave + g;
        const unsigned sA1 = (unsigned)(bA1 >> (16 * g)) & 0xffffu, sA2 = (unsigned)(bA2 >> (16 * g)) & 0xffffu;   // this row's 16 lanes
        const unsigned sB1 = (unsigned)(bB1 >> (16 * g)) & 0xffffu, sB2 = (unsigned)(bB2 >> (16 * g)) & 0xffffu;
        const bool cA1 = (sA1 >> i) & 1u, cA2 = (sA2 >> i) & 1u, cB1 = (sB1 >> i) & 1u, cB2 = (sB2 >> i) & 1u;
        const int n1 = __popc(sA1) + __popc(sB1), n2 = __popc(sA2) + __popc(sB2);
        const bool bad = __float_as_uint(rs[1]) != 0u;
        const long grow = (c.tile0 + (long)blockIdx.x + (long)tm * c.G) * TILE + r;
        const bool live = grow < c.M && tm >= 0;                        // (the pipelined loop also merges "tiles" -2, -1)
        const unsigned rowslot = (unsigned)(tm * TILE + r);
        const int w_src = i >> 1, h_src = i & 1;
        const bool slow = bad || n1 == 0;
        const bool unique = !slow && n1 == 1 && n2 == 0;
        const bool amb = live && !slow && !unique;
        if (live && unique && (cA1 || cB1)) {
            const unsigned id = __float_as_uint(cA1 ? sl[0] : sl[2]) & 15u;
            f32x2 kv;
            kv[0] = __uint_as_float((unsigned)entry_of(w_src, cA1 ? 0 : 1, (int)id, h_src));
            kv[1] = __uint_as_float(0u);
            ds_wr64<0>(c.lds0 + L_RES + rowslot * 8, kv);
        }
        if (__ballot(live && !unique) == 0ull) return;                  // common case: every row of the wave decided
        // one reservation per ambiguous row (leader lane i == 0): a slot with one score within eps takes one pair, a slot
        // whose second score is within eps too (it may hide a third) takes all its 16 entries
        const int need = (n1 - n2) + 16 * n2;
        unsigned pos = 0;
        if (amb && i == 0) pos = ds_add_rtn(c.lds0 + L_CNT, (unsigned)need);
        pos = (unsigned)__builtin_amdgcn_readlane((int)pos, 0) * (g == 0) + (unsigned)__builtin_amdgcn_readlane((int)pos, 16) * (g == 1) +
              (unsigned)__builtin_amdgcn_readlane((int)pos, 32) * (g == 2) + (unsigned)__builtin_amdgcn_readlane((int)pos, 48) * (g == 3);
        const bool fits = pos + (unsigned)need <= (unsigned)PAIR_CAP;
        if (amb && i == 0 && fits && need >= 32) asm volatile("ds_add_u32 %0, %1" ::"v"(c.lds0 + L_CNT + 8), "v"(1u) : "memory");   // heavily ambiguous row
        if (amb && fits && (cA1 || cB1)) {
            const unsigned lt = (1u << i) - 1u;
            unsigned off = pos + (unsigned)(__popc(sA1 & ~sA2 & lt) + __popc(sB1 & ~sB2 & lt)) + 16u * (unsigned)(__popc(sA2 & lt) + __popc(sB2 & lt));
            auto push = [&](int jn, bool one, bool all, float best) {
                if (all) {
                    for (int e = 0; e < 16; ++e) ds_wr32(c.lds0 + L_PAIR + (off + e) * 4, (rowslot << 16) | (unsigned)entry_of(w_src, jn, e, h_src));
                    off += 16;
                } else if (one) {
                    ds_wr32(c.lds0 + L_PAIR + off * 4, (rowslot << 16) | (unsigned)entry_of(w_src, jn, (int)(__float_as_uint(best) & 15u), h_src));
                    off += 1;
                }
            };
            push(0, cA1, cA2, sl[0]);
            push(1, cB1, cB2, sl[2]);
        }
        if (live && i == 0 && (slow || (amb && !fits))) {
            const unsigned sp = ds_add_rtn(c.lds0 + L_CNT + 4, 1u);
            ds_wr16(c.lds0 + L_SLOW + sp * 2, rowslot);
        }
    }
};

// ABL: timing-only ablations for diagnostics (DVQ_VQ_ABL; results invalid unless 0): 1 no row loads, 2 no conversion,
// 4 no merge, 8 no scoring, 16 no MFMA, 32 no barrier in the loop.  DBG: phase stamps (DVQ_VQ_DBG).
template <int ABL, bool DBG>
__global__ __launch_bounds__(NT, 2) void vq_stream_kernel(const float* __restrict__ z, const float* __restrict__ E, long M,
                                                          long tile0, long n_tiles, const char* __restrict__ packed,
                                                          int64_t* __restrict__ idx, unsigned long long* __restrict__ slow_rows,
                                                          unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x;
    Ctx c;
    c.z = z;
    c.M = M;
    c.tile0 = tile0;
    c.G = (int)gridDim.x;
    c.ntl = (int)((n_tiles - (long)blockIdx.x + c.G - 1) / c.G);          // blockIdx.x < n_tiles by construction
    c.lane = tid & 63;
    c.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    c.lds0 = lds_addr(lds);
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    // fp16 subnormals are kept by conversions (MODE.FP_DENORM[3:2] = 3, stated explicitly) as they are by the matrix core
    __builtin_amdgcn_s_setreg((2 - 1) << 11 | 6 << 6 | 1, 3);            // hwreg(HW_REG_MODE, offset 6, width 2) <- 3

    // ---- prologue: header and |e|^2 (small, needed first), the first two tiles, then the codebook slice (needed last)
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    c.emax = hdr->emax;
    c.demax = hdr->demax;
    c.e_valid = hdr->valid != 0;
    c.sEf = c.e_valid ? pow2f(hdr->sexp) : 1.0f;
    const float ee_mine = ee_g[tid];                                   // NT == K
    f32x4 x[4];                                                        // fp32 rows of the tile in flight (one tile ahead)
    if (!(ABL & 1)) load_rows(c, 0, x);
    f16x8 af[2][16];
    {
        const f16x8* img = reinterpret_cast<const f16x8*>(packed + PK_OFF_IMG) + (size_t)c.wave * (32 * 64) + c.lane;
#pragma unroll
        for (int f = 0; f < 32; ++f) af[f >> 4][f & 15] = img[f * 64];
    }
    for (int u = tid; u < MAX_TILES * TILE; u += NT) reinterpret_cast<unsigned long long*>(lds + L_RES)[u] = ~0ull;
    for (int u = tid; u < PAIR_CAP; u += NT) reinterpret_cast<unsigned*>(lds + L_PAIR)[u] = ~0u;   // holes of refused reservations stay invalid
    if (tid < 3) reinterpret_cast<unsigned*>(lds + L_CNT)[tid] = 0u;
    reinterpret_cast<float*>(lds + L_EES)[tid] = ee_mine * c.sEf;
    Convert cv;
    unsigned long long t0a = 0, t0b = 0;
    {
        if (DBG) { asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])); t0a = __builtin_amdgcn_s_memrealtime(); }
        cv.start();
        cv.cvt<0>(c, 0, x); const f32x2 p0 = cv.pk; cv.cvt<1>(c, 0, x); const f32x2 p1 = cv.pk;
        cv.cvt<2>(c, 0, x); const f32x2 p2 = cv.pk; cv.cvt<3>(c, 0, x); const f32x2 p3 = cv.pk;
        cv.err<0>(p0[0], p0[1], x); cv.err<1>(p1[0], p1[1], x); cv.err<2>(p2[0], p2[1], x); cv.err<3>(p3[0], p3[1], x);
        cv.hh = row16_sum(cv.hh);
        cv.dsq = row16_sum(cv.dsq);
        cv.finish(c, 0);
        if (c.ntl > 1 && !(ABL & 1)) load_rows(c, 1, x);
        if (DBG) t0b = __builtin_amdgcn_s_memrealtime();
    }
    wg_barrier();          // (not __syncthreads: its fence would wait for the codebook slice and the rows of tile 1 as well)
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();

    // ---- tile loop, software-pipelined by hand: 32 MFMA gaps per tile, the vector work of four other tiles spread EVENLY
    // over them (the vector port of a SIMD, shared by its two waves, is the scarce resource: a gap holds 6-9 vector
    // instructions).  The two accumulators run one behind the other so that each is free for many gaps while it is scored:
    //   B      barrier: the fp16 image of tile t is complete, nobody reads image t-1 any more
    //   G0-7   acc0 <- k-steps 0-7 (starting from sE|e|^2)   | 16 scores of acc1 (tile t-1) -> its slots; acc1's start values
    //   G8-15  acc1 <- k-steps 0-7 (fragments read again)    | merge of tile t-2; conversion of tile t+1 begins
    //   G16-23 acc0 <- k-steps 8-15                          | conversion of tile t+1, row statistics; loads of tile t+2
    //   G24-31 acc1 <- k-steps 8-15 (fragments read again)   | 16 scores of acc0 (tile t) -> its slots
    // The body is generated (tools/gen_vq_tile.py: gap table -> code, with the counted waits derived from the program order of
    // the reads that are ALWAYS issued; other LDS operations in between only make a wait longer, never shorter).
    // Nothing is in flight at the end of a tile: the loop header is a control-flow join (rule 1 of the file header).
    // Nothing is conditional on t except the row loads: tile "-1" is scored, tiles "-2", "-1" are merged (not live: no
    // effect) and the tile behind the last one is converted (from stale registers, never used).
    f32x16 acc0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    f16x8 bf[16];
    f32x4 ci0[4], ci1[4];                                               // start values: read into the (just scored, free) accumulator's registers
    const int r_l = c.lane & 31, h_l = c.lane >> 5;
    const unsigned ea = c.lds0 + L_EES + (64 * c.wave + 4 * h_l) * 4;
    const unsigned zbase = c.lds0 + L_Z16 + r_l * Z16_ROW + 16 * h_l;
    Merge mg;
#define DVQ_MF0(S) if (!(ABL & 16)) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][S], bf[S], acc0, 0, 0, 0)
#define DVQ_MF1(S) if (!(ABL & 16)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][S], bf[S], acc1, 0, 0, 0)
#define DVQ_SB() __builtin_amdgcn_sched_barrier(0)
#define DVQ_PIN2(A, B) asm volatile("" : "+v"(A), "+v"(B))
#define DVQ_RDF(S) ds_rd128<32 * (S)>(bf[S], za)
#define DVQ_STAMP(SLOT) if (DBG) stamp(c, t, SLOT)
#define DVQ_SCORE(ACC, B, E) if (!(ABL & 8)) { score<B, E>(ACC, 0, m1, m2); DVQ_PIN2(m1, m2); }
    auto tile = [&](const int t) __attribute__((always_inline)) {
    // <<< GENERATED by tools/gen_vq_tile.py
        const bool do_load = !(ABL & 1) && t + 2 < c.ntl;
        const unsigned za = zbase + (t & 1) * Z16_BUF;
        DVQ_STAMP(0);
        if (ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        DVQ_STAMP(1);
        ds_rd128<0>(ci0[0], ea); ds_rd128<32>(ci0[1], ea); ds_rd128<64>(ci0[2], ea); ds_rd128<96>(ci0[3], ea);
        DVQ_RDF(0); DVQ_RDF(1); DVQ_RDF(2);
        float m1 = INFINITY, m2 = INFINITY;                               // (min, second) of one accumulator's 16 scores
        f32x2 pk0, pk1, pk2, pk3;
        // gap 0: acc0, k-step 0
        DVQ_SCORE(acc1, 0, 3)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ci0[0]), "+v"(ci0[1]), "+v"(ci0[2]), "+v"(ci0[3]), "+v"(bf[0]));
        {
            f32x16 st;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[4 * q + e] = ci0[q][e];
            if (!(ABL & 16)) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bf[0], st, 0, 0, 0);
        }
        DVQ_RDF(3);
        DVQ_SB();
        // gap 1: acc0, k-step 1
        DVQ_SCORE(acc1, 3, 6)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[1]));
        DVQ_MF0(1);
        DVQ_RDF(4);
        DVQ_SB();
        // gap 2: acc0, k-step 2
        DVQ_SCORE(acc1, 6, 9)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[2]));
        DVQ_MF0(2);
        DVQ_RDF(5);
        DVQ_SB();
        // gap 3: acc0, k-step 3
        DVQ_SCORE(acc1, 9, 12)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[3]));
        DVQ_MF0(3);
        DVQ_RDF(6);
        DVQ_SB();
        // gap 4: acc0, k-step 4
        DVQ_SCORE(acc1, 12, 14)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[4]));
        DVQ_MF0(4);
        DVQ_RDF(7);
        DVQ_SB();
        // gap 5: acc0, k-step 5
        if (!(ABL & 8)) { score<14, 16>(acc1, 0, m1, m2); write_slot<1>(c, t - 1, m1, m2); }
        ds_rd128<128>(ci1[0], ea); ds_rd128<160>(ci1[1], ea); ds_rd128<192>(ci1[2], ea); ds_rd128<224>(ci1[3], ea);
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bf[5]));
        DVQ_MF0(5);
        DVQ_RDF(0);
        DVQ_SB();
        // gap 6: acc0, k-step 6
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bf[6]));
        DVQ_MF0(6);
        DVQ_RDF(1);
        DVQ_SB();
        // gap 7: acc0, k-step 7
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bf[7]));
        DVQ_MF0(7);
        DVQ_RDF(2);
        DVQ_SB();
        DVQ_STAMP(2);
        // gap 8: acc1, k-step 0
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ci1[0]), "+v"(ci1[1]), "+v"(ci1[2]), "+v"(ci1[3]), "+v"(bf[0]));
        {
            f32x16 st;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[4 * q + e] = ci1[q][e];
            if (!(ABL & 16)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][0], bf[0], st, 0, 0, 0);
        }
        DVQ_RDF(3);
        if (!(ABL & 4)) mg.read(c, t - 2);
        DVQ_SB();
        // gap 9: acc1, k-step 1
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[1]));
        DVQ_MF1(1);
        DVQ_RDF(4);
        DVQ_SB();
        // gap 10: acc1, k-step 2
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[2]));
        DVQ_MF1(2);
        DVQ_RDF(5);
        if (!(ABL & 4)) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(mg.sl), "+v"(mg.rs));
        if (!(ABL & 4)) { mg.min_a(); asm volatile("" : "+v"(mg.rmin)); }
        DVQ_SB();
        // gap 11: acc1, k-step 3
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[3]));
        DVQ_MF1(3);
        DVQ_RDF(6);
        if (!(ABL & 4)) { mg.min_b(c); asm volatile("" : "+s"(mg.bA1), "+s"(mg.bB1)); }
        DVQ_SB();
        // gap 12: acc1, k-step 4
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[4]));
        DVQ_MF1(4);
        DVQ_RDF(7);
        if (!(ABL & 4)) mg.act(c, t - 2);
        DVQ_SB();
        // gap 13: acc1, k-step 5
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[5]));
        DVQ_MF1(5);
        DVQ_RDF(8);
        if (!(ABL & 2)) { cv.start(); cv.template cvt<0>(c, t + 1, x); pk0 = cv.pk; }
        DVQ_SB();
        // gap 14: acc1, k-step 6
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[6]));
        DVQ_MF1(6);
        DVQ_RDF(9);
        if (!(ABL & 2)) { cv.template err<0>(pk0[0], pk0[1], x); DVQ_PIN2(cv.hh, cv.dsq); }
        DVQ_SB();
        // gap 15: acc1, k-step 7
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[7]));
        DVQ_MF1(7);
        DVQ_RDF(10);
        if (!(ABL & 2)) { cv.template cvt<1>(c, t + 1, x); pk1 = cv.pk; }
        DVQ_SB();
        DVQ_STAMP(3);
        // gap 16: acc0, k-step 8
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[8]));
        DVQ_MF0(8);
        DVQ_RDF(11);
        if (!(ABL & 2)) { cv.template err<1>(pk1[0], pk1[1], x); DVQ_PIN2(cv.hh, cv.dsq); }
        DVQ_SB();
        // gap 17: acc0, k-step 9
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[9]));
        DVQ_MF0(9);
        DVQ_RDF(12);
        if (!(ABL & 2)) { cv.template cvt<2>(c, t + 1, x); pk2 = cv.pk; }
        DVQ_SB();
        // gap 18: acc0, k-step 10
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[10]));
        DVQ_MF0(10);
        DVQ_RDF(13);
        if (!(ABL & 2)) { cv.template err<2>(pk2[0], pk2[1], x); DVQ_PIN2(cv.hh, cv.dsq); }
        DVQ_SB();
        // gap 19: acc0, k-step 11
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[11]));
        DVQ_MF0(11);
        DVQ_RDF(14);
        if (!(ABL & 2)) { cv.template cvt<3>(c, t + 1, x); pk3 = cv.pk; }
        DVQ_SB();
        // gap 20: acc0, k-step 12
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[12]));
        DVQ_MF0(12);
        DVQ_RDF(15);
        if (!(ABL & 2)) { cv.template err<3>(pk3[0], pk3[1], x); DVQ_PIN2(cv.hh, cv.dsq); }
        DVQ_SB();
        // gap 21: acc0, k-step 13
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[13]));
        DVQ_MF0(13);
        DVQ_RDF(8);
        if (do_load) load_rows(c, t + 2, x);                            // the registers are free: rows of tile t+2
        if (!(ABL & 2)) { cv.hh = row16_sum(cv.hh); DVQ_PIN2(cv.hh, cv.dsq); }
        DVQ_SB();
        // gap 22: acc0, k-step 14
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[14]));
        DVQ_MF0(14);
        DVQ_RDF(9);
        if (!(ABL & 2)) { cv.dsq = row16_sum(cv.dsq); DVQ_PIN2(cv.hh, cv.dsq); }
        DVQ_SB();
        // gap 23: acc0, k-step 15
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[15]));
        DVQ_MF0(15);
        DVQ_RDF(10);
        if (!(ABL & 2)) cv.finish(c, t + 1);
        DVQ_SB();
        DVQ_STAMP(4);
        // gap 24: acc1, k-step 8
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[8]));
        DVQ_MF1(8);
        DVQ_RDF(11);
        m1 = INFINITY;
        m2 = INFINITY;
        DVQ_SB();
        // gap 25: acc1, k-step 9
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[9]));
        DVQ_MF1(9);
        DVQ_RDF(12);
        DVQ_SCORE(acc0, 0, 3)
        DVQ_SB();
        // gap 26: acc1, k-step 10
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[10]));
        DVQ_MF1(10);
        DVQ_RDF(13);
        DVQ_SCORE(acc0, 3, 6)
        DVQ_SB();
        // gap 27: acc1, k-step 11
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[11]));
        DVQ_MF1(11);
        DVQ_RDF(14);
        DVQ_SCORE(acc0, 6, 8)
        DVQ_SB();
        // gap 28: acc1, k-step 12
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[12]));
        DVQ_MF1(12);
        DVQ_RDF(15);
        DVQ_SCORE(acc0, 8, 10)
        DVQ_SB();
        // gap 29: acc1, k-step 13
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bf[13]));
        DVQ_MF1(13);
        DVQ_SCORE(acc0, 10, 12)
        DVQ_SB();
        // gap 30: acc1, k-step 14
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bf[14]));
        DVQ_MF1(14);
        DVQ_SCORE(acc0, 12, 14)
        DVQ_SB();
        // gap 31: acc1, k-step 15
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[15]));
        DVQ_MF1(15);
        if (!(ABL & 8)) { score<14, 16>(acc0, 0, m1, m2); write_slot<0>(c, t, m1, m2); }
        DVQ_SB();
        DVQ_STAMP(5);
    // >>> GENERATED
    };
    for (int t = 0; t < c.ntl; ++t) tile(t);
#undef DVQ_MF0
#undef DVQ_MF1
#undef DVQ_SB
#undef DVQ_PIN2
#undef DVQ_RDF
#undef DVQ_STAMP
#undef DVQ_SCORE
    wg_barrier();                                                       // every slot of the accumulator-0 lists is written
    // last scores, last merges
    {
        float m1 = INFINITY, m2 = INFINITY;
        score<0, 16>(acc1, 0, m1, m2);
        write_slot<1>(c, c.ntl - 1, m1, m2);
    }
    if (c.ntl >= 2) {
        mg.read(c, c.ntl - 2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mg.sl), "+v"(mg.rs));
        mg.min_a(); mg.min_b(c); mg.act(c, c.ntl - 2);
    }
    wg_barrier();
    mg.read(c, c.ntl - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mg.sl), "+v"(mg.rs));
    mg.min_a(); mg.min_b(c); mg.act(c, c.ntl - 1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();

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
            const bool act = pr != ~0u;                           // ~0: hole left by a reservation that did not fit
            const int rowslot = (int)(pr >> 16), k = (int)(pr & 0xffffu);
            float zz, dot;
            chain_pair_x4(z + grow_of(rowslot) * D, E + (long)k * D, q, act, zz, dot);
            if (act && q == 0) {
                const float tsum = zz + ee_g[k];
                atomicMin(&s_res[rowslot], order_key(tsum - 2.0f * dot, k));
            }
        }
    }
    const unsigned long long t2b = __builtin_amdgcn_s_memrealtime();
    // what is left (NaN/Inf, fp16 overflow, invalid codebook image, overflowing list): all K entries canonically,
    // the whole workgroup per row, one single-lane chain per entry (NT == K)
    for (int o = 0; o < n_slow; ++o) {
        const int rowslot = s_slow[o];
        float zz2, dot2;
        chain_pair(z + grow_of(rowslot) * D, E + (long)tid * D, zz2, dot2);
        const float tsum = zz2 + ee_g[tid];
        atomicMin(&s_res[rowslot], order_key(tsum - 2.0f * dot2, tid));
    }
    // rows that cost far more than a filtered row: the all-entries path and rows with two or more expanded slots (callers watch
    // this count to fall back to the exact kernel on ill-conditioned input, see dvq.h)
    if (slow_rows && tid == 0 && n_slow + (int)s_cnt[2] > 0) atomicAdd(slow_rows, (unsigned long long)(n_slow + (int)s_cnt[2]));
    __syncthreads();
    if (tid < c.ntl * TILE) {
        const long gr = grow_of(tid);
        if (gr < M) idx[gr] = (int64_t)(unsigned)(s_res[tid] & 0xffffffffull);
    }
    if (DBG && dbg) {
        if (tid == 0) {
            const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
            unsigned long long* o = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(dbg) + (size_t)blockIdx.x * DBG_WG_BYTES);
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3;
            o[4] = (unsigned long long)total; o[5] = (unsigned long long)n_slow;
            o[6] = t2b;
            o[7] = ((t0a - t0) << 32) | ((t0b - t0) & 0xffffffffull);   // prologue: rows of tile 0 arrived, tile 0 converted
        }
        if (tid < NWV * 32)
            reinterpret_cast<unsigned*>(reinterpret_cast<char*>(dbg) + (size_t)blockIdx.x * DBG_WG_BYTES + 64)[tid] =
                reinterpret_cast<const unsigned*>(lds + L_DBG)[tid];
    }
}

int device_cus() {
    static int cus[128] = {0};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 127;
    int v = __atomic_load_n(&cus[d], __ATOMIC_RELAXED);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        __atomic_store_n(&cus[d], v, __ATOMIC_RELAXED);
    }
    return v;
}

}  // namespace

// K = 512 is the kernels' shape; fewer entries (a multiple of 32) run the same kernels on a padded image (vq_pack_norm_kernel)
extern "C" int dvq_vq_fast_supported(int Kq, int Dq) { return Dq == D && Kq >= 32 && Kq <= K && Kq % 32 == 0; }

extern "C" size_t dvq_vq_pack_bytes(int Kq, int Dq) { return dvq_vq_fast_supported(Kq, Dq) ? PK_BYTES : 0; }

extern "C" int dvq_vq_pack(const float* E, int Kq, int Dq, void* packed, size_t packed_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_pack: the fast path supports K<=%d (a multiple of 32), D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(E && packed && dvq_aligned16(E) && dvq_aligned16(packed), "vq_pack: null/unaligned pointer");
    DVQ_REQUIRE(packed_bytes >= PK_BYTES, "vq_pack: buffer %zu < %zu bytes", packed_bytes, PK_BYTES);
    hipStream_t st = (hipStream_t)stream;
    char* pk = (char*)packed;
    DVQ_LAUNCH(vq_pack_norm_kernel, dim3(1), dim3(K), 0, st, E, Kq, (float*)(pk + PK_OFF_EE), (PackHeader*)pk);
    DVQ_CHECK_LAUNCH("vq_pack_norm");
    DVQ_LAUNCH(vq_pack_img_kernel, dim3((K * D + 255) / 256), dim3(256), 0, st, E, Kq, (const PackHeader*)pk,
                       (_Float16*)(pk + PK_OFF_IMG));
    DVQ_CHECK_LAUNCH("vq_pack_img");
    DVQ_LAUNCH(vq_pack_err_kernel, dim3(1), dim3(K), 0, st, E, Kq, (const _Float16*)(pk + PK_OFF_IMG), (PackHeader*)pk);
    DVQ_CHECK_LAUNCH("vq_pack_err");
    return DVQ_OK;
}

extern "C" size_t dvq_vq_fast_workspace_bytes(int64_t M, int Kq, int Dq) {
    (void)M;
    if (!dvq_vq_fast_supported(Kq, Dq)) return 256;
    return dvq_round_up((size_t)1024 * DBG_WG_BYTES, 256);        // per-workgroup phase stamps (DVQ_VQ_DBG only)
}

extern "C" int dvq_vq_argmin_fast(const float* z, const float* E, const void* packed, int64_t M, int Kq, int Dq,
                                  int64_t* idx, unsigned long long* slow_rows, void* workspace, size_t workspace_bytes,
                                  dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_argmin_fast: supports K<=%d (a multiple of 32), D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(M >= 0 && M < (1L << 31), "vq_argmin_fast: bad M");
    if (M == 0) return DVQ_OK;
    DVQ_REQUIRE(z && E && packed && idx && workspace, "vq_argmin_fast: null pointer");
    DVQ_REQUIRE(dvq_aligned16(z) && dvq_aligned16(E) && dvq_aligned16(packed) && dvq_aligned16(workspace),
                "vq_argmin_fast: pointers must be 16-byte aligned (z must be dense [M,256])");
    if (workspace_bytes < dvq_vq_fast_workspace_bytes(M, Kq, Dq)) {
        dvq_set_error("vq_argmin_fast: workspace %zu < %zu bytes", workspace_bytes, dvq_vq_fast_workspace_bytes(M, Kq, Dq));
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            hipError_t e = hipSuccess;
#ifdef DVQ_DIAG
            for (const void* fn : {(const void*)&vq_stream_kernel<0, false>, (const void*)&vq_stream_kernel<0, true>,
                               (const void*)&vq_stream_kernel<1, true>, (const void*)&vq_stream_kernel<2, true>,
                               (const void*)&vq_stream_kernel<4, true>, (const void*)&vq_stream_kernel<8, true>,
                               (const void*)&vq_stream_kernel<16, true>, (const void*)&vq_stream_kernel<32, true>,
                               (const void*)&vq_stream_kernel<15, true>, (const void*)&vq_stream_kernel<47, true>}) {
#else
            for (const void* fn : {(const void*)&vq_stream_kernel<0, false>}) {
#endif
                const hipError_t e1 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
                if (e1 != hipSuccess) e = e1;
            }
            return e;
        });
        if (e != hipSuccess) {
            dvq_set_error("vq_argmin_fast: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const char* pk = (const char*)packed;
    // default (round 3, after the library lost its compiler-formed packed-fp32 instructions): the sixteen-wave kernel of
    // vq_stream16.hip (34.1-35.0 us against 35.6-35.9 us for this file's eight-wave kernel, DVQ_VQ_KERNEL=8, in bench.py's
    // microbenchmark; 36.4 us both before); DVQ_VQ_KERNEL=32: vq_rows.hip (rows resident, codebook streamed: 42 us).  Same indices.
    int which = dvq_knobs().vq_kernel;
    if (Kq != K) which = 16;                                        // a padded image (fewer than 512 entries): the default kernel knows where the real entries end
    unsigned long long* dbg16 = nullptr;
#ifdef DVQ_DIAG
    if (getenv("DVQ_VQ_DBG") || getenv("DVQ_VQ_ABL")) which = 8;                 // stamps / ablations of the eight-wave kernel
    if (getenv("DVQ_VQ16_DBG") || getenv("DVQ_VQ16_ABL")) which = 16;
    if (getenv("DVQ_VQ16_DBG")) dbg16 = (unsigned long long*)workspace;          // phase stamps of the sixteen-wave kernel
    if (getenv("DVQ_VQP_DBG") || getenv("DVQ_VQP_ABL") || getenv("DVQ_VQP_VAR")) which = 17;
    if (getenv("DVQ_VQP_DBG")) dbg16 = (unsigned long long*)workspace;           // phase / wave stamps of vq_pipe.hip
#endif
    if (which == 17) {
        DVQ_PROF("vq_argmin_fast", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
        return dvq_launch_vq_pipe(z, E, packed, (long)M, idx, slow_rows, dbg16, st);
    }
    if (which == 16) {
        DVQ_PROF("vq_argmin_fast", 2.0 * M * Kq * D, (double)M * D * 4 + (double)Kq * D * 4 + (double)M * 8, st);
        return dvq_launch_vq_stream16(z, E, packed, (long)M, idx, slow_rows, dbg16, st);
    }
    if (which == 32) {
        DVQ_PROF("vq_argmin_fast", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
        return dvq_launch_vq_rows(z, E, packed, (long)M, idx, slow_rows, st);
    }
    const int cus = device_cus();
    const long tiles = (M + TILE - 1) / TILE;
    const long per_launch = (long)cus * MAX_TILES;                 // one workgroup per CU, <= MAX_TILES tiles each
#ifdef DVQ_DIAG
    const bool want_dbg = getenv("DVQ_VQ_DBG") != nullptr;       // phase stamps / timing-only ablations: diagnostics build only
    const char* abl_s = getenv("DVQ_VQ_ABL");
    const int abl = abl_s ? atoi(abl_s) : 0;
#endif
    DVQ_PROF("vq_argmin_fast", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
    for (long t0 = 0; t0 < tiles; t0 += per_launch) {
        const long nt = (tiles - t0 < per_launch) ? tiles - t0 : per_launch;
        const unsigned grid = (unsigned)(nt < cus ? nt : cus);
#ifdef DVQ_DIAG
        unsigned long long* dbgp = (want_dbg && t0 == 0) ? (unsigned long long*)workspace : nullptr;
#define DVQ_VQ_GO(A) DVQ_LAUNCH((vq_stream_kernel<A, true>), dim3(grid), dim3(NT), LDS_BYTES, st, z, E, (long)M, t0, nt, pk, idx, slow_rows, dbgp)
        switch (abl) {
            case 1: DVQ_VQ_GO(1); break;
            case 2: DVQ_VQ_GO(2); break;
            case 4: DVQ_VQ_GO(4); break;
            case 8: DVQ_VQ_GO(8); break;
            case 16: DVQ_VQ_GO(16); break;
            case 32: DVQ_VQ_GO(32); break;
            case 15: DVQ_VQ_GO(15); break;
            case 47: DVQ_VQ_GO(47); break;
            default:
                if (dbgp) DVQ_VQ_GO(0);
                else DVQ_LAUNCH((vq_stream_kernel<0, false>), dim3(grid), dim3(NT), LDS_BYTES, st, z, E, (long)M, t0, nt, pk, idx, slow_rows, dbgp);
                break;
        }
#undef DVQ_VQ_GO
#else
        DVQ_LAUNCH((vq_stream_kernel<0, false>), dim3(grid), dim3(NT), LDS_BYTES, st, z, E, (long)M, t0, nt, pk, idx, slow_rows,
                   (unsigned long long*)nullptr);
#endif
        DVQ_CHECK_LAUNCH("vq_stream");
    }
    return DVQ_OK;
}
