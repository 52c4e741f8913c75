// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256), fourth structure: ROWS stay, the CODEBOOK streams.
// Same result as vq_stream.hip / vq_stream16.hip / vq.hip / oracle/vq_canonical.c, bit for bit.  Reference:
// VectorQuantizer.forward(z, istrain=False), network/vqvae/quantizer.py:46-49.
//
// The three earlier structures keep the codebook in registers and pass every 32-row tile of z through the LDS to all waves of the
// CU: per tile a workgroup-wide barrier, a cross-wave merge of (min, second) slots and one dependency chain load -> convert ->
// LDS -> barrier -> MFMAs -> scores -> slots -> merge; all three take 36-40 us (vq_stream16.hip's header has the measurements).
// Here the roles are swapped:
//   * a wave owns 32 rows for its whole life: it loads them (coalesced), converts them to fp16 once (|h|^2, the measured rounding
//     error, eps_row as in vq_stream.hip), passes them through its own LDS slot to get the MFMA B-operand layout and keeps them
//     in 64 VGPRs;
//   * the codebook image (fp16 of -2 sE e_k in MFMA-fragment order, dvq_vq_pack) streams L2 -> LDS in 32-entry blocks of 16 KiB,
//     loaded once per workgroup (four waves, 1 KiB per wave and instruction) into a four-slot ring, one barrier per PAIR of
//     blocks (eight per workgroup instead of one per tile);
//   * D = block x rows: a lane holds ONE row's scores against 16 entries per block, so the running five smallest of a row are
//     lane-local (6 vector instructions per score: id into the low 8 mantissa bits, four v_med3, one v_min, placed under the
//     MFMAs of the next pair); no slots, no cross-wave merge;
//   * two workgroups per CU (128 rows each): the second one's rows stream from HBM while the first one multiplies.
// MEASURED (M = 65 536, tools/vq_kernel_ab.py): 42 us against 34-36 us for the other two -- selectable (DVQ_VQ_KERNEL=32), not the
// default.  Compile-time ablations (-DVQR_ABL, tools/vq_rows_abl.py): rows -> registers (phase 0) 15.9 us: the 64 MiB arrive at
// ~5.5 TB/s and NOTHING overlaps them, every wave of the chip waits for its rows at the same time; the stream 18.5 us, of which
// 8.5 us are MFMAs (the floor for two waves per SIMD) -- the rest is the LDS: every wave reads the whole 256 KiB image, 2 MiB per CU
// = 16 k cycles at 128 B per clock, as long as the matrix work; scoring 2.3 us; refine tail ~8 us.  With a top-three instead of
// a top-five per lane half 49 rows per call took the all-entries scan (+35 us).  What this structure would need: 64 rows per
// wave (each fragment read feeds two MFMAs: half the LDS traffic) AND a second batch of rows in flight under the first one's
// products -- both at once do not fit the register file at two waves per SIMD.
// After the stream the two lane halves of a row are merged; a unique score within eps_row of the minimum decides the row,
// otherwise the candidates (at most four per half; a half whose fifth score is in range may hide a sixth: all-entries scan)
// are evaluated in the canonical fp32 order, eight lanes per 256-step chain, exactly as in the other kernels.
#include "dvq_internal.h"
#include "vq_pack.h"
#include <vector>

#ifndef VQR_ABL
#define VQR_ABL 0          // timing experiments (results invalid unless 0): 1 no refine tail, 2 no scoring, 4 no MFMA, 8 no stream
#endif
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int K = VQ_K, D = VQ_D;
constexpr int WPB = 4, NT = 64 * WPB;               // four waves per workgroup, one per SIMD; two workgroups per CU
constexpr int ROWS_B = 32 * WPB;                    // 128 rows per workgroup
constexpr int NBLK = K / 32;                        // 16 codebook blocks of 32 entries
constexpr int Z16_ROW = 528;                        // padded fp16 row: 512 B + 16 B (conflict-free 16-byte reads down a column)
constexpr int SLOT = 32 * Z16_ROW;                  // 16 896 B: a wave's fp16 row tile, later a codebook block (16 384 B)
constexpr int PAIR_CAP = 512;

constexpr int L_RING = 0;                           // 4 slots
constexpr int L_EES = L_RING + 4 * SLOT;            // [K] f32: sE |e_k|^2 (accumulator start values)
constexpr int L_RS = L_EES + K * 4;                 // [128] {eps sE, bad}
constexpr int L_RES = L_RS + ROWS_B * 8;            // [128] u64 row results (ordered distance bits : entry)
constexpr int L_PAIR = L_RES + ROWS_B * 8;          // [PAIR_CAP] u32 (row << 16 | entry)
constexpr int L_SLOW = L_PAIR + PAIR_CAP * 4;       // [128] u16 rows for the all-entries path
constexpr int L_CNT = L_SLOW + ROWS_B * 2;          // [0] pairs, [1] slow rows
constexpr int LDS_BYTES = L_CNT + 64;
static_assert(2 * LDS_BYTES <= 160 * 1024 && L_EES % 16 == 0 && L_RS % 8 == 0 && L_RES % 8 == 0, "LDS layout");

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {           // all-reduce over the 16 lanes of a DPP row
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}
__device__ __forceinline__ float min_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -3.0e38f); }
template <int HI>
__device__ __forceinline__ float mix_diff(float hp, float x) {           // h - x with h one fp16 half of hp: exact (one v_fma_mix_f32)
    float d;
    if (HI) asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    else asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hp), "v"(x));
    return d;
}
__device__ __forceinline__ unsigned long long order_key(float d, int k) {
    const unsigned b = __float_as_uint(d);
    const unsigned u = (d != d) ? 0u : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
    return ((unsigned long long)u << 32) | (unsigned)k;
}
// entry of a packed score: id = block << 4 | accumulator register (v_mfma_f32_32x32x16 D layout), lane half h
__device__ __forceinline__ int entry_of(unsigned id, int h) { return 32 * (int)(id >> 4) + 8 * (int)((id >> 2) & 3) + 4 * h + (int)(id & 3); }

// Canonical chains threaded through 8 lanes (vq_stream16.hip): lane q holds floats [32q, 32q+32) of the z row and of the
// candidate's codebook row, the k-ordered fmaf chain runs as eight 32-step rounds.  Bit-identical to one 256-step chain.
__device__ __forceinline__ void chain_pair_x8(const float* __restrict__ zr, const float* __restrict__ er, int q, bool active,
                                              float& zz, float& dot) {
    f32x4 x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        x[u] = active ? *reinterpret_cast<const f32x4*>(zr + 32 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
        y[u] = active ? *reinterpret_cast<const f32x4*>(er + 32 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float a = 0.f, b = 0.f;
    const int lane = threadIdx.x & 63, base = lane & ~7;
#pragma unroll
    for (int round = 0; round < 8; ++round) {
        const float a_in = round ? __shfl(a, base + round - 1) : 0.f;
        const float b_in = round ? __shfl(b, base + round - 1) : 0.f;
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
    zz = __shfl(a, base + 7);
    dot = __shfl(b, base + 7);
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

// one 32-entry block: 16 fragment reads + 16 MFMAs into `acc` (start values sE |e_k|^2)
struct Pair { f32x16 a0, a1; };

// the five smallest of a lane's scores so far, ids in the low 8 mantissa bits (6 vector instructions per score).  Five, not
// three: a lane half that holds k scores within eps of the row minimum is only safe while k < the depth -- beyond it an entry may
// hide and the row needs the all-entries scan (~17 us for its workgroup).  Rows with >= 3 in one half: 7.5e-4 of N(0,1) rows
// (49 per 65 536-row call with a depth of three: +35 us); with >= 5: ~1e-7.
struct Top5 {
    float m1, m2, m3, m4, m5;
    __device__ __forceinline__ void put(float score, unsigned id) {
        const float p = __uint_as_float((__float_as_uint(score) & ~255u) | id);
        m5 = __builtin_amdgcn_fmed3f(m4, m5, p);
        m4 = __builtin_amdgcn_fmed3f(m3, m4, p);
        m3 = __builtin_amdgcn_fmed3f(m2, m3, p);
        m2 = __builtin_amdgcn_fmed3f(m1, m2, p);
        m1 = min_nc(m1, p);
    }
};

__global__ __launch_bounds__(NT, 2) void vq_rows_kernel(const float* __restrict__ z, const float* __restrict__ E, long M,
                                                        const char* __restrict__ packed, int64_t* __restrict__ idx,
                                                        unsigned long long* __restrict__ slow_rows, int delay_ticks, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    // Stagger: the second half of the grid (the second workgroup of every CU when the dispatcher fills the chip breadth first;
    // placement is for speed only) starts its row loads `delay_ticks` x 10 ns late.  Otherwise every wave of the chip waits for
    // its rows at the same time and multiplies at the same time: 16 us of HBM stream with idle matrix cores, then the matrix work
    // with an idle HBM.  Staggered, the first half's rows arrive at the full HBM rate, and its products and refine tail run
    // while the second half's rows stream in.
    const unsigned long long t_begin = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (delay_ticks > 0 && blockIdx.x >= (gridDim.x + 1) / 2) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)delay_ticks) __builtin_amdgcn_s_sleep(32);
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r_l = lane & 31, h_l = lane >> 5;
    const long row0 = (long)blockIdx.x * ROWS_B;                          // first row of the workgroup
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    __builtin_amdgcn_s_setreg((2 - 1) << 11 | 6 << 6 | 1, 3);            // MODE.FP_DENORM[3:2] = 3: fp16 subnormals kept
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    const float emax = hdr->emax, demax = hdr->demax;
    const bool e_valid = hdr->valid != 0;
    const float sEf = e_valid ? pow2f(hdr->sexp) : 1.0f;

    // ---- phase 0: the wave's 32 rows, HBM -> registers (all 32 loads in flight), fp16 image in its ring slot, eps per row
    char* tile = lds + L_RING + wave * SLOT;
    {
        f32x4 x[8][4];                                                    // group u = rows 4u + (lane >> 4); 16 lanes per row
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long gr = row0 + 32 * wave + 4 * u + (lane >> 4);
            if (gr >= M) gr = M - 1;                                      // rows behind the end repeat the last row (never written)
            const f32x4* p = reinterpret_cast<const f32x4*>(z + gr * D) + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) x[u][q] = __builtin_nontemporal_load(p + 16 * q);
        }
        if (tid < K / 2) {                                                // accumulator start values (two per thread)
            reinterpret_cast<float*>(lds + L_EES)[tid] = ee_g[tid] * sEf;
            reinterpret_cast<float*>(lds + L_EES)[tid + K / 2] = ee_g[tid + K / 2] * sEf;
        }
        if (tid < ROWS_B) reinterpret_cast<unsigned long long*>(lds + L_RES)[tid] = ~0ull;
        if (tid < 16) reinterpret_cast<unsigned*>(lds + L_CNT)[tid] = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int g = lane >> 4, i = lane & 15, row = 4 * u + g;
            char* za = tile + row * Z16_ROW + 8 * i;
            float hh = 0.f, dsq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x2 a, b, pk;
                a[0] = x[u][q][0]; a[1] = x[u][q][1]; b[0] = x[u][q][2]; b[1] = x[u][q][3];
                const f16x2 lo = __builtin_convertvector(a, f16x2), hi = __builtin_convertvector(b, f16x2);
                hh = __builtin_amdgcn_fdot2(lo, lo, hh, false);
                hh = __builtin_amdgcn_fdot2(hi, hi, hh, false);
                pk[0] = __builtin_bit_cast(float, lo);
                pk[1] = __builtin_bit_cast(float, hi);
                *reinterpret_cast<f32x2*>(za + 128 * q) = pk;
                const float d0 = mix_diff<0>(pk[0], x[u][q][0]), d1 = mix_diff<1>(pk[0], x[u][q][1]);
                const float d2 = mix_diff<0>(pk[1], x[u][q][2]), d3 = mix_diff<1>(pk[1], x[u][q][3]);
                dsq = fmaf(d0, d0, dsq); dsq = fmaf(d1, d1, dsq); dsq = fmaf(d2, d2, dsq); dsq = fmaf(d3, d3, dsq);
            }
            hh = row16_sum(hh);
            dsq = row16_sum(dsq);
            const float hn = __builtin_amdgcn_sqrtf(hh);
            const float dzn = __builtin_amdgcn_sqrtf(dsq) * 1.0001f;      // |z - h(z)|, measured
            const float zn = (hn + dzn) * 1.0001f;                        // |z| <= |h(z)| + |z - h(z)|
            const float uu = zn + emax;
            // vq_stream.hip's bound with 8 instead of 5 id bits in the score's mantissa: + (2^-15 - 2^-18) (|z| + Emax)^2
            const float eps = 4.004f * (dzn * emax + zn * demax + dzn * demax) + 1.30e-4f * uu * uu;
            const float epsS = eps * sEf;
            const bool bad = !e_valid || !(hh <= 3.0e38f) || !(dsq <= 3.0e38f) || !(epsS <= 3.0e38f);   // NaN/Inf, fp16 overflow
            if (i == 0) {
                f32x2 rs;
                rs[0] = epsS;
                rs[1] = __uint_as_float(bad ? 1u : 0u);
                *reinterpret_cast<f32x2*>(lds + L_RS + (32 * wave + row) * 8) = rs;
            }
        }
    }
    const unsigned long long t_rows = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // the rows as MFMA B operand: lane (row r_l, half h_l) holds k = 16 s + 8 h .. + 7 of its row for every k-step s
    f16x8 bz[16];
    {
        const char* zt = tile + r_l * Z16_ROW + 16 * h_l;
#pragma unroll
        for (int s = 0; s < 16; ++s) bz[s] = *reinterpret_cast<const f16x8*>(zt + 32 * s);
    }
    // ---- phase 1: the codebook streams through the ring; wave w moves fragments 4w .. 4w+3 of every block
    const char* img = packed + PK_OFF_IMG;
    u32x4 st[8];                                                          // the next pair of blocks in flight (2 x 4 KiB per wave)
    auto load_pair = [&](int p) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                st[4 * b + f] = *reinterpret_cast<const u32x4*>(img + ((size_t)(2 * p + b) * 16 + 4 * wave + f) * 1024 + 16 * lane);
    };
    auto store_pair = [&](int p) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                *reinterpret_cast<u32x4*>(lds + L_RING + ((2 * p + b) & 3) * SLOT + (4 * wave + f) * 1024 + 16 * lane) = st[4 * b + f];
    };
    load_pair(0);
    __syncthreads();                                                      // every wave has its rows in registers: the ring is free; EES filled
    store_pair(0);
    load_pair(1);
    Top5 top = {INFINITY, INFINITY, INFINITY, INFINITY, INFINITY};
    auto init_acc = [&](int blk, f32x16& acc) {
        const f32x4* ci = reinterpret_cast<const f32x4*>(lds + L_EES + (32 * blk + 4 * h_l) * 4);    // registers 4q..4q+3 <-> entries 8q + 4h ..
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = ci[2 * q];
            acc[4 * q] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
        }
    };
    auto multiply = [&](int p, Pair& acc) {                                // the two blocks of pair p, interleaved (independent chains)
        const char* s0 = lds + L_RING + ((2 * p) & 3) * SLOT + 16 * lane;
        const char* s1 = lds + L_RING + ((2 * p + 1) & 3) * SLOT + 16 * lane;
        init_acc(2 * p, acc.a0);
        init_acc(2 * p + 1, acc.a1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f16x8 f0 = *reinterpret_cast<const f16x8*>(s0 + 1024 * s);
            const f16x8 f1 = *reinterpret_cast<const f16x8*>(s1 + 1024 * s);
            if (!(VQR_ABL & 4)) {
                acc.a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f0, bz[s], acc.a0, 0, 0, 0);
                acc.a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1, bz[s], acc.a1, 0, 0, 0);
            } else asm volatile("" :: "v"(f0), "v"(f1));
        }
    };
    auto score = [&](int p, const Pair& acc) {
        if (VQR_ABL & 2) { top.put(acc.a0[0] + acc.a1[3], 1u); return; }
#pragma unroll
        for (int e = 0; e < 16; ++e) top.put(acc.a0[e], (unsigned)(((2 * p) << 4) | e));
#pragma unroll
        for (int e = 0; e < 16; ++e) top.put(acc.a1[e], (unsigned)(((2 * p + 1) << 4) | e));
    };
    // 1 MFMA : 6 vector instructions: the previous pair's 32 scores per lane under this pair's 32 MFMAs
#define VQR_INTERLEAVE()                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 32; ++i_) {                   \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                  \
        __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);                  \
    }
    Pair pa, pb;
    __syncthreads();                                                      // pair 0 is in the ring
#pragma unroll 1
    for (int p = 0; p < ((VQR_ABL & 8) ? 0 : NBLK / 2); p += 2) {
        // even pair -> pa (scores of the odd pair before it), odd pair -> pb
        store_pair(p + 1);                                                // (slots of pair p - 1: everybody left them at the last barrier)
        if (p + 2 < NBLK / 2) load_pair(p + 2);
        multiply(p, pa);
        if (p > 0) { score(p - 1, pb); VQR_INTERLEAVE(); }
        __syncthreads();
        if (p + 2 < NBLK / 2) store_pair(p + 2);
        if (p + 3 < NBLK / 2) load_pair(p + 3);
        multiply(p + 1, pb);
        score(p, pa);
        VQR_INTERLEAVE();
        __syncthreads();
    }
    score(NBLK / 2 - 1, pb);
#undef VQR_INTERLEAVE

    const unsigned long long t_stream = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // ---- phase 2: the two lane halves of a row -> decision or candidates (lanes 0..31, one row each)
    {
        auto upper = [](float v) { return __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false)[1]); };
        const float o1 = upper(top.m1), o2 = upper(top.m2), o3 = upper(top.m3), o4 = upper(top.m4), o5 = upper(top.m5);
        const int row = 32 * wave + r_l;
        const long gr = row0 + row;
        if (h_l == 0 && gr < M) {
            const f32x2 rs = *reinterpret_cast<const f32x2*>(lds + L_RS + row * 8);
            const bool bad = __float_as_uint(rs[1]) != 0u;
            const float thr = min_nc(top.m1, o1) + rs[0];
            const float v[8] = {top.m1, top.m2, top.m3, top.m4, o1, o2, o3, o4};
            int n = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) n += v[i] <= thr ? 1 : 0;
            const bool hidden = top.m5 <= thr || o5 <= thr;               // five of a half in range: a sixth may hide behind them
            unsigned* cnt = reinterpret_cast<unsigned*>(lds + L_CNT);
            if (bad || hidden || n == 0) {
                const unsigned sp = atomicAdd(cnt + 1, 1u);
                reinterpret_cast<uint16_t*>(lds + L_SLOW)[sp] = (uint16_t)row;
            } else if (n == 1) {
                const bool first = top.m1 <= thr;
                idx[gr] = (int64_t)entry_of(__float_as_uint(first ? top.m1 : o1) & 255u, first ? 0 : 1);   // decided: its result slot stays ~0
            } else {
                const unsigned pos = atomicAdd(cnt, (unsigned)n);
                if (pos + (unsigned)n <= (unsigned)PAIR_CAP) {
                    unsigned o = pos;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (v[i] <= thr) reinterpret_cast<unsigned*>(lds + L_PAIR)[o++] = ((unsigned)row << 16) | (unsigned)entry_of(__float_as_uint(v[i]) & 255u, i / 4);
                } else {
                    const unsigned sp = atomicAdd(cnt + 1, 1u);
                    reinterpret_cast<uint16_t*>(lds + L_SLOW)[sp] = (uint16_t)row;
                }
            }
        }
    }
    __syncthreads();
    // ---- refine: canonical distances of the listed pairs, eight lanes per chain; key minimum per row
    unsigned long long* s_res = reinterpret_cast<unsigned long long*>(lds + L_RES);
    const unsigned* s_pair = reinterpret_cast<const unsigned*>(lds + L_PAIR);
    const uint16_t* s_slow = reinterpret_cast<const uint16_t*>(lds + L_SLOW);
    const unsigned* s_cnt = reinterpret_cast<const unsigned*>(lds + L_CNT);
    const int total = (int)min(s_cnt[0], (unsigned)PAIR_CAP);
    const int n_slow = (int)s_cnt[1];
    for (int s0 = 0; s0 < ((VQR_ABL & 1) ? 0 : total); s0 += NT / 8) {
        if (s0 + wave * 8 < total) {                                      // wave-uniform: this wave has at least one pair
            const int slot = s0 + (tid >> 3), q = tid & 7;
            const unsigned pr = slot < total ? s_pair[slot] : ~0u;
            const bool act = pr != ~0u;
            const int row = act ? (int)(pr >> 16) : 0, k = act ? (int)(pr & 0xffffu) : 0;
            float zz, dot;
            chain_pair_x8(z + (row0 + row) * D, E + (long)k * D, q, act, zz, dot);
            if (act && q == 0) {
                const float tsum = zz + ee_g[k];
                atomicMin(&s_res[row], order_key(tsum - 2.0f * dot, k));
            }
        }
    }
    // what is left (NaN/Inf, fp16 overflow, invalid codebook image, a possible fourth candidate, a full list): all K entries
    for (int o = 0; o < ((VQR_ABL & 1) ? 0 : n_slow); ++o) {
        const int row = s_slow[o];
#pragma unroll 1
        for (int k = tid; k < K; k += NT) {
            float zz2, dot2;
            chain_pair(z + (row0 + row) * D, E + (long)k * D, zz2, dot2);
            const float tsum = zz2 + ee_g[k];
            atomicMin(&s_res[row], order_key(tsum - 2.0f * dot2, k));
        }
    }
    if (slow_rows && tid == 0 && n_slow > 0) atomicAdd(slow_rows, (unsigned long long)n_slow);
    __syncthreads();
    if (dbg && tid == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = dbg + (size_t)blockIdx.x * 8;
        o[0] = t_begin; o[1] = t_rows; o[2] = t_stream; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = ((unsigned long long)xcc << 32) | hw;
    }
    if (tid < ROWS_B) {
        const long gr = row0 + tid;
        const unsigned long long res = s_res[tid];
        if (gr < M && res != ~0ull) idx[gr] = (int64_t)(unsigned)(res & 0xffffffffull);      // rows the refine or the all-entries scan decided
    }
}

}  // namespace

// called by dvq_vq_argmin_fast (vq_stream.hip) after argument validation
int dvq_launch_vq_rows(const float* z, const float* E, const void* packed, long M, int64_t* idx, unsigned long long* slow_rows,
                       hipStream_t st) {
    // DVQ_VQ_ROWS_DBG=1, diagnostics builds (-DDVQ_DIAG, tools/diag/) only: per-workgroup phase stamps, printed after the launch
    // (synchronises; one buffer, first device only).  The product library compiles none of it: dbg_on is a constant false there.
#ifdef DVQ_DIAG
    static unsigned long long* dbg_buf = nullptr;
    static const bool dbg_on = getenv("DVQ_VQ_ROWS_DBG") != nullptr;
    if (dbg_on && !dbg_buf && hipMalloc(&dbg_buf, 8 * 8 * 4096) != hipSuccess) {
        dvq_set_error("vq_rows: debug buffer allocation failed");
        return DVQ_ELAUNCH;
    }
#else
    constexpr unsigned long long* dbg_buf = nullptr;
    constexpr bool dbg_on = false;
#endif
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            return hipFuncSetAttribute((const void*)&vq_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        });
        if (e != hipSuccess) {
            dvq_set_error("vq_argmin_fast: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const long blocks = (M + ROWS_B - 1) / ROWS_B;
    const long per_launch = 1L << 20;
    for (long b0 = 0; b0 < blocks; b0 += per_launch) {
        const long nb = blocks - b0 < per_launch ? blocks - b0 : per_launch;
        const long r0 = b0 * ROWS_B;
        DVQ_LAUNCH(vq_rows_kernel, dim3((unsigned)nb), dim3(NT), LDS_BYTES, st, z + r0 * D, E, M - r0, (const char*)packed, idx + r0, slow_rows,
                   nb >= 256 ? dvq_knobs().vq_rows_delay : 0, (dbg_on && nb <= 4096) ? dbg_buf : nullptr);
        DVQ_CHECK_LAUNCH("vq_rows");
        if (dbg_on && nb <= 4096) {
            static int printed = 0;
            std::vector<unsigned long long> h((size_t)nb * 8);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost);
            if (printed++ == 12) {
                unsigned long long t0 = ~0ull;
                for (long b = 0; b < nb; ++b) t0 = h[8 * b] < t0 ? h[8 * b] : t0;
                double acc[2][4] = {{0}};
                for (long b = 0; b < nb; ++b) { const int g = b >= (nb + 1) / 2; for (int q = 0; q < 4; ++q) acc[g][q] += (double)(h[8 * b + q] - t0) * 0.01 / ((nb + 1) / 2); }
                for (int g = 0; g < 2; ++g) fprintf(stderr, "[vq_rows dbg] %s half: start %.2f us, rows in registers %.2f, stream done %.2f, end %.2f\n", g ? "second" : "first", acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
                // do blocks b and b + nb/2 share a CU?
                long same = 0;
                for (long b = 0; b < nb / 2; ++b) same += (h[8 * b + 4] & 0xffffffff0000ff00ull) == (h[8 * (b + nb / 2) + 4] & 0xffffffff0000ff00ull);   // xcc, se/sh/cu bits
                fprintf(stderr, "[vq_rows dbg] blocks b and b + %ld on the same (XCC, SE, CU): %ld of %ld; hw_id[0]=%llx hw_id[%ld]=%llx\n", nb / 2, same, nb / 2, h[4], nb / 2, h[8 * (nb / 2) + 4]);
            }
        }
    }
    return DVQ_OK;
}
