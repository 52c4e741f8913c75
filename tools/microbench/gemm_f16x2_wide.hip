// Microbench (GPU box): the three-product fp16 split GEMM ("f16x2") on the gated PixelCNN shapes, both MFMA shapes
// (v_mfma_f32_32x32x16_f16 and v_mfma_f32_16x16x32_f16) at the same 64 x 64 output tile per wave, 128 x 256 workgroup tile,
// eight waves, two LDS stages -- the structure of gemm_bf16x3_wide_kernel with half the matrix work.
//   a = a1 + a2,  a1 = fp16(a), a2' = fp16((a - a1) * 2^11);  w scaled per output row by 2^t so that max|w| lands in [2^14, 2^15)
//   hi += a1 w1 ;  lo += a1 w2' + a2' w1 ;  out = (hi + lo * 2^-11) * 2^-t (+ bias)
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/microbench/gemm_f16x2_wide.hip -o gpurun_out/gemm_f16x2_wide -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <type_traits>
#include "../../include/dvq.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct P {
    const float* A; long lda;
    const uint16_t* Wp; long wp_plane; long ldw;
    int K, N; long M;
    const float* cs; const float* bias; float* out; long ldo;
    unsigned long long* dbg;     // ABL 9: per-wave phase stamps (8 sums per wave)
    const uint16_t* Ap;          // ABL 13: the activations as fp16 plane pairs [M][2][K] (a row's pitch = the fp32 row's), both operands by LDS-DMA
};

constexpr int A_PL = 128 * 64, W_PL = 256 * 64, STAGE = 2 * A_PL + 2 * W_PL;   // 49 152 B
constexpr int BK = 32;

template <int SHAPE>
__device__ __forceinline__ int swz(int row) {
    const int g = (row >> 2) & 3;
    return SHAPE == 16 ? ((4 - g) & 3) : g;
}

__device__ __forceinline__ void split2(const f32x4& lo, const f32x4& hi, h8& p1, h8& p2) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float a0 = j < 4 ? lo[j] : hi[j - 4], a1 = j < 4 ? lo[j + 1] : hi[j - 3];
        const h2 h = __builtin_convertvector((float __attribute__((ext_vector_type(2)))){a0, a1}, h2);      // v_cvt_pk_f16_f32 (RNE)
        const float r0 = __builtin_fmaf((float)h[0], -1.0f, a0), r1 = __builtin_fmaf((float)h[1], -1.0f, a1);   // v_fma_mix_f32: exact
        const h2 l = __builtin_convertvector((float __attribute__((ext_vector_type(2)))){r0 * 2048.0f, r1 * 2048.0f}, h2);
        p1[j] = h[0]; p1[j + 1] = h[1];
        p2[j] = l[0]; p2[j + 1] = l[1];
    }
}

// ABL 13: what a producing epilogue would store -- row m: K first pieces, then K second pieces (same bytes as the fp32 row)
__global__ void presplit_rows(const float* __restrict__ A, uint16_t* __restrict__ Ap, long M, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;            // one thread per eight values
    const long per_row = K / 8, m = i / per_row;
    if (m >= M) return;
    const int k = (int)(i % per_row) * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(A + m * K + k), hi = *reinterpret_cast<const f32x4*>(A + m * K + k + 4);
    h8 p1, p2;
    split2(lo, hi, p1, p2);
    *reinterpret_cast<h8*>(Ap + m * 2 * K + k) = p1;
    *reinterpret_cast<h8*>(Ap + m * 2 * K + K + k) = p2;
}

__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int SHAPE, bool DEPHASE>
__global__ __launch_bounds__(512, 1) void gemm_f16x2_wide(const P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int tiles_n = p.N / 256;
    const long tiles_m = (p.M + 127) / 128;
    const long b = blockIdx.x;
    const long j = b >> 3;
    const long mt = (j / tiles_n) * 8 + (b & 7);
    const int nt = (int)(j % tiles_n);
    if (mt >= tiles_m) return;
    const long m0 = mt * 128;
    const int n0 = nt * 256;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // staging: activations through registers (row tid >> 2, eight k per thread), weights by LDS-DMA (four 1 KiB pieces per wave)
    const float* a_ptr;
    {
        long m = m0 + (tid >> 2);
        if (m >= p.M) m = p.M - 1;
        a_ptr = p.A + m * p.lda + 8 * (tid & 3);
    }
    const int a_dst = (tid >> 2) * 64 + 16 * ((tid & 3) ^ swz<SHAPE>(tid >> 2));
    const uint16_t* w_ptr[4];
    int w_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = wave * 4 + i;
        const int pl = id >> 4, rb = id & 15;
        const int row = rb * 16 + (lane >> 2);
        w_ptr[i] = p.Wp + pl * p.wp_plane + (long)(n0 + row) * p.ldw + 8 * ((lane & 3) ^ swz<SHAPE>(row));
        w_dst[i] = 2 * A_PL + pl * W_PL + rb * 1024;
    }
    int k_left = p.K;
    f32x4 alo, ahi;
    auto load_a = [&]() {
        alo = *reinterpret_cast<const f32x4*>(a_ptr);
        ahi = *reinterpret_cast<const f32x4*>(a_ptr + 4);
    };
    auto store_a = [&](char* stage) {
        h8 p1, p2;
        split2(alo, ahi, p1, p2);
        *reinterpret_cast<h8*>(stage + a_dst) = p1;
        *reinterpret_cast<h8*>(stage + A_PL + a_dst) = p2;
    };
    auto issue_w = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                             (__attribute__((address_space(3))) void*)(stage + w_dst[i]), 16, 0, 0);
    };
    auto advance = [&]() {
        a_ptr += BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) w_ptr[i] += BK;
        k_left -= BK;
    };

    load_a();
    issue_w(smem);
    advance();
    store_a(smem);
    bool more = k_left > 0;
    if (more) load_a();

    constexpr int NB = SHAPE == 16 ? 4 : 2;                    // blocks per 64 rows / columns
    constexpr int NR = SHAPE == 16 ? 4 : 16;                   // accumulator registers per block
    typedef float accv __attribute__((ext_vector_type(NR)));
    accv hi[NB][NB], lo[NB][NB];                               // [jn][i]
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int c = 0; c < NB; ++c)
#pragma unroll
            for (int e = 0; e < NR; ++e) { hi[a][c][e] = 0.f; lo[a][c][e] = 0.f; }

    int stage = 0;
    while (true) {
        dma_barrier();
        const char* st = smem + stage * STAGE;
        char* nx = smem + (stage ^ 1) * STAGE;
        auto feed = [&]() {
            store_a(nx);
            issue_w(nx);
            advance();
            if (k_left > 0) load_a();
        };
        if (more && (wave < 4 || !DEPHASE)) feed();
        if constexpr (SHAPE == 16) {
            const int lr = lane & 15, lc = lane >> 4;
            const int rd = lr * 64 + 16 * (lc ^ swz<16>(lr));
            h8 af[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    af[i][pl] = *reinterpret_cast<const h8*>(st + pl * A_PL + (wm * 64 + i * 16) * 64 + rd);
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                if (DEPHASE && jn == 2 && more && wave >= 4) feed();
                const h8 w1 = *reinterpret_cast<const h8*>(st + 2 * A_PL + (wn * 64 + jn * 16) * 64 + rd);
                const h8 w2 = *reinterpret_cast<const h8*>(st + 2 * A_PL + W_PL + (wn * 64 + jn * 16) * 64 + rd);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, af[i][0], hi[jn][i], 0, 0, 0);
                    lo[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, af[i][0], lo[jn][i], 0, 0, 0);
                    lo[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, af[i][1], lo[jn][i], 0, 0, 0);
                }
            }
        } else {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (DEPHASE && ks == 1 && more && wave >= 4) feed();
                h8 w[2][2], a[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        const int wrow = wn * 64 + q * 32 + r, arow = wm * 64 + q * 32 + r;
                        w[q][pl] = *reinterpret_cast<const h8*>(st + 2 * A_PL + pl * W_PL + wrow * 64 + 16 * ((2 * ks + h) ^ swz<32>(wrow)));
                        a[q][pl] = *reinterpret_cast<const h8*>(st + pl * A_PL + arow * 64 + 16 * ((2 * ks + h) ^ swz<32>(arow)));
                    }
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        hi[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[jn][0], a[i][0], hi[jn][i], 0, 0, 0);
                        lo[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[jn][1], a[i][0], lo[jn][i], 0, 0, 0);
                        lo[jn][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[jn][0], a[i][1], lo[jn][i], 0, 0, 0);
                    }
            }
        }
        if (!more) break;
        more = k_left > 0;
        stage ^= 1;
    }
    // epilogue: lanes <-> rows m, registers <-> four consecutive columns n
    if constexpr (SHAPE == 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + wm * 64 + i * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                const int n = n0 + wn * 64 + jn * 16 + 4 * (lane >> 4);
                f32x4 v = hi[jn][i] + lo[jn][i] * (1.0f / 2048.0f);
                v *= *reinterpret_cast<const f32x4*>(p.cs + n);
                v += *reinterpret_cast<const f32x4*>(p.bias + n);
                *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
            }
        }
    } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long m = m0 + wm * 64 + i * 32 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * 64 + jn * 32 + 8 * g + 4 * h;
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = hi[jn][i][4 * g + q] + lo[jn][i][4 * g + q] * (1.0f / 2048.0f);
                    v *= *reinterpret_cast<const f32x4*>(p.cs + n);
                    v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Ping-pong variant: the two waves of a SIMD (w and w + 4: rows 0..63 and 64..127 of the tile) alternate between a LOAD phase
// (fragment reads of tile t, conversion + LDS write of its share of tile t + 2, LDS-DMA of tile t + 2, global loads of tile t + 3)
// and a COMPUTE phase (48 MFMAs on the fragments in registers), one workgroup barrier per phase, waves 4..7 half a period behind.
// Three LDS stages: what is issued in a load phase has two phases to land.
constexpr int NSTAGE_PP = 3;
template <int PRIO, int ABL>
__global__ __launch_bounds__(512, 1) void gemm_f16x2_pp(const P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int tiles_n = p.N / 256;
    const long tiles_m = (p.M + 127) / 128;
    const long b = blockIdx.x;
    const long j = b >> 3;
    const long mt = (j / tiles_n) * 8 + (b & 7);
    const int nt = (int)(j % tiles_n);
    if (mt >= tiles_m) return;
    const long m0 = mt * 128;
    const int n0 = nt * 256;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const float* a_ptr;
    {
        long m = m0 + (tid >> 2);
        if (m >= p.M) m = p.M - 1;
        a_ptr = p.A + m * p.lda + 8 * (tid & 3);
    }
    const int a_dst = (tid >> 2) * 64 + 16 * ((tid & 3) ^ swz<16>(tid >> 2));
    const uint16_t* w_ptr[4];
    int w_dst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = wave * 4 + i;
        const int pl = id >> 4, rb = id & 15;
        const int row = rb * 16 + (lane >> 2);
        w_ptr[i] = p.Wp + pl * p.wp_plane + (long)(n0 + row) * p.ldw + 8 * ((lane & 3) ^ swz<16>(row));
        w_dst[i] = 2 * A_PL + pl * W_PL + rb * 1024;
    }
    const int T = p.K / BK;
    const uint16_t* ap_ptr[2];                              // ABL 13: two 1 KiB pieces of the activation planes per wave
    int ap_dst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = wave * 2 + i;
        const int pl = id >> 3, rb = id & 7;
        const int row = rb * 16 + (lane >> 2);
        long m = m0 + row;
        if (m >= p.M) m = p.M - 1;
        ap_ptr[i] = p.Ap + m * 2 * p.K + pl * p.K + 8 * ((lane & 3) ^ swz<16>(row));
        ap_dst[i] = pl * A_PL + rb * 1024;
    }
    auto issue_ap = [&](char* stage, int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ap_ptr[i] + (long)t * BK),
                                             (__attribute__((address_space(3))) void*)(stage + ap_dst[i]), 16, 0, 0);
    };
    f32x4 alo, ahi, blo, bhi;                               // ABL 4: a second register set, loads two periods ahead
    auto load_a = [&](int t) {
        alo = *reinterpret_cast<const f32x4*>(a_ptr + (long)t * BK);
        ahi = *reinterpret_cast<const f32x4*>(a_ptr + (long)t * BK + 4);
    };
    // ABL 10 / 11 / 12 are TIMING experiments only: with the DMA wait at the end of the compute phase, waves 0-3 may read a tile whose
    // DMA pieces waves 4-7 (half a period behind) have not waited for yet -- results can be wrong; not carried into the library.
    auto load_a_asm = [&](int t) {                          // ABL 10: loads the compiler does not track (waits are counted by hand)
        const float* q = a_ptr + (long)t * BK;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(alo), "=&v"(ahi) : "v"(q) : "memory");
    };
    auto load_b_asm = [&](int t) {
        const float* q = a_ptr + (long)t * BK;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(blo), "=&v"(bhi) : "v"(q) : "memory");
    };
    auto load_b = [&](int t) {
        blo = *reinterpret_cast<const f32x4*>(a_ptr + (long)t * BK);
        bhi = *reinterpret_cast<const f32x4*>(a_ptr + (long)t * BK + 4);
    };
    auto store_a = [&](char* stage) {
        h8 p1, p2;
        split2(alo, ahi, p1, p2);
        *reinterpret_cast<h8*>(stage + a_dst) = p1;
        *reinterpret_cast<h8*>(stage + A_PL + a_dst) = p2;
    };
    auto store_b = [&](char* stage) {
        h8 p1, p2;
        split2(blo, bhi, p1, p2);
        *reinterpret_cast<h8*>(stage + a_dst) = p1;
        *reinterpret_cast<h8*>(stage + A_PL + a_dst) = p2;
    };
    auto issue_w = [&](char* stage, int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[i] + (long)t * BK),
                                             (__attribute__((address_space(3))) void*)(stage + w_dst[i]), 16, 0, 0);
    };
    // ABL 3: weight planes through registers (global_load one period ahead, ds_write_b128 in the next load phase)
    uint4 wreg[4];
    auto load_w = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[i] = *reinterpret_cast<const uint4*>(w_ptr[i] + (long)t * BK);
    };
    auto store_w = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(stage + w_dst[i] + 16 * lane) = wreg[i];
    };
    // prologue: tiles 0 and 1 staged, tile 2's activations in registers
    if (ABL == 13) {
        issue_ap(smem, 0); issue_w(smem, 0);
        if (T > 1) { issue_ap(smem + STAGE, 1); issue_w(smem + STAGE, 1); }
    } else {
    load_a(0);
    issue_w(smem, 0);
    store_a(smem);
    if (T > 1) { load_a(1); issue_w(smem + STAGE, 1); store_a(smem + STAGE); }
    }
    if (ABL == 13) {} else
    if (ABL == 10 || ABL == 11 || ABL == 12) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (T > 2) load_a_asm(2);
        if (ABL != 10 && T > 3) load_b_asm(3);
    } else if (T > 2) load_a(2);
    if (ABL == 4 && T > 3) load_b(3);
    if (ABL == 3 && T > 2) load_w(2);
    if (ABL == 10 || ABL == 11 || ABL == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 hi[4][4], lo[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) { hi[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int lr = lane & 15, lc = lane >> 4;
    const int rd = lr * 64 + 16 * (lc ^ swz<16>(lr));
    unsigned long long clk0 = 0, rt0 = 0;
    if (ABL == 9) { clk0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    if (PRIO == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half, no per-phase flips
    if (wave >= 4) __builtin_amdgcn_s_barrier();           // the second half runs half a period behind
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto phase = [&](int t, auto setc) {
        constexpr int SET = decltype(setc)::value;
        unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0;
        if (ABL == 9) t0 = __builtin_amdgcn_s_memtime();
        // ---- load phase
        const char* st = smem + (t % NSTAGE_PP) * STAGE;
        h8 af[4][2], wf[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                af[i][pl] = *reinterpret_cast<const h8*>(st + pl * A_PL + (wm * 64 + i * 16) * 64 + rd);
                wf[i][pl] = *reinterpret_cast<const h8*>(st + 2 * A_PL + pl * W_PL + (wn * 64 + i * 16) * 64 + rd);
            }
        if (ABL == 9) { __builtin_amdgcn_sched_barrier(0); t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        if (t + 2 < T) {
            char* nx = smem + ((t + 2) % NSTAGE_PP) * STAGE;
            if (ABL == 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (ABL == 13) {                                // both operands by DMA: what this wave issued a period ago (tile t + 1) has landed
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                issue_ap(nx, t + 2);
                issue_w(nx, t + 2);
            } else
            if (ABL == 12) {                                // as 11 with the activation loads issued BEFORE the DMA pieces: they are older than
                                                            // DMA(t) .. forced at the end of compute phase t - 1: no wait here except at the start
                if (t < 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi) :: "memory");
                else asm volatile("" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi) :: "memory");
                if (SET) store_b(nx); else store_a(nx);
                if (t + 4 < T) { if (SET) load_b_asm(t + 4); else load_a_asm(t + 4); }
                issue_w(nx, t + 2);
            } else
            if (ABL == 11) {
                // Issue order per load phase: four DMA pieces (tile t + 2), then two activation loads (tile t + 4, into the register
                // set converted just above).  vmcnt counts in order: the activations of tile t + 2 -- issued two periods ago, behind
                // the DMA of tile t and ahead of everything younger -- are ready when all but the 6 + 2... youngest have landed.
                if (t < 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi) :: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" : "+v"(alo), "+v"(ahi), "+v"(blo), "+v"(bhi) :: "memory");
                if (SET) store_b(nx); else store_a(nx);
                issue_w(nx, t + 2);
                if (t + 4 < T) { if (SET) load_b_asm(t + 4); else load_a_asm(t + 4); }
            } else
            if (ABL == 10) {
                // activations of tile t + 2 (two loads, issued before the four DMA pieces of the previous load phase)
                if (t == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(alo), "+v"(ahi) :: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" : "+v"(alo), "+v"(ahi) :: "memory");
                store_a(nx);
                if (t + 3 < T) load_a_asm(t + 3);
                issue_w(nx, t + 2);
            } else
            if (ABL == 4) { if (SET) store_b(nx); else store_a(nx); }
            else if (ABL == 7) asm volatile("" :: "v"(alo), "v"(ahi));                  // loads kept, no split / LDS write
            else if (ABL != 2 && ABL != 5 && ABL != 13) store_a(nx);     // tile t + 2, loaded a period ago
            if (ABL == 9) { __builtin_amdgcn_sched_barrier(0); t3 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (ABL == 3) store_w(nx);
            else if (ABL != 1 && ABL != 10 && ABL != 11 && ABL != 12 && ABL != 13) issue_w(nx, t + 2);
            if (ABL == 4) { if (t + 4 < T) { if (SET) load_b(t + 4); else load_a(t + 4); } }
            else if (ABL == 5) { if (t + 3 < T) { if (SET) load_a(t + 3); else load_b(t + 3); } }      // the set that is NOT converted in this period
            else if (ABL == 10 || ABL == 11 || ABL == 12 || ABL == 13) {}
            else if (t + 3 < T) { if (ABL != 2 && ABL != 8) load_a(t + 3); if (ABL == 3) load_w(t + 3); }   // 8: split + write of stale registers, no loads
        }
        if (ABL == 9) { __builtin_amdgcn_sched_barrier(0); t4 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        if (ABL == 13 && t + 2 >= T) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers before the phase ends
        if (ABL == 9) { t5 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_barrier();
        if (ABL == 9) { t6 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        // ---- compute phase
        if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (ABL == 5) {                                     // conversion of tile t + 2 (loaded a period ago) in the MFMA shadow; past the
                                                            // last tile it rewrites a stage nobody reads (no branch: one scheduling region)
            char* nx = smem + ((t + 2) % NSTAGE_PP) * STAGE;
            if (SET) store_b(nx); else store_a(nx);
        }
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int i = 0; i < 4; ++i) hi[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jn][0], af[i][0], hi[jn][i], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int i = 0; i < 4; ++i) lo[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jn][1], af[i][0], lo[jn][i], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < 4; ++jn)
#pragma unroll
            for (int i = 0; i < 4; ++i) lo[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jn][0], af[i][1], lo[jn][i], 0, 0, 0);
        if (ABL == 5) {
#pragma unroll
            for (int q = 0; q < 20; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // two vector instructions of the split
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);         // the two LDS writes
            __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (ABL == 12) {                                    // DMA(t + 1) landed (and A(t + 3) before it): younger = A(t + 4), DMA(t + 2)
            if (t + 4 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (ABL == 11) {                                    // DMA(t + 1) landed: younger = A(t + 3), DMA(t + 2), A(t + 4)
            if (t + 4 < T) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (ABL == 10) {                                    // the DMA pieces of tile t + 1 (issued a period and a half ago) have landed
            if (t + 3 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (ABL == 9) { __builtin_amdgcn_sched_barrier(0); t7 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_barrier();
        if (ABL == 9 && t + 3 < T && t >= 2) {
            const unsigned long long t8 = __builtin_amdgcn_s_memtime();
            acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += t4 - t3; acc_t[4] += t5 - t4; acc_t[5] += t6 - t5;
            acc_t[6] += t7 - t6; acc_t[7] += t8 - t7;
        }
    };
    for (int t = 0; t < T; t += 2) {
        phase(t, std::integral_constant<int, 0>{});
        if (t + 1 < T) phase(t + 1, std::integral_constant<int, 1>{});
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();            // pairs with the late half's extra barrier
    if (ABL == 9 && lane == 0 && blockIdx.x < 64) {
#pragma unroll
        for (int q = 0; q < 8; ++q) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + q] = acc_t[q];
        p.dbg[4096 + ((long)blockIdx.x * 8 + wave) * 2] = __builtin_amdgcn_s_memtime() - clk0;
        p.dbg[4096 + ((long)blockIdx.x * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int n = n0 + wn * 64 + jn * 16 + 4 * (lane >> 4);
            f32x4 v = hi[jn][i] + lo[jn][i] * (1.0f / 2048.0f);
            v *= *reinterpret_cast<const f32x4*>(p.cs + n);
            v += *reinterpret_cast<const f32x4*>(p.bias + n);
            *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double frand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand()); }

static const uint16_t* g_Ap[3];
template <int PRIO, int ABL>
static double run_pp(const P& p, int iters, const float* const* Abufs, int nbuf) {
    const int smem_b = NSTAGE_PP * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_pp<PRIO, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_b));
    const long tiles_m = (p.M + 127) / 128, tiles_n = p.N / 256;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    P q = p;
    for (int i = 0; i < 5; ++i) { q.A = Abufs[i % nbuf]; q.Ap = g_Ap[i % nbuf]; hipLaunchKernelGGL((gemm_f16x2_pp<PRIO, ABL>), dim3((unsigned)grid), dim3(512), smem_b, 0, q); }
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) { q.A = Abufs[i % nbuf]; q.Ap = g_Ap[i % nbuf]; hipLaunchKernelGGL((gemm_f16x2_pp<PRIO, ABL>), dim3((unsigned)grid), dim3(512), smem_b, 0, q); }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    q.A = Abufs[0]; q.Ap = g_Ap[0];
    hipLaunchKernelGGL((gemm_f16x2_pp<PRIO, ABL>), dim3((unsigned)grid), dim3(512), smem_b, 0, q);
    CK(hipDeviceSynchronize());
    return ms * 1e3 / iters;
}

template <int SHAPE, bool DEPHASE>
static double run(const P& p, int iters, const float* const* Abufs, int nbuf) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_wide<SHAPE, DEPHASE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE));
    const long tiles_m = (p.M + 127) / 128, tiles_n = p.N / 256;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    P q = p;
    for (int i = 0; i < 5; ++i) { q.A = Abufs[i % nbuf]; hipLaunchKernelGGL((gemm_f16x2_wide<SHAPE, DEPHASE>), dim3((unsigned)grid), dim3(512), 2 * STAGE, 0, q); }
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) { q.A = Abufs[i % nbuf]; hipLaunchKernelGGL((gemm_f16x2_wide<SHAPE, DEPHASE>), dim3((unsigned)grid), dim3(512), 2 * STAGE, 0, q); }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    q.A = Abufs[0];
    hipLaunchKernelGGL((gemm_f16x2_wide<SHAPE, DEPHASE>), dim3((unsigned)grid), dim3(512), 2 * STAGE, 0, q);   // leave out = f(Abufs[0])
    CK(hipDeviceSynchronize());
    return ms * 1e3 / iters;
}

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "d-vqvae_amd/libdvq_hip.so";
    void* lib = dlopen(libpath, RTLD_NOW);
    typedef int (*linear_t)(const dvq_gemm_src*, int, int64_t, int, const float*, int, float*, int64_t, dvq_stream_t);
    typedef int (*split_t)(const float*, int64_t, uint16_t*, dvq_stream_t);
    linear_t lin = lib ? (linear_t)dlsym(lib, "dvq_linear") : nullptr;
    split_t spl = lib ? (split_t)dlsym(lib, "dvq_split_bf16x3") : nullptr;
    if (!lin) printf("(no libdvq_hip.so: bf16x3 baseline skipped)\n");
    const long M = 16384;
    const int shapes[3][2] = {{1024, 1536}, {512, 2560}, {512, 1024}};
    const int only = argc > 2 ? atoi(argv[2]) : -1;             // shape filter (PMC runs)
    for (int si = 0; si < 3; ++si) {
        if (only >= 0 && si != only) continue;
        const int N = shapes[si][0], K = shapes[si][1];
        srand(1234 + si);
        std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hcs(N);
        for (auto& v : hA) v = (float)nrand();
        for (auto& v : hW) v = (float)(nrand() * 0.03);
        for (auto& v : hb) v = (float)nrand();
        std::vector<uint16_t> hp((size_t)2 * N * K);
        for (int n = 0; n < N; ++n) {
            float amax = 0;
            for (int k = 0; k < K; ++k) amax = fmaxf(amax, fabsf(hW[(size_t)n * K + k]));
            int e = 0;
            if (amax > 0) frexpf(amax, &e);
            const int t = amax > 0 ? 15 - e : 0;
            hcs[n] = ldexpf(1.0f, -t);
            for (int k = 0; k < K; ++k) {
                const float w = ldexpf(hW[(size_t)n * K + k], t);
                const _Float16 h1 = (_Float16)w;
                const float r = w - (float)h1;
                const _Float16 h2 = (_Float16)(r * 2048.0f);
                hp[(size_t)n * K + k] = *reinterpret_cast<const uint16_t*>(&h1);
                hp[(size_t)N * K + (size_t)n * K + k] = *reinterpret_cast<const uint16_t*>(&h2);
            }
        }
        const int nbuf = 3;
        float* dA[nbuf]; float *dW, *db, *dcs, *dout, *dout2; uint16_t *dp, *dp3;
        for (int i = 0; i < nbuf; ++i) { CK(hipMalloc(&dA[i], hA.size() * 4)); CK(hipMemcpy(dA[i], hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); }
        CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&db, N * 4)); CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dcs, N * 4)); CK(hipMemcpy(dcs, hcs.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&dp, hp.size() * 2)); CK(hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&dp3, (size_t)3 * N * K * 2));
        CK(hipMalloc(&dout, (size_t)M * N * 4)); CK(hipMalloc(&dout2, (size_t)M * N * 4));
        unsigned long long* ddbg; CK(hipMalloc(&ddbg, 8192 * 8)); CK(hipMemset(ddbg, 0, 8192 * 8));
        uint16_t* dAp[nbuf];
        for (int i = 0; i < nbuf; ++i) {
            CK(hipMalloc(&dAp[i], (size_t)M * K * 4));
            hipLaunchKernelGGL(presplit_rows, dim3((unsigned)((M * (K / 8) + 255) / 256)), dim3(256), 0, 0, dA[i], dAp[i], M, K);
        }
        CK(hipDeviceSynchronize());
        for (int i = 0; i < nbuf; ++i) g_Ap[i] = dAp[i];
        P p{dA[0], K, dp, (long)N * K, K, K, N, M, dcs, db, dout, N, ddbg, dAp[0]};
        const double flop = 2.0 * M * N * K;
        std::vector<float> hout((size_t)M * N);
        auto check = [&](const char* name, float* dev) {
            CK(hipMemcpy(hout.data(), dev, hout.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, worst_rel = 0;
            for (int s = 0; s < 4000; ++s) {
                const long m = s < 256 ? s * 64 + (s & 63) : (long)(frand() * (M - 1));
                const int n = s < 256 ? (s * 37) % N : (int)(frand() * (N - 1));
                double ref = hb[n], sc = 0;
                for (int k = 0; k < K; ++k) { const double t = (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k]; ref += t; sc += fabs(t); }
                const double err = fabs(hout[(size_t)m * N + n] - ref);
                if (err > worst) worst = err;
                if (err / sc > worst_rel) worst_rel = err / sc;
            }
            printf("   %-22s max |err| %.3e, max |err| / sum|a w| %.3e\n", name, worst, worst_rel);
        };
        printf("M=%ld N=%d K=%d\n", M, N, K);
        double us;
        us = run<16, false>(p, 40, dA, nbuf); printf("  f16x2 16x16x32          : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("16x16x32", dout);
        us = run<16, true>(p, 40, dA, nbuf);  printf("  f16x2 16x16x32 dephased : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("16x16x32 dephased", dout);
        us = run_pp<0, 0>(p, 40, dA, nbuf); printf("  f16x2 ping-pong         : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("ping-pong", dout);
        us = run_pp<1, 0>(p, 40, dA, nbuf); printf("  f16x2 ping-pong setprio : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("ping-pong setprio", dout);
        us = run_pp<2, 0>(p, 40, dA, nbuf); printf("  f16x2 pp, static priority for waves 4-7, no flips : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp static prio", dout);
        us = run_pp<1, 13>(p, 40, dA, nbuf); printf("  f16x2 pp setprio, activations as plane pairs by LDS-DMA : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp A planes", dout);
        us = run_pp<1, 13>(p, 40, dA, 1); printf("  f16x2 pp setprio, A plane pairs by DMA, ONE buffer (cache-resident) : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6);
        us = run_pp<1, 5>(p, 40, dA, nbuf); printf("  f16x2 pp setprio, split in the compute phase : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp split in C", dout);
        us = run_pp<0, 5>(p, 40, dA, nbuf); printf("  f16x2 pp (no setprio), split in the compute phase : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp split in C", dout);
        us = run_pp<1, 4>(p, 40, dA, nbuf); printf("  f16x2 pp setprio, activations two periods ahead : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp A x2", dout);
        us = run_pp<1, 0>(p, 40, dA, 1); printf("  f16x2 pp setprio, ONE activation buffer (cache-resident) : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6);
        {
            us = run_pp<1, 9>(p, 10, dA, nbuf); printf("  (stamped) pp setprio : %7.1f us\n", us);
            std::vector<unsigned long long> h(8192);
            CK(hipMemcpy(h.data(), ddbg, h.size() * 8, hipMemcpyDeviceToHost));
            const char* nm[8] = {"frag read issue", "wait vmcnt(0) (A loads, W DMA of last period)", "split + ds_write A", "DMA issue + A loads", "wait lgkmcnt(0)", "barrier L->C", "compute (48 MFMA)", "barrier C->L"};
            {
                double c = 0, r = 0; for (int i = 0; i < 512; ++i) { c += (double)h[4096 + 2 * i]; r += (double)h[4096 + 2 * i + 1]; }
                printf("    main loop: %.0f shader cycles in %.2f us per wave -> in-kernel clock %.0f MHz\n", c / 512, r / 512 * 0.01, c / r * 100.0);
            }
            const int steps = K / 32 - 5;
            for (int half = 0; half < 2; ++half) {
                printf("    waves %d..%d, cycles per K-tile:", 4 * half, 4 * half + 3);
                double tot = 0;
                for (int q = 0; q < 8; ++q) {
                    double sum = 0; for (int b = 0; b < 64; ++b) for (int w = 4 * half; w < 4 * half + 4; ++w) sum += (double)h[(b * 8 + w) * 8 + q];
                    printf(" [%s %.0f]", nm[q], sum / (64 * 4) / steps); tot += sum / (64 * 4) / steps;
                }
                printf(" total %.0f\n", tot);
            }
        }
        us = run_pp<1, 10>(p, 40, dA, nbuf); printf("  f16x2 pp setprio, counted waits (loads and DMA decoupled) : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp counted", dout);
        us = run_pp<1, 11>(p, 40, dA, nbuf); printf("  f16x2 pp setprio, counted waits, activations two periods ahead : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp counted x2", dout);
        us = run_pp<1, 12>(p, 40, dA, nbuf); printf("  f16x2 pp setprio, counted waits, loads before DMA, 1.5 periods : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("pp counted 12", dout);
        us = run_pp<1, 7>(p, 40, dA, nbuf); printf("  (timing only) pp setprio, activation loads but no split / LDS write : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6);
        us = run_pp<1, 8>(p, 40, dA, nbuf); printf("  (timing only) pp setprio, split / LDS write but no activation loads : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6);
        us = run_pp<1, 1>(p, 40, dA, nbuf); printf("  (timing only) pp setprio, no weight DMA : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6);
        us = run_pp<1, 2>(p, 40, dA, nbuf); printf("  (timing only) pp setprio, no activation path : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6);
        us = run<32, false>(p, 40, dA, nbuf); printf("  f16x2 32x32x16          : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("32x32x16", dout);
        us = run<32, true>(p, 40, dA, nbuf);  printf("  f16x2 32x32x16 dephased : %7.1f us  %6.1f TF\n", us, flop / us * 1e-6); check("32x32x16 dephased", dout);
        if (lin) {
            spl(dW, (int64_t)N * K, dp3, nullptr);
            dvq_gemm_src src{dA[0], dW, K, K, K, 0, dp3, (int64_t)N * K};
            for (int i = 0; i < 5; ++i) lin(&src, 1, M, N, db, 0, dout2, N, nullptr);
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            for (int i = 0; i < 40; ++i) { src.x = dA[i % nbuf]; lin(&src, 1, M, N, db, 0, dout2, N, nullptr); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            src.x = dA[0]; lin(&src, 1, M, N, db, 0, dout2, N, nullptr); CK(hipDeviceSynchronize());
            printf("  bf16x3 (library)        : %7.1f us  %6.1f TF\n", ms * 1e3 / 40, flop / (ms * 1e3 / 40) * 1e-6); check("bf16x3", dout2);
        }
        for (int i = 0; i < nbuf; ++i) { CK(hipFree(dA[i])); CK(hipFree(dAp[i])); }
        CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dcs)); CK(hipFree(dp)); CK(hipFree(dp3)); CK(hipFree(dout)); CK(hipFree(dout2));
    }
    return 0;
}
