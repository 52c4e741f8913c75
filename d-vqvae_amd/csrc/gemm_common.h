// Epilogues shared by the fp32-MFMA and the split-bf16 GEMM kernels.  Both use 32x32 MFMA tiles whose C/D layout is
// dtype-independent on gfx950: acc[i][jn][e] holds row m = m0 + wm*64 + i*32 + (e&3) + 8*(e>>2) + 4*h and
// column n = n0 + wn*64 + jn*32 + r  (wave = 2*wm + wn, r = lane & 31, h = lane >> 5).
#pragma once
#include "dvq_internal.h"

constexpr int GEMM_BM = 128, GEMM_BN = 128;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[2][2], long m0, int n0, long mt, int nt,
                                              int tid, float* smem) {
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // ------------------------------------------------------------------ epilogues
    // acc[i][jn][e]: row m = m0 + wm*64 + i*32 + (e&3) + 8*(e>>2) + 4*h ; col n = n0 + wn*64 + jn*32 + r
    if constexpr (EPI == EPI_BIAS || EPI == EPI_RESID) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int n = n0 + wn * 64 + jn * 32 + r;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (m >= p.M) continue;
                    float v = acc[i][jn][e] + bv;
                    if constexpr (EPI == EPI_RESID) v += p.resid[m * p.ldr + n];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.out[m * p.ldo + n] = v;
                }
        }
    } else if constexpr (EPI == EPI_GATE) {
        // gate-packed channels: jn = 0 holds the tanh half, jn = 1 its sigmoid partner
        const int na = n0 + wn * 64 + r;        // packed index of the tanh channel
        const int nb = na + 32;                 // packed index of the sigmoid partner
        const int c = nt * 64 + wn * 32 + r;    // natural output channel
        const float ba = p.bias ? p.bias[na] : 0.f;
        const float bb = p.bias ? p.bias[nb] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float a = acc[i][0][e] + ba;
                float g = acc[i][1][e] + bb;
                if (p.pre) {
                    p.pre[m * p.ldpre + na] = a;
                    p.pre[m * p.ldpre + nb] = g;
                }
                if (p.cls) {
                    const float* crow = p.cls + (long)p.label[m] * p.N;
                    a += crow[na];
                    g += crow[nb];
                }
                p.out[m * p.ldo + c] = tanhf(a) * sigmoidf_(g);
            }
    } else if constexpr (EPI == EPI_COLMAX) {
        __syncthreads();                         // everyone is done with the staging buffers
        float* red = smem;                       // [2][128]
        const int row_base = (int)(m0 % p.rows_per_group);   // a 128-row tile never straddles two groups
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int n = n0 + wn * 64 + jn * 32 + r;
            const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.f;
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const bool valid = (m < p.M) && (row_base + (int)(m - m0) < p.valid_rows);
                    float v = acc[i][jn][e] + bv;
                    if (p.relu) v = fmaxf(v, 0.f);
                    mx = valid ? fmaxf(mx, v) : mx;
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (h == 0) red[wm * 128 + wn * 64 + jn * 32 + r] = mx;
        }
        __syncthreads();
        if (tid < 128) {
            const int n = n0 + tid;
            if (n < p.N) p.partial[mt * p.N + n] = fmaxf(red[tid], red[128 + tid]);
        }
    } else if constexpr (EPI == EPI_ARGMIN) {
        __syncthreads();
        float* red_v = smem;                                  // [2][128]
        int* red_i = reinterpret_cast<int*>(smem + 256);      // [2][128]
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int n = n0 + wn * 64 + jn * 32 + r;
            const float zz = (n < p.N) ? p.col_norm[n] : 0.f;
            float bv = INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long k = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (k >= p.M) continue;
                    const float t = zz + p.row_norm[k];
                    const float d = t - 2.0f * acc[i][jn][e];
                    if (dvq_argmin_better(d, (int)k, bv, bi)) { bv = d; bi = (int)k; }
                }
            const float ov = __shfl_xor(bv, 32);
            const int oi = __shfl_xor(bi, 32);
            if (dvq_argmin_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            if (h == 0) {
                red_v[wm * 128 + wn * 64 + jn * 32 + r] = bv;
                red_i[wm * 128 + wn * 64 + jn * 32 + r] = bi;
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int n = n0 + tid;
            if (n < p.N) {
                float bv = red_v[tid];
                int bi = red_i[tid];
                if (dvq_argmin_better(red_v[128 + tid], red_i[128 + tid], bv, bi)) { bv = red_v[128 + tid]; bi = red_i[128 + tid]; }
                p.part_val[mt * p.N + n] = bv;
                p.part_idx[mt * p.N + n] = bi;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed accumulator layout (MFMA called with the weight fragment as operand A): lanes <-> rows m, registers <-> columns n.
// acc[i][jn][e] holds row m = m0 + wm*64 + i*32 + r and column n = n0 + wn*64 + jn*32 + (e&3) + 8*(e>>2) + 4*h, so registers
// 4g..4g+3 are FOUR CONSECUTIVE columns of one row: every store is a 16-byte store (the default layout needs 4x as many
// 4-byte store instructions, and the store tail is issue-bound).  Store epilogues only (bias / residual / gate).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_t_at(const GemmParams& p, f32x16 (&acc)[2][2], long m0, int n0, int tid, int wm, int wn);

template <int EPI>
__device__ __forceinline__ void gemm_epilogue_t(const GemmParams& p, f32x16 (&acc)[2][2], long m0, int n0, int nt, int tid) {
    const int wave = tid >> 6;
    gemm_epilogue_t_at<EPI>(p, acc, m0, n0, tid, wave >> 1, wave & 1);      // 128 x 128 tile, waves 2 x 2
}

// the same for any wave grid: the wave owns rows m0 + wm*64 .. +63 and columns n0 + wn*64 .. +63 (n0 a multiple of 128)
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_t_at(const GemmParams& p, f32x16 (&acc)[2][2], long m0, int n0, int tid, int wm, int wn) {
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const bool vec_ok = (p.N % 4 == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long m = m0 + wm * 64 + i * 32 + r;
        if (m >= p.M) continue;
        if constexpr (EPI == EPI_BIAS || EPI == EPI_RESID) {
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * 64 + jn * 32 + 8 * g + 4 * h;
                    if (n >= p.N) continue;
                    f32x4 v = {acc[i][jn][4 * g], acc[i][jn][4 * g + 1], acc[i][jn][4 * g + 2], acc[i][jn][4 * g + 3]};
                    const bool full = vec_ok && n + 3 < p.N && (p.ldo % 4 == 0);
                    if (full) {
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if constexpr (EPI == EPI_RESID) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.ldr + n);
                        if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                        *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (n + c >= p.N) continue;
                            float x = v[c] + (p.bias ? p.bias[n + c] : 0.f);
                            if constexpr (EPI == EPI_RESID) x += p.resid[m * p.ldr + n + c];
                            if (p.relu) x = fmaxf(x, 0.f);
                            p.out[m * p.ldo + n + c] = x;
                        }
                    }
                }
        } else if constexpr (EPI == EPI_GATE) {
            const float* crow = p.cls ? p.cls + (long)p.label[m] * p.N : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 8 * g + 4 * h;
                const int na = n0 + wn * 64 + nl;        // gate-packed index of the tanh channels
                const int nb = na + 32;                  // ... of their sigmoid partners
                const int c = (n0 >> 1) + wn * 32 + nl;  // natural output channels
                f32x4 a = {acc[i][0][4 * g], acc[i][0][4 * g + 1], acc[i][0][4 * g + 2], acc[i][0][4 * g + 3]};
                f32x4 gg = {acc[i][1][4 * g], acc[i][1][4 * g + 1], acc[i][1][4 * g + 2], acc[i][1][4 * g + 3]};
                if (p.bias) {
                    a += *reinterpret_cast<const f32x4*>(p.bias + na);
                    gg += *reinterpret_cast<const f32x4*>(p.bias + nb);
                }
                if (p.pre) {
                    *reinterpret_cast<f32x4*>(p.pre + m * p.ldpre + na) = a;
                    *reinterpret_cast<f32x4*>(p.pre + m * p.ldpre + nb) = gg;
                }
                if (crow) {
                    a += *reinterpret_cast<const f32x4*>(crow + na);
                    gg += *reinterpret_cast<const f32x4*>(crow + nb);
                }
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = tanhf(a[q]) * sigmoidf_(gg[q]);
                *reinterpret_cast<f32x4*>(p.out + m * p.ldo + c) = o;
            }
        }
    }
}
