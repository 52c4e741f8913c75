// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256): same indices as the exact
// kernel (vq.hip / oracle/vq_canonical.c), bit for bit, at a fraction of the fp32 contraction cost.
//
//   filter (this file, vq_filter_kernel): bf16 MFMA scores  s_k = ee_k + sum_j bf16(z_j) * bf16(-2 e_kj)
//       for all 512 entries of every row; candidates = { k : s_k <= min_k s_k + eps_row } with a PROVEN
//       eps_row (below), so the exact fp32 argmin is always among them.  A row with exactly one candidate
//       is decided; all others go to
//   refine (vq_refine_kernel): the canonical fp32 evaluation d_k = (zz + ee_k) - 2*dot_k (k-ordered fmaf
//       chains) of the <= 8 candidates, torch.argmin ordering (first minimum, NaN first).  Rows whose
//       candidate set is empty (NaN/Inf anywhere) or overflows are evaluated over all K entries.
//
// Error bound.  With u = 2^-9 (bf16 round-to-nearest), for every k
//   |s_k - (true_k - |z|^2)| <= 2 (2u + u^2) |z||e_k|  +  gamma_257 (|e_k|^2 + 2|z||e_k|)        (filter)
//   |d_k - true_k|           <= gamma_260 (|z| + |e_k|)^2                                          (exact side)
// hence for the exact winner k*:  s_k* <= min_k s_k + eps_row,
//   eps_row = 2^-6 (1 + 2^-9) |z| Emax + 2^-11 (|z| + Emax)^2      (gamma_n = n 2^-24, with >5x slack).
//
// Structure (one persistent 512-thread workgroup per CU, 2 waves per SIMD):
//   * wave w keeps the bf16 fragments of codebook entries [64w, 64w+64) in 128 VGPRs for the whole kernel
//     (A operand of v_mfma_f32_32x32x16_bf16: the codebook never goes through LDS again);
//   * z streams HBM -> LDS with global_load_lds (1 KiB row pieces, source-swizzled so that the B-fragment
//     ds_read_b128 are bank-conflict free), double buffered: tile t+1 is in flight while tile t computes;
//   * scores land with the z row on the lane (C layout: entry on registers, row on lanes), so min /
//     threshold scans are lane-local; waves exchange 64 floats per tile through LDS.
// Algorithmic HBM bytes per row: D*4 (z) + 8 (int64 index); the codebook (K*D*4) is read once.
#include "dvq_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 512, D = 256, TR = 32;          // entries, dims, rows per tile
constexpr int NBUF = 4;                            // LDS ring: up to 3 tiles in flight behind the one being consumed
constexpr int NW = K / 64;                         // 8 waves, 64 entries each
constexpr int KS = D / 16;                         // 16 MFMA k-steps
constexpr int MAXC = 8;                            // candidate slots per row

// LDS carve (bytes)
constexpr int OFF_Z = 0;                           // [NBUF][TR][D] fp32
constexpr int OFF_EE = NBUF * TR * D * 4;          // [K] fp32
constexpr int OFF_WMIN = OFF_EE + K * 4;           // [NW][TR]
constexpr int OFF_NPART = OFF_WMIN + NW * TR * 4;  // [NW][TR]
constexpr int OFF_CNT = OFF_NPART + NW * TR * 4;   // [TR] int
constexpr int OFF_CAND = OFF_CNT + TR * 4;         // [TR][MAXC] u16
constexpr int LDS_BYTES = OFF_CAND + TR * MAXC * 2;

struct PackHeader {
    float emax;        // upper bound of max_k |e_k|_2 (inf if the codebook is not finite)
    int K, D;
    int pad;
};
constexpr size_t PK_OFF_EE = 256;
constexpr size_t PK_OFF_FRAG = PK_OFF_EE + (size_t)K * 4;
constexpr size_t PK_BYTES = PK_OFF_FRAG + (size_t)K * D * 2;

// ------------------------------------------------------------------------------------------------ pack
// fragment order: [wave w][entry tile et][k-step s][lane][8 x bf16]  (1 KiB per (w, et, s))
__global__ void vq_pack_frag_kernel(const float* __restrict__ E, __bf16* __restrict__ frag) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (w, et, s, lane)
    if (gid >= NW * 2 * KS * 64) return;
    const int lane = gid & 63, s = (gid >> 6) % KS, et = (gid / (64 * KS)) & 1, w = gid / (64 * KS * 2);
    const int entry = 64 * w + 32 * et + (lane & 31);
    const int k0 = 16 * s + 8 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 8; ++j) frag[(size_t)gid * 8 + j] = (__bf16)(-2.0f * E[entry * D + k0 + j]);
}

__global__ void vq_pack_norm_kernel(const float* __restrict__ E, float* __restrict__ ee, PackHeader* hdr) {
    __shared__ float red[K];
    const int k = threadIdx.x;                                   // blockDim = K
    const float* p = E + k * D;
    float acc = 0.f;
    bool finite = true;
    for (int j = 0; j < D; ++j) {
        acc = fmaf(p[j], p[j], acc);                             // canonical chain (same as rownorm_kernel)
        finite = finite && (fabsf(p[j]) <= 3.0e38f);
    }
    ee[k] = acc;
    red[k] = (finite && acc <= 3.0e38f) ? acc : INFINITY;
    __syncthreads();
    for (int o = K / 2; o > 0; o >>= 1) {
        if (k < o) red[k] = fmaxf(red[k], red[k + o]);
        __syncthreads();
    }
    if (k == 0) {
        hdr->emax = sqrtf(red[0]) * 1.00001f;
        hdr->K = K;
        hdr->D = D;
        hdr->pad = 0;
    }
}

// ------------------------------------------------------------------------------------------------ filter
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One LDS-DMA (global_load_lds_dwordx4): 64 lanes x 16 B land at lds_dst + 16*lane.  Issued through inline asm so that
// hipcc does not track it: with the builtin it drains vmcnt(0) in front of every later LDS write/atomic (the DMA is a
// pending LDS write to it), which would kill the multi-tile prefetch.  Completion is waited for by hand (counted vmcnt).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ void issue_tile(const float* __restrict__ z, long M, long tile, unsigned zbuf_lds, int wave, int lane) {
    // wave w loads rows 4w..4w+3 of the tile; one 1 KiB LDS-DMA per row; LDS chunk `lane` <- global chunk lane ^ (row & 15)
#pragma unroll
    for (int i = 0; i < TR / NW; ++i) {
        const int row = wave * (TR / NW) + i;
        long grow = tile * TR + row;
        if (grow >= M) grow = M - 1;
        glds16(z + grow * D + 4 * (lane ^ (row & 15)), zbuf_lds + (unsigned)(row * D * 4));
    }
}

__global__ __launch_bounds__(512, 2) void vq_filter_kernel(const float* __restrict__ z, long M, const char* __restrict__ packed,
                                                           long n_tiles, int64_t* __restrict__ idx, uint16_t* __restrict__ cand_out,
                                                           uint8_t* __restrict__ cnt_out, int* __restrict__ amb_count,
                                                           uint8_t* __restrict__ amb_rows) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* zbuf = reinterpret_cast<float*>(lds + OFF_Z);
    float* ee_s = reinterpret_cast<float*>(lds + OFF_EE);
    float* wmin = reinterpret_cast<float*>(lds + OFF_WMIN);
    float* npart = reinterpret_cast<float*>(lds + OFF_NPART);
    int* cnt = reinterpret_cast<int*>(lds + OFF_CNT);
    uint16_t* cand = reinterpret_cast<uint16_t*>(lds + OFF_CAND);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const float emax = reinterpret_cast<const PackHeader*>(packed)->emax;
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    const uint4* frag_g = reinterpret_cast<const uint4*>(packed + PK_OFF_FRAG);

    const unsigned zbuf_lds = (unsigned)(uintptr_t)zbuf;        // low 32 bits of a flat LDS address = LDS offset
    const long first = blockIdx.x, stride = gridDim.x;
    if (first >= n_tiles) return;

    // codebook fragments of this wave: 2 entry tiles x 16 k-steps, resident for the whole kernel
    bf16x8 efrag[2][KS];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = frag_g[((wave * 2 + et) * KS + s) * 64 + lane];
            efrag[et][s] = __builtin_bit_cast(bf16x8, v);
        }
    ee_s[tid] = ee_g[tid];                       // K == blockDim
    if (tid < TR) cnt[tid] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // fragments + ee are in registers before any LDS-DMA is queued
    // prologue: NBUF-1 tiles in flight (always issue, clamped, so that the vmcnt arithmetic below is uniform)
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) {
        long t = first + p * stride;
        if (t >= n_tiles) t = n_tiles - 1;
        issue_tile(z, M, t, zbuf_lds + (unsigned)(p * TR * D * 4), wave, lane);
    }

    int slot_i = 0;
    for (long tile = first; tile < n_tiles; tile += stride, slot_i = (slot_i + 1) & (NBUF - 1)) {
        // each wave has (NBUF-1) tiles x 4 LDS-DMAs queued; the oldest tile must have landed: all but the 8 youngest
        // vector-memory operations done (global stores of the finalising wave only make this wait stricter)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        lds_barrier();                                           // ... for every wave; previous tile fully consumed
        {
            long t = tile + (NBUF - 1) * stride;
            if (t >= n_tiles) t = n_tiles - 1;                   // harmless re-load keeps the queue depth constant
            issue_tile(z, M, t, zbuf_lds + (unsigned)(((slot_i + NBUF - 1) & (NBUF - 1)) * TR * D * 4), wave, lane);
        }
        const float* zt = zbuf + slot_i * TR * D;

        f32x16 acc[2];                                           // [entry tile]; row = lane & 31
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int g = 0; g < 4; ++g) {                        // regs 4g..4g+3 <-> entries base + 8g + 4h + 0..3
                const f32x4 v = *reinterpret_cast<const f32x4*>(ee_s + 64 * wave + 32 * et + 8 * g + 4 * h);
                acc[et][4 * g] = v[0]; acc[et][4 * g + 1] = v[1]; acc[et][4 * g + 2] = v[2]; acc[et][4 * g + 3] = v[3];
            }
        float nrm = 0.f;
        const float* zrow = zt + r * D;
        const int sw = r & 15;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c0 = 4 * s + 2 * h;                        // 16-byte chunk index of floats [16s + 8h, +4)
            const f32x4 lo = *reinterpret_cast<const f32x4*>(zrow + 4 * (c0 ^ sw));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(zrow + 4 * ((c0 + 1) ^ sw));
            bf16x8 b;
            b[0] = (__bf16)lo[0]; b[1] = (__bf16)lo[1]; b[2] = (__bf16)lo[2]; b[3] = (__bf16)lo[3];
            b[4] = (__bf16)hi[0]; b[5] = (__bf16)hi[1]; b[6] = (__bf16)hi[2]; b[7] = (__bf16)hi[3];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(efrag[0][s], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(efrag[1][s], b, acc[1], 0, 0, 0);
            if (s == wave || s == wave + 8) {                    // each wave sums 1/8 of the row norms
                float t = nrm;
                t = fmaf(lo[0], lo[0], t); t = fmaf(lo[1], lo[1], t); t = fmaf(lo[2], lo[2], t); t = fmaf(lo[3], lo[3], t);
                t = fmaf(hi[0], hi[0], t); t = fmaf(hi[1], hi[1], t); t = fmaf(hi[2], hi[2], t); t = fmaf(hi[3], hi[3], t);
                nrm = t;
            }
        }
        // per-wave row minima (64 entries) and norm partials -> LDS
        {
            float mn = acc[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mn = fminf(mn, acc[0][e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) mn = fminf(mn, acc[1][e]);
            mn = fminf(mn, __shfl_xor(mn, 32));
            const float nn = nrm + __shfl_xor(nrm, 32);
            if (h == 0) {
                wmin[wave * TR + r] = mn;
                npart[wave * TR + r] = nn;
            }
        }
        lds_barrier();
        float thr;
        {
            float mn = wmin[r], n2 = npart[r];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                mn = fminf(mn, wmin[w * TR + r]);
                n2 += npart[w * TR + r];
            }
            const float zn = sqrtf(n2) * 1.0001f;
            const float eps = 0.0156556f * zn * emax + 0.000488282f * (zn + emax) * (zn + emax);
            thr = mn + eps;                                      // non-finite anywhere -> no candidate -> exact fallback
        }
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (acc[et][e] <= thr) {
                    const int slot = atomicAdd(&cnt[r], 1);
                    if (slot < MAXC) cand[r * MAXC + slot] = (uint16_t)(64 * wave + 32 * et + (e & 3) + 8 * (e >> 2) + 4 * h);
                }
        lds_barrier();
        if (wave == 0) {                                         // one lane per row finalises it
            const int row = r;
            const long grow = tile * TR + row;
            const int c = cnt[row];
            const bool live = (h == 0) && grow < M;
            const bool amb = live && c != 1;
            if (live) {
                idx[grow] = (c == 1) ? (int64_t)cand[row * MAXC] : (int64_t)-1;
                if (c != 1) {
                    cnt_out[grow] = (c >= 2 && c <= MAXC) ? (uint8_t)c : (uint8_t)255;
                    const uint4 v = *reinterpret_cast<const uint4*>(cand + row * MAXC);
                    *reinterpret_cast<uint4*>(cand_out + grow * MAXC) = v;
                }
            }
            const unsigned long long mask = __ballot(amb);
            if (amb) amb_rows[tile * TR + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint8_t)row;
            if (lane == 0) amb_count[tile] = __builtin_popcountll(mask);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (h == 0) cnt[row] = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // drain the clamped tail prefetches before the LDS goes away
}

// ------------------------------------------------------------------------------------------------ refine
__device__ __forceinline__ float chain_dot(const float* __restrict__ a, const float* __restrict__ b) {
    float acc = 0.f;
#pragma unroll 4
    for (int j = 0; j < D; j += 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(a + j);
        const f32x4 y = *reinterpret_cast<const f32x4*>(b + j);
        acc = fmaf(x[0], y[0], acc);
        acc = fmaf(x[1], y[1], acc);
        acc = fmaf(x[2], y[2], acc);
        acc = fmaf(x[3], y[3], acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void vq_refine_kernel(const float* __restrict__ z, const float* __restrict__ E,
                                                        const float* __restrict__ ee, long M, int64_t* __restrict__ idx,
                                                        const uint16_t* __restrict__ cand_out, const uint8_t* __restrict__ cnt_out,
                                                        const int* __restrict__ amb_count, const uint8_t* __restrict__ amb_rows) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const long tile = blockIdx.x;
    const int n_amb = amb_count[tile];
    if (n_amb == 0) return;
    const int tid = threadIdx.x, g = tid >> 3, j = tid & 7;
    bool any_overflow = false;
    for (int i0 = 0; i0 < n_amb; i0 += 32) {                      // 32 groups of 8 lanes, one row each
        const int i = i0 + g;
        float d = INFINITY;
        int k = 0x7fffffff;
        long grow = -1;
        if (i < n_amb) {
            grow = tile * TR + amb_rows[tile * TR + i];
            const int c = cnt_out[grow];
            if (c == 255) {
                any_overflow = true;
                grow = -1;
            } else if (j < c) {
                k = cand_out[grow * MAXC + j];
                const float* zr = z + grow * D;
                const float zz = chain_dot(zr, zr);
                const float dot = chain_dot(zr, E + (long)k * D);
                const float t = zz + ee[k];
                d = t - 2.0f * dot;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float od = __shfl_xor(d, o);
            const int ok = __shfl_xor(k, o);
            if (dvq_argmin_better(od, ok, d, k)) { d = od; k = ok; }
        }
        if (grow >= 0 && j == 0) idx[grow] = k;
    }
    if (!__syncthreads_or(any_overflow)) return;
    // rows without a usable candidate list (NaN/Inf, or > 8 candidates): all K entries, whole block per row
    for (int i = 0; i < n_amb; ++i) {
        const long grow = tile * TR + amb_rows[tile * TR + i];
        if (cnt_out[grow] != 255) continue;
        const float* zr = z + grow * D;
        const float zz = chain_dot(zr, zr);
        float bv = INFINITY;
        int bi = 0x7fffffff;
        for (int k = tid; k < K; k += 256) {
            const float dot = chain_dot(zr, E + (long)k * D);
            const float t = zz + ee[k];
            const float d = t - 2.0f * dot;
            if (dvq_argmin_better(d, k, bv, bi)) { bv = d; bi = k; }
        }
        s_v[tid] = bv;
        s_i[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o && dvq_argmin_better(s_v[tid + o], s_i[tid + o], s_v[tid], s_i[tid])) {
                s_v[tid] = s_v[tid + o];
                s_i[tid] = s_i[tid + o];
            }
            __syncthreads();
        }
        if (tid == 0) idx[grow] = s_i[0];
        __syncthreads();
    }
}

struct FastScratch {
    uint16_t* cand;
    uint8_t* cnt;
    int* amb_count;
    uint8_t* amb_rows;
    long n_tiles;
    size_t bytes;
};

FastScratch plan(int64_t M, void* ws) {
    FastScratch s;
    s.n_tiles = (M + TR - 1) / TR;
    char* p = (char*)ws;
    auto take = [&](size_t n) { char* q = p; p += dvq_round_up(n, 256); return q; };
    s.cand = (uint16_t*)take((size_t)M * MAXC * 2);
    s.cnt = (uint8_t*)take((size_t)M);
    s.amb_count = (int*)take((size_t)s.n_tiles * 4);
    s.amb_rows = (uint8_t*)take((size_t)s.n_tiles * TR);
    s.bytes = (size_t)(p - (char*)ws);
    return s;
}

int g_cus = 0;

}  // namespace

extern "C" int dvq_vq_fast_supported(int Kq, int Dq) { return Kq == K && Dq == D; }

extern "C" size_t dvq_vq_pack_bytes(int Kq, int Dq) { return dvq_vq_fast_supported(Kq, Dq) ? PK_BYTES : 0; }

extern "C" int dvq_vq_pack(const float* E, int Kq, int Dq, void* packed, size_t packed_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_pack: the fast path supports K=%d, D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(E && packed && dvq_aligned16(E) && dvq_aligned16(packed), "vq_pack: null/unaligned pointer");
    DVQ_REQUIRE(packed_bytes >= PK_BYTES, "vq_pack: buffer %zu < %zu bytes", packed_bytes, PK_BYTES);
    hipStream_t st = (hipStream_t)stream;
    char* pk = (char*)packed;
    hipLaunchKernelGGL(vq_pack_norm_kernel, dim3(1), dim3(K), 0, st, E, (float*)(pk + PK_OFF_EE), (PackHeader*)pk);
    DVQ_CHECK_LAUNCH("vq_pack_norm");
    const int n = NW * 2 * KS * 64;
    hipLaunchKernelGGL(vq_pack_frag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, E, (__bf16*)(pk + PK_OFF_FRAG));
    DVQ_CHECK_LAUNCH("vq_pack_frag");
    return DVQ_OK;
}

extern "C" size_t dvq_vq_fast_workspace_bytes(int64_t M, int Kq, int Dq) {
    if (M <= 0 || !dvq_vq_fast_supported(Kq, Dq)) return 256;
    return plan(M, nullptr).bytes;
}

extern "C" int dvq_vq_argmin_fast(const float* z, const float* E, const void* packed, int64_t M, int Kq, int Dq,
                                  int64_t* idx, void* workspace, size_t workspace_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_argmin_fast: supports K=%d, D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(M >= 0, "vq_argmin_fast: negative M");
    if (M == 0) return DVQ_OK;
    DVQ_REQUIRE(z && E && packed && idx && workspace, "vq_argmin_fast: null pointer");
    DVQ_REQUIRE(dvq_aligned16(z) && dvq_aligned16(E) && dvq_aligned16(packed) && dvq_aligned16(workspace),
                "vq_argmin_fast: pointers must be 16-byte aligned (z must be dense [M,256])");
    const FastScratch s = plan(M, workspace);
    if (workspace_bytes < s.bytes) {
        dvq_set_error("vq_argmin_fast: workspace %zu < %zu bytes", workspace_bytes, s.bytes);
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) {
            dvq_set_error("vq_argmin_fast: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cus = prop.multiProcessorCount;
        if (g_cus <= 0) g_cus = 256;
        attr_set = true;
    }
    const long grid = s.n_tiles < g_cus ? s.n_tiles : g_cus;
    const char* pk = (const char*)packed;
    DVQ_PROF("vq_argmin_total", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
    {
        DVQ_PROF("vq_filter", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 2 + (double)M * 8, st);
        hipLaunchKernelGGL(vq_filter_kernel, dim3((unsigned)grid), dim3(512), LDS_BYTES, st, z, (long)M, pk, s.n_tiles, idx,
                           s.cand, s.cnt, s.amb_count, s.amb_rows);
    }
    DVQ_CHECK_LAUNCH("vq_filter");
    {
        DVQ_PROF("vq_refine", 0, 0, st);
        hipLaunchKernelGGL(vq_refine_kernel, dim3((unsigned)s.n_tiles), dim3(256), 0, st, z, E,
                           (const float*)(pk + PK_OFF_EE), (long)M, idx, s.cand, s.cnt, s.amb_count, s.amb_rows);
    }
    DVQ_CHECK_LAUNCH("vq_refine");
    return DVQ_OK;
}
