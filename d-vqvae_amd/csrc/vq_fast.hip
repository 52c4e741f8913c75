// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256): same indices as the exact
// kernel (vq.hip / oracle/vq_canonical.c), bit for bit, at a fraction of the fp32 contraction cost.
//
//   filter (vq_filter_kernel): fp16 MFMA scores  s_k = ee_k + (sum_j h(sz z_j) h(-2 sE e_kj)) / (sz sE)
//       for all 512 entries of every row (h = round to fp16; sz per row and sE per codebook are powers of two
//       that put the largest magnitude in [2^13, 2^14)).  Candidates = { k : s_k <= min_k s_k + eps_row } with
//       a PROVEN eps_row (below), so the exact fp32 argmin is always among them.  A row with exactly one
//       candidate is decided; all others go to
//   refine (vq_refine_kernel): the canonical fp32 evaluation d_k = (zz + ee_k) - 2*dot_k (k-ordered fmaf
//       chains) of the <= 8 candidates, torch.argmin ordering (first minimum, NaN first).  Rows whose
//       candidate set is empty (NaN/Inf, magnitudes outside 2^+-40) or overflows are evaluated over all K.
//
// Error bound.  With u = 2^-12 (fp16 round-to-nearest; elements below the fp16 normal range add at most
// 2^-20 |z||e| in total, flushed or not), for every k
//   |s_k - (true_k - |z|^2)| <= 2 (2u + u^2 + 2^-20) |z||e_k| + gamma_258 (|e_k|^2 + 2|z||e_k|)   (filter)
//   |d_k - true_k|           <= gamma_260 (|z| + |e_k|)^2                                          (exact side)
// hence for the exact winner k*:  s_k* <= min_k s_k + eps_row,
//   eps_row = 2^-9 (1 + 2^-8) |z| Emax + 2^-13 (|z| + Emax)^2     (gamma_n = n 2^-24; > 2.5x slack on that term).
//
// Structure (one persistent 512-thread workgroup per CU, 2 waves per SIMD):
//   * wave w keeps the fp16 fragments of codebook entries [64w, 64w+64) in 128 VGPRs for the whole kernel
//     (A operand of v_mfma_f32_32x32x16_f16: the codebook never goes through LDS);
//   * z streams HBM -> LDS with global_load_lds (1 KiB row pieces, source-swizzled), 4-slot ring: three
//     32-row tiles are in flight behind the one being consumed;
//   * a pre-pass turns the landed fp32 tile into the scaled fp16 B-operand image (XOR-swizzled 16-byte chunks:
//     conflict-free ds_read_b128) and produces |z| and the row scale;
//   * scores land with the z row on the lane (C layout: entry on registers, row on lanes), so min /
//     threshold scans are lane-local; waves exchange 32 floats per tile through LDS.
// Algorithmic HBM bytes per row: D*4 (z) + 8 (int64 index); the codebook (K*D*4) is read once.
#include "dvq_internal.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int K = 512, D = 256, TR = 32;          // entries, dims, rows per tile
constexpr int NBUF = 4;                            // LDS ring: up to 3 tiles in flight behind the one being consumed
constexpr int NW = K / 64;                         // 8 waves, 64 entries each
constexpr int KS = D / 16;                         // 16 MFMA k-steps
constexpr int MAXC = 8;                            // candidate slots per row
constexpr int EXP_LIMIT = 40;                      // |log2(max magnitude)| beyond this -> exact fallback

// LDS carve (bytes)
constexpr int OFF_Z = 0;                           // [NBUF][TR][D] fp32 (LDS-DMA ring)
constexpr int OFF_H = NBUF * TR * D * 4;           // [TR][D] fp16 image of the current tile
constexpr int OFF_EE = OFF_H + TR * D * 2;         // [K] fp32
constexpr int OFF_WMIN = OFF_EE + K * 4;           // [NW][TR]
constexpr int OFF_INV = OFF_WMIN + NW * TR * 4;    // [TR] 1 / (sz sE)
constexpr int OFF_ZN = OFF_INV + TR * 4;           // [TR] |z| upper bound
constexpr int OFF_CNT = OFF_ZN + TR * 4;           // [TR] int
constexpr int OFF_CAND = OFF_CNT + TR * 4;         // [TR][MAXC] u16
constexpr int LDS_BYTES = OFF_CAND + TR * MAXC * 2;

struct PackHeader {
    float emax;        // upper bound of max_k |e_k|_2 (inf if the codebook is not finite)
    int sexp;          // codebook scale sE = 2^sexp
    int valid;         // 0: codebook magnitudes outside the filter's range -> every row takes the exact path
    int K, D;
};
constexpr size_t PK_OFF_EE = 256;
constexpr size_t PK_OFF_FRAG = PK_OFF_EE + (size_t)K * 4;
constexpr size_t PK_BYTES = PK_OFF_FRAG + (size_t)K * D * 2;

__device__ __forceinline__ float pow2f(int e) { return __int_as_float((e + 127) << 23); }   // e in [-126, 127]

// ------------------------------------------------------------------------------------------------ pack
__global__ void vq_pack_norm_kernel(const float* __restrict__ E, float* __restrict__ ee, PackHeader* hdr) {
    __shared__ float red[K];
    __shared__ float redm[K];
    const int k = threadIdx.x;                                   // blockDim = K
    const float* p = E + k * D;
    float acc = 0.f, mx = 0.f;
    bool finite = true;
    for (int j = 0; j < D; ++j) {
        acc = fmaf(p[j], p[j], acc);                             // canonical chain (same as rownorm_kernel)
        finite = finite && (fabsf(p[j]) <= 3.0e38f);
        mx = fmaxf(mx, fabsf(p[j]));
    }
    ee[k] = acc;
    red[k] = (finite && acc <= 3.0e38f) ? acc : INFINITY;
    redm[k] = finite ? mx : INFINITY;
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
        const float m2 = 2.0f * redm[0];                         // the fragments hold -2 e
        const int e = (int)((__float_as_uint(m2) >> 23) & 0xff) - 127;
        const bool ok = redm[0] > 0.f && e >= -EXP_LIMIT && e <= EXP_LIMIT && red[0] <= 3.0e38f;
        hdr->sexp = ok ? 13 - e : 0;
        hdr->valid = ok ? 1 : 0;
        hdr->K = K;
        hdr->D = D;
    }
}

// fragment order: [wave w][entry tile et][k-step s][lane][8 x fp16]  (1 KiB per (w, et, s))
__global__ void vq_pack_frag_kernel(const float* __restrict__ E, const PackHeader* __restrict__ hdr, _Float16* __restrict__ frag) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;       // one thread per (w, et, s, lane)
    if (gid >= NW * 2 * KS * 64) return;
    const int lane = gid & 63, s = (gid >> 6) % KS, et = (gid / (64 * KS)) & 1, w = gid / (64 * KS * 2);
    const int entry = 64 * w + 32 * et + (lane & 31);
    const int k0 = 16 * s + 8 * (lane >> 5);
    const float sc = -2.0f * pow2f(hdr->sexp);
#pragma unroll
    for (int j = 0; j < 8; ++j) frag[(size_t)gid * 8 + j] = (_Float16)(sc * E[entry * D + k0 + j]);
}

// ------------------------------------------------------------------------------------------------ filter
// all-reduce over the 16 lanes of a DPP row at VALU speed (no LDS crossbar): xor 1, xor 2 (quad_perm), then the
// half-mirror and mirror permutations fold quads and halves together (max / sum are order-insensitive here)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));     // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E>(v));     // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f<0x141>(v));    // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));    // row_mirror
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    v += dpp_f<0x140>(v);
    return v;
}

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

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define STAMP(i) do { if (ABL == 9) { const unsigned long long n__ = stamp(); ph[i] += n__ - t_prev; t_prev = n__; } } while (0)

template <int ABL>
__global__ __launch_bounds__(512, 2) void vq_filter_kernel(const float* __restrict__ z, long M, const char* __restrict__ packed,
                                                           long n_tiles, int64_t* __restrict__ idx, uint16_t* __restrict__ cand_out,
                                                           uint8_t* __restrict__ cnt_out, int* __restrict__ amb_count,
                                                           int* __restrict__ amb_list, int seg_cap, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = (ABL == 9) ? stamp() : 0;
    float* zbuf = reinterpret_cast<float*>(lds + OFF_Z);
    _Float16* himg = reinterpret_cast<_Float16*>(lds + OFF_H);
    float* ee_s = reinterpret_cast<float*>(lds + OFF_EE);
    float* wmin = reinterpret_cast<float*>(lds + OFF_WMIN);
    float* inv_s = reinterpret_cast<float*>(lds + OFF_INV);
    float* zn_s = reinterpret_cast<float*>(lds + OFF_ZN);
    int* cnt = reinterpret_cast<int*>(lds + OFF_CNT);
    uint16_t* cand = reinterpret_cast<uint16_t*>(lds + OFF_CAND);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    const float emax = hdr->emax;
    const int e_sexp = hdr->sexp;
    const bool e_valid = hdr->valid != 0;
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    const uint4* frag_g = reinterpret_cast<const uint4*>(packed + PK_OFF_FRAG);

    const unsigned zbuf_lds = (unsigned)(uintptr_t)zbuf;        // low 32 bits of a flat LDS address = LDS offset
    const long first = blockIdx.x, stride = gridDim.x;
    if (first >= n_tiles) return;

    // prologue: NBUF-1 tiles in flight (always issue, clamped, so that the vmcnt arithmetic below is uniform) ...
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) {
        long t = first + p * stride;
        if (t >= n_tiles) t = n_tiles - 1;
        if (ABL < 3) issue_tile(z, M, t, zbuf_lds + (unsigned)(p * TR * D * 4), wave, lane);
    }
    // ... while the codebook fragments of this wave (2 entry tiles x 16 k-steps) stream from L2 into registers
    f16x8 efrag[2][KS];
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = frag_g[((wave * 2 + et) * KS + s) * 64 + lane];
            efrag[et][s] = __builtin_bit_cast(f16x8, v);
        }
    ee_s[tid] = ee_g[tid];                       // K == blockDim
    if (tid < TR) cnt[tid] = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(0);

    int n_seg = 0;                                              // ambiguous rows of this workgroup so far (wave 0)
    int slot_i = 0;
    for (long tile = first; tile < n_tiles; tile += stride, slot_i = (slot_i + 1) & (NBUF - 1)) {
        // each wave has (NBUF-1) tiles x 4 LDS-DMAs queued; the oldest tile must have landed: all but the 8 youngest
        // vector-memory operations done (global stores of the finalising wave only make this wait stricter)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        lds_barrier();                                           // ... for every wave; previous tile fully consumed
        STAMP(1);
        {
            long t = tile + (NBUF - 1) * stride;
            if (t >= n_tiles) t = n_tiles - 1;                   // harmless re-load keeps the queue depth constant
            if (ABL < 3) issue_tile(z, M, t, zbuf_lds + (unsigned)(((slot_i + NBUF - 1) & (NBUF - 1)) * TR * D * 4), wave, lane);
        }
        STAMP(2);
        // ---- pre-pass: 16 lanes per row, 16 floats per lane: max |z|, sum z^2, then the scaled fp16 image
        {
            const int prow = wave * 4 + (lane >> 4), q = lane & 15, sw = prow & 15;
            const float* src = zbuf + slot_i * TR * D + prow * D;
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + 4 * ((4 * q + i) ^ sw));
            float mx = 0.f, ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    mx = fmaxf(mx, fabsf(v[i][c]));
                    ss = fmaf(v[i][c], v[i][c], ss);
                }
            mx = row16_max(mx);
            ss = row16_sum(ss);
            const bool bad = !(ss <= 3.0e38f);                  // NaN / Inf anywhere in the row poisons the sum of squares
            const int ez = (int)((__float_as_uint(mx) >> 23) & 0xff) - 127;
            const bool ok = e_valid && !bad && (mx == 0.f || (ez >= -EXP_LIMIT && ez <= EXP_LIMIT));
            const int zs = (mx == 0.f || !ok) ? 0 : 13 - ez;
            const float sc = pow2f(zs);
            _Float16* dst = himg + prow * D;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f16x8 o8;
                o8[0] = (_Float16)(v[2 * i][0] * sc); o8[1] = (_Float16)(v[2 * i][1] * sc);
                o8[2] = (_Float16)(v[2 * i][2] * sc); o8[3] = (_Float16)(v[2 * i][3] * sc);
                o8[4] = (_Float16)(v[2 * i + 1][0] * sc); o8[5] = (_Float16)(v[2 * i + 1][1] * sc);
                o8[6] = (_Float16)(v[2 * i + 1][2] * sc); o8[7] = (_Float16)(v[2 * i + 1][3] * sc);
                *reinterpret_cast<f16x8*>(dst + 8 * ((2 * q + i) ^ sw)) = o8;
            }
            if (q == 0) {
                inv_s[prow] = ok ? pow2f(-(zs + e_sexp)) : __int_as_float(0x7fc00000);   // NaN marks "exact path"
                zn_s[prow] = __builtin_amdgcn_sqrtf(ss) * 1.0001f;          // v_sqrt_f32 (1 ulp) with 1e-4 head-room
            }
        }
        STAMP(3);
        lds_barrier();
        STAMP(4);

        f32x16 acc[2];                                           // [entry tile]; row = lane & 31
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[et][e] = 0.f;
        const _Float16* hrow = himg + r * D;
        const int sw = r & 15;
        if (ABL < 2)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 b = *reinterpret_cast<const f16x8*>(hrow + 8 * ((2 * s + h) ^ sw));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(efrag[0][s], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(efrag[1][s], b, acc[1], 0, 0, 0);
        }
        STAMP(5);
        // scores s_k = ee_k + acc / (sz sE); per-wave row minima (64 entries) -> LDS
        {
            const float inv = inv_s[r];
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                    // regs 4g..4g+3 <-> entries base + 8g + 4h + 0..3
                    const f32x4 ev = *reinterpret_cast<const f32x4*>(ee_s + 64 * wave + 32 * et + 8 * g + 4 * h);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[et][4 * g + i] = fmaf(acc[et][4 * g + i], inv, ev[i]);
                }
            float mn = acc[0][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) mn = fminf(mn, acc[0][e]);
#pragma unroll
            for (int e = 0; e < 16; ++e) mn = fminf(mn, acc[1][e]);
            mn = fminf(mn, __shfl_xor(mn, 32));
            if (h == 0) wmin[wave * TR + r] = mn;
        }
        STAMP(6);
        lds_barrier();
        STAMP(7);
        {
            float mn = wmin[r];
#pragma unroll
            for (int w = 1; w < NW; ++w) mn = fminf(mn, wmin[w * TR + r]);
            const float zn = zn_s[r];
            const float eps = 0.00196076f * zn * emax + 0.00012208f * (zn + emax) * (zn + emax);
            const float thr = mn + eps + (inv_s[r] - inv_s[r]);  // NaN scale (exact path) poisons the threshold
            unsigned hits = 0;
            if (ABL < 1)
#pragma unroll
            for (int et = 0; et < 2; ++et)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool hit = acc[et][e] <= thr;
                    if (__ballot(hit)) hits |= hit ? (1u << (16 * et + e)) : 0u;      // wave-uniform skip: hits are rare
                }
            if (hits) {                                          // ~1 candidate per row over the 8 waves
                int slot = atomicAdd(&cnt[r], __builtin_popcount(hits));
                while (hits) {
                    const int bit = __builtin_ctz(hits);
                    hits &= hits - 1;
                    const int e = bit & 15, et = bit >> 4;
                    if (slot < MAXC) cand[r * MAXC + slot] = (uint16_t)(64 * wave + 32 * et + (e & 3) + 8 * (e >> 2) + 4 * h);
                    ++slot;
                }
            }
        }
        STAMP(8);
        lds_barrier();
        STAMP(9);
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
            // append the ambiguous rows to THIS workgroup's segment of the work list (no global atomics: one shared
            // counter saturates at ~88 atomics/us and 256 workgroups arrive together every tile)
            const unsigned long long mask = __ballot(amb);
            if (amb) amb_list[(long)blockIdx.x * seg_cap + n_seg + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (int)grow;
            n_seg += __builtin_popcountll(mask);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (h == 0) cnt[row] = 0;
        }
        STAMP(10);
    }
    if (wave == 0 && lane == 0) amb_count[blockIdx.x] = n_seg;
    if (ABL == 9 && lane == 0 && (blockIdx.x == 0 || blockIdx.x == 131))
        for (int i = 0; i < 12; ++i) dbg[((blockIdx.x ? 1 : 0) * 8 + wave) * 12 + i] = ph[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // drain the clamped tail prefetches before the LDS goes away
}

// ------------------------------------------------------------------------------------------------ refine
// Canonical chains threaded through G = 4 lanes: lane q of a group holds floats [64q, 64q+64) of its z row and of its
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

// full-row single-lane form (used by the all-K fallback)
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

__global__ __launch_bounds__(256) void vq_refine_kernel(const float* __restrict__ z, const float* __restrict__ E,
                                                        const float* __restrict__ ee, long M, int64_t* __restrict__ idx,
                                                        const uint16_t* __restrict__ cand_out, const uint8_t* __restrict__ cnt_out,
                                                        const int* __restrict__ amb_count, const int* __restrict__ amb_list_all,
                                                        int seg_cap) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int seg = blockIdx.x >> 3, part = blockIdx.x & 7;     // 8 blocks share one filter workgroup's segment
    const int n_amb = amb_count[seg];
    const int* amb_list = amb_list_all + (long)seg * seg_cap;
    const int tid = threadIdx.x;
    const int g = tid >> 5;              // 8 rows per block pass, 32 lanes each: 8 candidate slots x 4 lanes
    const int j = (tid >> 2) & 7;        // candidate slot
    const int q = tid & 3;               // quarter of the row
    for (int i0 = part * 8; i0 < n_amb; i0 += 64) {
        const int i = i0 + g;
        float d = INFINITY;
        int k = 0x7fffffff;
        long grow = -1;
        bool work = false, overflow = false;
        if (i < n_amb) {
            grow = amb_list[i];
            const int c = cnt_out[grow];
            if (c == 255) overflow = true;
            else if (j < c) {
                k = cand_out[grow * MAXC + j];
                work = true;
            }
        }
        // every lane runs the (shuffling) chain code; lanes without work load nothing and are ignored
        float zz, dot;
        chain_pair_x4(z + (work ? grow : 0) * D, E + (long)(work ? k : 0) * D, q, work, zz, dot);
        if (work) {
            const float t = zz + ee[k];
            d = t - 2.0f * dot;
        } else {
            k = 0x7fffffff;
        }
#pragma unroll
        for (int o = 4; o < 32; o <<= 1) {                        // over the 8 candidates (lanes 4 apart)
            const float od = __shfl_xor(d, o);
            const int ok = __shfl_xor(k, o);
            if (dvq_argmin_better(od, ok, d, k)) { d = od; k = ok; }
        }
        if (grow >= 0 && !overflow && (tid & 31) == 0) idx[grow] = k;
        if (!__syncthreads_or(overflow)) continue;
        // rows without a usable candidate list (NaN/Inf, out-of-range magnitudes, > 8 candidates): all K entries,
        // the whole block per row
        for (int u = 0; u < 8; ++u) {
            const int iu = i0 + u;
            if (iu >= n_amb) break;
            const long gr = amb_list[iu];
            if (cnt_out[gr] != 255) continue;
            const float* zr = z + gr * D;
            float bv = INFINITY;
            int bi = 0x7fffffff;
            for (int kk = tid; kk < K; kk += 256) {
                float zz2, dot2;
                chain_pair(zr, E + (long)kk * D, zz2, dot2);
                const float t = zz2 + ee[kk];
                const float dd = t - 2.0f * dot2;
                if (dvq_argmin_better(dd, kk, bv, bi)) { bv = dd; bi = kk; }
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
            if (tid == 0) idx[gr] = s_i[0];
            __syncthreads();
        }
    }
}

struct FastScratch {
    uint16_t* cand;
    uint8_t* cnt;
    int* amb_count;
    int* amb_list;
    char* dbg;
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
    s.amb_count = (int*)take(1024 * 4);
    s.amb_list = (int*)take(((size_t)M + 1024 * TR) * 4);
    s.dbg = take(16 * 12 * 8);
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
    hipLaunchKernelGGL(vq_pack_frag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, E, (const PackHeader*)pk,
                       (_Float16*)(pk + PK_OFF_FRAG));
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
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel<0>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
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
    DVQ_REQUIRE(M < (1L << 31), "vq_argmin_fast: M too large");
    DVQ_REQUIRE(grid <= 1024, "vq_argmin_fast: unexpected CU count %ld", grid);
    const int seg_cap = (int)((s.n_tiles + grid - 1) / grid) * TR;
    DVQ_PROF("vq_argmin_total", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
    {
        DVQ_PROF("vq_filter", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 2 + (double)M * 8, st);
        const char* ab = getenv("DVQ_VQ_ABL");
        const int abl = ab ? atoi(ab) : 0;
#define LAUNCH_F(A) hipLaunchKernelGGL(vq_filter_kernel<A>, dim3((unsigned)grid), dim3(512), LDS_BYTES, st, z, (long)M, pk, s.n_tiles, idx, s.cand, s.cnt, s.amb_count, s.amb_list, seg_cap, (unsigned long long*)s.dbg)
        if (abl == 1) LAUNCH_F(1); else if (abl == 2) LAUNCH_F(2); else if (abl == 3) LAUNCH_F(3); else if (abl == 9) LAUNCH_F(9); else LAUNCH_F(0);
#undef LAUNCH_F
    }
    DVQ_CHECK_LAUNCH("vq_filter");
    {
        DVQ_PROF("vq_refine", 0, 0, st);
        hipLaunchKernelGGL(vq_refine_kernel, dim3((unsigned)(grid * 8)), dim3(256), 0, st, z, E,
                           (const float*)(pk + PK_OFF_EE), (long)M, idx, s.cand, s.cnt, s.amb_count, s.amb_list, seg_cap);
    }
    DVQ_CHECK_LAUNCH("vq_refine");
    return DVQ_OK;
}
