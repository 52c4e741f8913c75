// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256): same indices as the exact
// kernel (vq.hip / oracle/vq_canonical.c), bit for bit, at a fraction of the fp32 contraction cost.
//
//   filter (vq_filter_kernel): fp16 MFMA scores  s_k = ee_k + (sum_j h(sz z_j) h(-2 sE e_kj)) / (sz sE)
//       for all 512 entries of every row (h = round to fp16; sz per row and sE per codebook are powers of two
//       that put the largest magnitude in [2^13, 2^14)).  Every lane keeps the FOUR smallest scores it has seen
//       (entry id packed into the low 8 mantissa bits); a row's candidates are the scores within a PROVEN eps_row
//       of the row minimum, so the exact fp32 argmin is always among them.  One candidate -> decided; else
//   refine (same kernel, same workgroup): the canonical fp32 evaluation d_k = (zz + ee_k) - 2*dot_k (k-ordered fmaf
//       chains) of the <= 6 candidates, torch.argmin ordering (first minimum, NaN first).  Rows whose candidate
//       set may be incomplete (a lane's fourth-smallest score is still within eps) pass a second-level filter
//       (plain fp32 distances of all K entries, threshold 4 gamma_260 (|z| + Emax)^2) and evaluate its <= 64
//       survivors; rows with an empty set (NaN/Inf, magnitudes outside 2^+-40) or more survivors are evaluated
//       canonically over all K entries.
//
// Error bound.  The fp16 rounding errors are MEASURED, not bounded a priori: with dz_j = z_j - h(sz z_j)/sz and
// de_kj = e_kj - h(-2 sE e_kj)/(-2 sE) (both exact in fp32; underflow and flushes included), the products
// h.h are exact in the MFMA's fp32, so for every k
//   |s_k - (true_k - |z|^2)| <= 2 (|dz||e_k| + |z||de_k| + |dz||de_k|) + gamma_258' (|e_k|^2 + 2|z||e_k|)   (filter)
//   |d_k - true_k|           <= gamma_260 (|z| + |e_k|)^2                                                  (exact side)
//   |packed(s_k) - s_k|      <= 2^-15 |s_k| <= 2^-15 (|z| + |e_k|)^2                                  (id in the mantissa)
// (gamma_n = n 2^-24; gamma' allows a truncating accumulator, n 2^-23), hence for the exact winner k*:
//   packed(s_k*) <= min_k packed(s_k) + eps_row,
//   eps_row = 4 (|dz| Emax + |z| dEmax + |dz| dEmax) + (2^-13 + 2^-14) (|z| + Emax)^2,
// |dz| per row from the kernel's own conversion, dEmax = max_k |de_k| from dvq_vq_pack, norms rounded up.  (The a-priori
// bound would be |dz| <= 2^-11 |z|: eps = 2^-8 |z| Emax + ...; the measured norms are ~0.4 of that.)
//
// Structure (the fused PointNet trunk's, pointnet.hip): a wave owns 32 rows; their fp16 fragments (64 VGPRs) are
// the MFMA B operand for the whole kernel; the codebook image (256 KB, L2-resident) streams L2 -> LDS by
// global_load_lds in 32 KB chunks of 64 entries (XOR-swizzled source, conflict-free ds_read_b128), double buffered,
// one barrier per chunk; scores land with the row on the lane and the entry on the registers, so the top-4 tracking
// is lane-local VALU work interleaved with the next chunk's MFMAs.  128 rows per 256-thread workgroup, two
// workgroups per CU.  z is read from HBM exactly once (whole 1 KiB lines by LDS-DMA, then fragments to registers);
// ambiguous rows re-read their 1 KiB (L2/MALL) in the refine tail.
// Algorithmic HBM bytes per row: D*4 (z) + 8 (int64 index); the codebook (K*D*4) is read once.
#include "dvq_internal.h"

#ifndef DVQ_PF
#define DVQ_PF 2      // fragment prefetch depth of the chunk loop in k-steps (2, 3, 4, 6 measured: no difference)
#endif
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 512, D = 256;
constexpr int NW = 4;                              // waves per workgroup; two workgroups per CU (NW = 8, one workgroup per CU with
                                                   // half the DMA issues per wave, measured slower: 19.6 vs 14.3 us chunk loop --
                                                   // eight waves stall together at every chunk barrier, two workgroups of four do not)
constexpr int NT = 64 * NW;                        // threads
constexpr int WG_ROWS = 32 * NW;                   // a wave owns 32 rows
constexpr int KS = D / 16;                         // 16 MFMA k-steps
constexpr int NCHUNK = K / 64;                     // 8 chunks of 64 entries
constexpr int EXP_LIMIT = 40;                      // |log2(max magnitude)| beyond this -> exact fallback

constexpr int CHUNK_B = 64 * D * 2;                // 32 768 B: 64 entries x 256 fp16
constexpr int OFF_EE = 2 * CHUNK_B;                // [K] fp32
constexpr int OFF_CNT = OFF_EE + K * 4;            // [4] ints (per-wave ambiguous counts)
constexpr int Z_STAGE_B = NW * 16 * 1024;          // row staging of the prologue: 16 rows per wave per round
constexpr int LDS_BYTES = (OFF_CNT + 64 > Z_STAGE_B) ? OFF_CNT + 64 : Z_STAGE_B;
// refine tail: regions inside the (then free) codebook stages
constexpr int RF_PAIRS = WG_ROWS * 6;              // pair list capacity
constexpr int RF_ROW = 0;                          // [RF_PAIRS] u16 local row of a pair
constexpr int RF_K = RF_ROW + RF_PAIRS * 2;        // [RF_PAIRS] u16 entry of a pair
constexpr int RF_D = RF_K + RF_PAIRS * 2;          // [RF_PAIRS] f32 canonical distance of a pair
constexpr int RF_OVER = RF_D + RF_PAIRS * 4;       // [WG_ROWS] u16 rows that need more than their candidate list
constexpr int RF_RED = RF_OVER + WG_ROWS * 2;      // [NT] f32 + [NT] int block reduction
constexpr int RF_AP = RF_RED + NT * 8;             // [K] f32 second-level (plain fp32) distances of one row
constexpr int RF_OBASE = RF_AP + K * 4;            // [WG_ROWS] u16 first pair of a second-level row
constexpr int RF_OCNT = RF_OBASE + WG_ROWS * 2;    // [WG_ROWS] u16 its pair count
constexpr int RF_OEPS = RF_OCNT + WG_ROWS * 2;     // [WG_ROWS] f32 its eps2
constexpr int RF_OKEEP = RF_OEPS + WG_ROWS * 4;    // [WG_ROWS][4] u16 its candidates from the lane half whose list is complete
static_assert(RF_OKEEP + WG_ROWS * 8 <= OFF_EE, "refine regions must fit in the codebook stages");

struct PackHeader {
    float emax;        // upper bound of max_k |e_k|_2 (inf if the codebook is not finite)
    int sexp;          // codebook scale sE = 2^sexp
    int valid;         // 0: codebook magnitudes outside the filter's range -> every row takes the exact path
    int K, D;
    float demax;       // upper bound of max_k |e_k - image_k / (-2 sE)|_2: the image's MEASURED fp16 rounding error
};
constexpr size_t PK_OFF_EE = 256;
constexpr size_t PK_OFF_IMG = PK_OFF_EE + (size_t)K * 4;
constexpr size_t PK_BYTES = PK_OFF_IMG + (size_t)K * D * 2;

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
        const float m2 = 2.0f * redm[0];                         // the image holds -2 e
        const int e = (int)((__float_as_uint(m2) >> 23) & 0xff) - 127;
        const bool ok = redm[0] > 0.f && e >= -EXP_LIMIT && e <= EXP_LIMIT && red[0] <= 3.0e38f;
        hdr->sexp = ok ? 13 - e : 0;
        hdr->valid = ok ? 1 : 0;
        hdr->K = K;
        hdr->D = D;
    }
}

// image: [K][D] fp16 of -2 sE e, natural order
__global__ void vq_pack_img_kernel(const float* __restrict__ E, const PackHeader* __restrict__ hdr, _Float16* __restrict__ img) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= K * D) return;
    img[gid] = (_Float16)(-2.0f * pow2f(hdr->sexp) * E[gid]);
}

// measured rounding error of the image, per entry, as a 2-norm in codebook units; its maximum goes into the header
__global__ void vq_pack_err_kernel(const float* __restrict__ E, const _Float16* __restrict__ img, PackHeader* hdr) {
    __shared__ float red[K];
    const int k = threadIdx.x;                                   // blockDim = K
    const float m2s = -2.0f * pow2f(hdr->sexp);
    float acc = 0.f;
    for (int j = 0; j < D; ++j) {
        const float sv = m2s * E[k * D + j];                     // exact (power-of-two scale, range checked by `valid`)
        const float d = sv - (float)img[k * D + j];              // exact: both are fp32 values 11 significant bits apart
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

// ------------------------------------------------------------------------------------------------ filter
// DMA one chunk (64 entries x 512 B) into an LDS stage: 32 pieces of 1 KiB (two rows each), eight per wave; a row has 32
// 16-byte chunks, chunk c of row r lands at chunk c ^ (r & 15)
// per-lane byte offsets of the 8 DMA pieces a wave issues per chunk (constant across chunks: the chunk moves the
// uniform base, so an issue is one instruction, no vector address arithmetic)
constexpr int PPW = 32 / NW;                       // DMA pieces (1 KiB = two image rows) per wave per chunk
__device__ __forceinline__ void chunk_offsets(unsigned (&voff)[PPW], int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int row = 2 * (wave * PPW + i) + (lane >> 5);
        voff[i] = (unsigned)(row * (D * 2) + 16 * ((lane & 31) ^ (row & 15)));
    }
}

__device__ __forceinline__ void issue_piece(const char* __restrict__ img_chunk, unsigned voff, char* stage, int wave, int i) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img_chunk + voff),
                                     (__attribute__((address_space(3))) void*)(stage + (wave * PPW + i) * 1024), 16, 0, 0);
}

__device__ __forceinline__ void issue_chunk(const char* __restrict__ img_chunk, const unsigned (&voff)[PPW], char* stage, int wave) {
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(img_chunk + voff[i]),
                                         (__attribute__((address_space(3))) void*)(stage + (wave * PPW + i) * 1024), 16, 0, 0);
}

__device__ __forceinline__ f16x8 efrag(const char* stage, int row, int chunk) {
    return *reinterpret_cast<const f16x8*>(stage + row * 512 + 16 * (chunk ^ (row & 15)));
}

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p;
}

// Hand-issued LDS reads of the chunk loop.  The compiler's own wait insertion turns every wait after an LDS-DMA issue
// into lgkmcnt(0), which exposes the full LDS latency once per k-step; reads it does not see can be waited for by
// count.  `off` is one of four values (stage x tile), folded after unrolling.
__device__ __forceinline__ void lds_read16(f16x8& d, unsigned addr, int off) {
    switch (off) {
        case 0: asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); break;
        case 16384: asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(d) : "v"(addr)); break;
        case 32768: asm volatile("ds_read_b128 %0, %1 offset:32768" : "=v"(d) : "v"(addr)); break;
        default: asm volatile("ds_read_b128 %0, %1 offset:49152" : "=v"(d) : "v"(addr)); break;
    }
}
__device__ __forceinline__ void lds_read16(f32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr));
}
// wait until at most n younger LDS reads are outstanding; the operands pin the consumers below the wait
#define DVQ_LGKM_CASE(N, ...) case N: asm volatile("s_waitcnt lgkmcnt(" #N ")" : __VA_ARGS__); break;
template <class A, class B>
__device__ __forceinline__ void lds_wait(int n, A& a, B& b) {   // (never pass one object twice: the copy would be made before the wait)
    switch (n) {
        DVQ_LGKM_CASE(0, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(1, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(2, "+v"(a), "+v"(b))
        DVQ_LGKM_CASE(3, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(4, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(5, "+v"(a), "+v"(b))
        DVQ_LGKM_CASE(6, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(7, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(8, "+v"(a), "+v"(b))
        DVQ_LGKM_CASE(9, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(10, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(11, "+v"(a), "+v"(b))
        DVQ_LGKM_CASE(12, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(13, "+v"(a), "+v"(b)) DVQ_LGKM_CASE(14, "+v"(a), "+v"(b))
        default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a), "+v"(b)); break;
    }
}
// wait until at most n younger LDS reads are outstanding; the operands pin the consumers below the wait
template <class A, class B, class C>
__device__ __forceinline__ void lds_wait(int n, A& a, B& b, C& c) {
    switch (n) {
        DVQ_LGKM_CASE(0, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(1, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(2, "+v"(a), "+v"(b), "+v"(c))
        DVQ_LGKM_CASE(3, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(4, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(5, "+v"(a), "+v"(b), "+v"(c))
        DVQ_LGKM_CASE(6, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(7, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(8, "+v"(a), "+v"(b), "+v"(c))
        DVQ_LGKM_CASE(9, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(10, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(11, "+v"(a), "+v"(b), "+v"(c))
        DVQ_LGKM_CASE(12, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(13, "+v"(a), "+v"(b), "+v"(c)) DVQ_LGKM_CASE(14, "+v"(a), "+v"(b), "+v"(c))
        default: asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(a), "+v"(b), "+v"(c)); break;
    }
}

// keep the four smallest of (m1 <= m2 <= m3 <= m4) and s: four median-of-3 (min(a, b) = med3(a, b, -inf); plain fminf
// costs a second instruction for sNaN canonicalisation)
__device__ __forceinline__ void top4(float s, float& m1, float& m2, float& m3, float& m4) {
    m4 = __builtin_amdgcn_fmed3f(m3, m4, s);
    m3 = __builtin_amdgcn_fmed3f(m2, m3, s);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, s);
    m1 = __builtin_amdgcn_fmed3f(m1, s, -3.0e38f);
}

// Number of hand-issued LDS reads younger than the last one k-step s waits for, from the chunk loop's issue order:
//   prologue: fragments of steps 0 .. PF-1 (two reads each), then (c > 0) the |e|^2 quads of score groups 0 .. QF-1;
//   step t:   (c > 0, t even, t/2 + QF < KS/2) quad of group t/2 + QF; (t + PF < KS) the two fragments of step t + PF; wait(t).
// wait(s) needs both fragments of step s and (c > 0) the quad of group s/2.
constexpr int lds_younger(int s, bool quads, int PF, int QF, int KSn) {
    int pos = 0, need = -1, frag_pos = -1, quad_pos = -1;
    for (int t = 0; t < PF; ++t) { pos += 2; if (t == s) frag_pos = pos; }
    if (quads) for (int g = 0; g < QF; ++g) { pos += 1; if (g == s / 2) quad_pos = pos; }
    for (int t = 0; t <= s; ++t) {
        if (quads && (t & 1) == 0 && t / 2 + QF < KSn / 2) { pos += 1; if (t / 2 + QF == s / 2) quad_pos = pos; }
        if (t + PF < KSn) { pos += 2; if (t + PF == s) frag_pos = pos; }
    }
    need = frag_pos > quad_pos ? frag_pos : quad_pos;
    return pos - need;
}

// one score s = acc / (sz sE) + ee, its 8-bit entry id replacing the low mantissa byte, top-4 update.  The id comes
// out of a register holding four ids (one v_perm_b32: gfx950's three-operand encodings take no literal, so an
// and/or with two constants would be two instructions).
__device__ __forceinline__ void absorb1(float acc, float inv, float ee, unsigned ids4, int k,
                                        float& m1, float& m2, float& m3, float& m4) {
    const float sc = fmaf(acc, inv, ee);
    const float p = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(sc), ids4, 0x07060500u + (unsigned)k));
    top4(p, m1, m2, m3, m4);
}

__device__ __forceinline__ int decode_entry(float packed, int half) {
    const unsigned id = __float_as_uint(packed) & 0xffu;
    const int c = id >> 5, jn = (id >> 4) & 1, e = id & 15;
    return 64 * c + 32 * jn + (e & 3) + 8 * (e >> 2) + 4 * half;
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) v += __shfl_xor(v, w);
    return v;
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

__global__ __launch_bounds__(NT, 8 / NW) void vq_filter_kernel(const float* __restrict__ z, const float* __restrict__ E, long M,
                                                           const char* __restrict__ packed, int64_t* __restrict__ idx,
                                                           unsigned long long* __restrict__ slow_rows,
                                                           unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float* ee_s = reinterpret_cast<float*>(lds + OFF_EE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    const float emax = hdr->emax, demax = hdr->demax;
    const int e_sexp = hdr->sexp;
    const bool e_valid = hdr->valid != 0;
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    const _Float16* img = reinterpret_cast<const _Float16*>(packed + PK_OFF_IMG);

    // ---- this lane's row: 128 of its 256 floats (k = 16 s + 8 h + j).  Rows come HBM -> LDS as whole 1 KiB lines by
    // global_load_lds (coalesced; fragment-shaped register loads would touch 32 cache lines per instruction and make the
    // texture path the bottleneck), 64 rows at a time through the (still unused) codebook stages, then to registers.
    long grow = (long)blockIdx.x * WG_ROWS + wave * 32 + r;
    const bool live = grow < M;
    if (!live) grow = M - 1;
    f32x4 zf[KS][2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // rows [64 half, 64 half + 64) of the workgroup tile: 16 DMAs per wave; LDS chunk `lane` <- source chunk lane ^ (row & 15)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lrow = wave * 16 + i;                       // inside this half
            long gr = (long)blockIdx.x * WG_ROWS + (WG_ROWS / 2) * half + lrow;
            if (gr >= M) gr = M - 1;
            const float* src = z + gr * D + 4 * (lane ^ (lrow & 15));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds + lrow * 1024), 16, 0, 0);
        }
        __syncthreads();                                          // landed for every wave
        if (wave / (NW / 2) == half) {
            const int lrow = (wave % (NW / 2)) * 32 + r;
            const float* zrow = reinterpret_cast<const float*>(lds + lrow * 1024);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int c0 = 4 * s + 2 * h;
                zf[s][0] = *reinterpret_cast<const f32x4*>(zrow + 4 * (c0 ^ (lrow & 15)));
                zf[s][1] = *reinterpret_cast<const f32x4*>(zrow + 4 * ((c0 + 1) ^ (lrow & 15)));
            }
        }
        __syncthreads();                                          // consumed: the area may be overwritten
    }
    for (int i = tid; i < K; i += NT) ee_s[i] = ee_g[i];
    if (tid < 2) reinterpret_cast<int*>(lds + OFF_CNT)[tid] = 0;   // refine counters (the row staging is over; first use is barriers away)
    float mx = 0.f, ss = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mx = fmaxf(mx, fabsf(zf[s][q][i]));
                ss = fmaf(zf[s][q][i], zf[s][q][i], ss);
            }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    ss += __shfl_xor(ss, 32);
    const bool bad = !(ss <= 3.0e38f);                          // NaN / Inf anywhere in the row poisons the sum of squares
    const int ez = (int)((__float_as_uint(mx) >> 23) & 0xff) - 127;
    const bool ok = e_valid && !bad && (mx == 0.f || (ez >= -EXP_LIMIT && ez <= EXP_LIMIT));
    const int zs = (mx == 0.f || !ok) ? 0 : 13 - ez;
    const float sc = pow2f(zs);
    const float zn = __builtin_amdgcn_sqrtf(ss) * 1.0001f;
    f16x8 zh[KS];                                                // MFMA B operand: B[k = 8h + j][col = row r]
    float dsq = 0.f;                                             // measured rounding error of the row: sum (sz z_j - h(sz z_j))^2
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 v;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sv = zf[s][q][i] * sc;               // exact
                const _Float16 hv = (_Float16)sv;
                const float d = sv - (float)hv;                  // exact
                dsq = fmaf(d, d, dsq);
                v[4 * q + i] = hv;
            }
        zh[s] = v;
    }
    dsq += __shfl_xor(dsq, 32);
    const float dzn = __builtin_amdgcn_sqrtf(dsq) * pow2f(-zs) * 1.0001f;    // |z - h(sz z)/sz|_2, rounded up

    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    // ---- stream the codebook: chunk c in stage c & 1; scores of chunk c are absorbed while chunk c+1 multiplies
    unsigned voff[PPW];
    chunk_offsets(voff, wave, lane);
    const char* img_b = reinterpret_cast<const char*>(img);
    issue_chunk(img_b, voff, lds, wave);
    float m1 = INFINITY, m2 = INFINITY, m3 = INFINITY, m4 = INFINITY;
    const float inv = ok ? pow2f(-(zs + e_sexp)) : __int_as_float(0x7fc00000);     // NaN marks "exact path"
    unsigned ids4[8];                                             // ids4[j] byte k = id of score 4j + k of the chunk being absorbed
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ids4[j] = 0x03020100u + 0x04040404u * j;
        asm volatile("" : "+v"(ids4[j]));                         // keep them in registers (a folded constant would cost a move per score)
    }
    unsigned fa[KS];                                              // LDS byte address of this lane's fragment of k-step s (stage 0, tile 0)
#pragma unroll
    for (int s = 0; s < KS; ++s) fa[s] = lds_addr(lds) + r * 512 + 16 * ((2 * s + h) ^ (r & 15));
    const unsigned ee_a = lds_addr(lds) + OFF_EE + 16 * h;
    f32x16 accA[2], accB[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's DMA pieces of chunk c (the compiler does not see the asm readers)
        __syncthreads();                                          // chunk c landed everywhere; the other stage is free
        f32x16 (&cur)[2] = (c & 1) ? accB : accA;
        f32x16 (&prev)[2] = (c & 1) ? accA : accB;
        // Issue order per k-step s: [|e|^2 quad of score group s/2+1 (even s)], fragments of step s+PF, wait for step s's
        // fragments (and the quad its scores need, which is older), two MFMAs, two scores of the previous chunk.
        f16x8 ef[KS][2];
        f32x4 ev[KS / 2];
        constexpr int PF = DVQ_PF, QF = 2;                     // fragments PF k-steps ahead, quads QF score groups ahead
        const int soff = (c & 1) * CHUNK_B;
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            lds_read16(ef[s][0], fa[s], soff);
            lds_read16(ef[s][1], fa[s], soff + 16384);
        }
        if (c > 0) {
#pragma unroll
            for (int g = 0; g < QF; ++g) lds_read16(ev[g], ee_a + 256 * (c - 1) + 128 * (g >> 2) + 32 * (g & 3));
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // the next chunk's DMA pieces go out one per k-step, in the vector-heavy gaps: issued back to back behind the
            // barrier (with the burst of first fragment reads) each cost 100-185 issue cycles, here a fraction of that
            if (c + 1 < NCHUNK && s < PPW) issue_piece(img_b + (c + 1) * CHUNK_B, voff[s], lds + ((c + 1) & 1) * CHUNK_B, wave, s);
            if (c > 0 && (s & 1) == 0 && s / 2 + QF < KS / 2) {
                const int g = s / 2 + QF;
                lds_read16(ev[g], ee_a + 256 * (c - 1) + 128 * (g >> 2) + 32 * (g & 3));
            }
            if (s + PF < KS) {
                lds_read16(ef[s + PF][0], fa[s + PF], soff);
                lds_read16(ef[s + PF][1], fa[s + PF], soff + 16384);
            }
            const int younger = lds_younger(s, c > 0, PF, QF, KS);
            if (c > 0) lds_wait(younger, ef[s][0], ef[s][1], ev[s / 2]);
            else lds_wait(younger, ef[s][0], ef[s][1]);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ef[s][0], zh[s], s == 0 ? zero16 : cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ef[s][1], zh[s], s == 0 ? zero16 : cur[1], 0, 0, 0);
            if (c > 0) {              // two scores of the previous chunk per k-step: vector work in the MFMA shadow
                const int q = 2 * s, jn = q >> 4, e = q & 15;     // acc register e of tile jn: entry 64 (c-1) + 32 jn + 8 (e>>2) + 4 h + (e&3)
                absorb1(prev[jn][e], inv, ev[q >> 2][q & 3], ids4[q >> 2], q & 3, m1, m2, m3, m4);
                absorb1(prev[jn][e + 1], inv, ev[q >> 2][(q + 1) & 3], ids4[q >> 2], (q + 1) & 3, m1, m2, m3, m4);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c > 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ids4[j] += 0x20202020u;
                asm volatile("" : "+v"(ids4[j]));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        const int jn = q >> 4, e = q & 15;
        const float ee = ee_s[64 * (NCHUNK - 1) + 32 * jn + 8 * (e >> 2) + 4 * h + (e & 3)];
        absorb1(accB[jn][e], inv, ee, ids4[q >> 2], q & 3, m1, m2, m3, m4);
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    // ---- per row: merge the two lane halves' top-4 lists, threshold, decide or refine
    const float p1 = __shfl_xor(m1, 32), p2 = __shfl_xor(m2, 32), p3 = __shfl_xor(m3, 32), p4 = __shfl_xor(m4, 32);
    const float best = fminf(m1, p1);
    const float eps = 4.004f * (dzn * emax + zn * demax + dzn * demax) + 0.00018311f * (zn + emax) * (zn + emax);
    const float thr = best + eps + (inv - inv);                  // NaN scale (exact path) poisons the threshold
    const int own = (m1 <= thr) + (m2 <= thr) + (m3 <= thr);
    const int oth = (p1 <= thr) + (p2 <= thr) + (p3 <= thr);
    const int c = own + oth;
    const bool complete = !(m4 <= thr) && !(p4 <= thr);          // no lane may be hiding a fourth score within eps
    const bool unique = c == 1;
    const bool owner = live && h == 0;
    const bool amb = owner && !unique;
    const bool usable = complete && c >= 2;                       // c == 0: NaN/Inf or out-of-range magnitudes
    // The refine runs inside the workgroup (no second launch, no hand-off through HBM): ambiguous rows claim slots
    // of a (row, entry) pair list in LDS (the codebook stages are free now), the pairs are evaluated four lanes per
    // canonical chain, and each row's owner lane picks its winner.
    int* s_tot = reinterpret_cast<int*>(lds + OFF_CNT);          // [0] pairs, [1] rows that need all K entries
    uint16_t* s_row = reinterpret_cast<uint16_t*>(lds + RF_ROW);
    uint16_t* s_k = reinterpret_cast<uint16_t*>(lds + RF_K);
    float* s_d = reinterpret_cast<float*>(lds + RF_D);
    uint16_t* s_over = reinterpret_cast<uint16_t*>(lds + RF_OVER);
    float* s_v = reinterpret_cast<float*>(lds + RF_RED);
    int* s_i = reinterpret_cast<int*>(lds + RF_RED + NT * 4);
    float* s_ap = reinterpret_cast<float*>(lds + RF_AP);
    uint16_t* s_obase = reinterpret_cast<uint16_t*>(lds + RF_OBASE);
    uint16_t* s_ocnt = reinterpret_cast<uint16_t*>(lds + RF_OCNT);
    float* s_oeps = reinterpret_cast<float*>(lds + RF_OEPS);
    uint16_t* s_okeep = reinterpret_cast<uint16_t*>(lds + RF_OKEEP);
    __syncthreads();                                              // every wave is done with the stages
    int base = 0;
    if (owner && unique) idx[grow] = (m1 <= thr) ? decode_entry(m1, 0) : decode_entry(p1, 1);
    if (amb) {
        const int rl = wave * 32 + r;
        if (usable) {
            base = atomicAdd(&s_tot[0], c);
            int n = base;
            auto push = [&](bool take, int v) {
                if (take) { s_row[n] = (uint16_t)rl; s_k[n] = (uint16_t)v; ++n; }
            };
            push(m1 <= thr, decode_entry(m1, 0));
            push(m2 <= thr, decode_entry(m2, 0));
            push(m3 <= thr, decode_entry(m3, 0));
            push(p1 <= thr, decode_entry(p1, 1));
            push(p2 <= thr, decode_entry(p2, 1));
            push(p3 <= thr, decode_entry(p3, 1));
        } else {
            // !complete: finite row, more than four scores of one lane within eps -> second-level filter below;
            // c == 0: NaN/Inf or out-of-range magnitudes -> all K entries canonically (bit 15)
            // bits 13 / 14: which lane half (entry index bit 2) hides more scores than its list holds and must be rescanned;
            // the other half's (complete) list is kept
            const int o = atomicAdd(&s_tot[1], 1);
            const bool inc0 = m4 <= thr, inc1 = p4 <= thr;
            s_over[o] = (uint16_t)(rl | (complete ? 0x8000 : 0) | (inc0 ? 0x2000 : 0) | (inc1 ? 0x4000 : 0));
            s_oeps[o] = 0.00018311f * (zn + emax) * (zn + emax);
            int nk = 0;
            auto keep = [&](bool take, int v) {
                if (take) { s_okeep[4 * o + 1 + nk] = (uint16_t)v; ++nk; }
            };
            if (!inc0) { keep(m1 <= thr, decode_entry(m1, 0)); keep(m2 <= thr, decode_entry(m2, 0)); keep(m3 <= thr, decode_entry(m3, 0)); }
            if (!inc1) { keep(p1 <= thr, decode_entry(p1, 1)); keep(p2 <= thr, decode_entry(p2, 1)); keep(p3 <= thr, decode_entry(p3, 1)); }
            s_okeep[4 * o] = (uint16_t)nk;
        }
    }
    __syncthreads();
    const int n_over = s_tot[1];
    if (slow_rows && n_over > 0 && tid == 0) atomicAdd(slow_rows, (unsigned long long)n_over);
    // Second-level filter for the (rare) finite rows with a possibly incomplete list: every entry's distance by plain
    // fp32 dot products, a wave per entry so the row loads are coalesced.  Any summation order keeps
    // |approx_k - true_k| <= gamma_260 (|z| + |e_k|)^2, and so does the canonical chain, hence the canonical winner
    // is within 4 gamma_260 (|z| + Emax)^2 < eps2 = (2^-13 + 2^-14) (|z| + Emax)^2 of the approximate minimum.
    for (int o = 0; o < n_over; ++o) {
        const int ovf = s_over[o];
        if (ovf & 0x8000) continue;
        const int ov = ovf & 0x1fff;
        const bool both = (ovf & 0x6000) == 0x6000;
        const int hsel = (ovf & 0x4000) ? 1 : 0;                  // the half to scan when only one is incomplete
        const int n_scan = both ? K : K / 2;
        auto entry_of = [&](int e) { return both ? e : 8 * (e >> 2) + 4 * hsel + (e & 3); };
        const long gr = (long)blockIdx.x * WG_ROWS + ov;
        const f32x4 z4 = *reinterpret_cast<const f32x4*>(z + gr * D + 4 * lane);
        const float zz_a = wave_sum(z4[0] * z4[0] + z4[1] * z4[1] + z4[2] * z4[2] + z4[3] * z4[3]);
        f32x4 nx[16];                                             // next batch: 16 coalesced row loads always in flight behind the math
#pragma unroll
        for (int u = 0; u < 16; ++u) nx[u] = *reinterpret_cast<const f32x4*>(E + (long)entry_of(NW * u + wave) * D + 4 * lane);
        for (int i0 = 0; i0 < n_scan / NW; i0 += 16) {
            f32x4 e4[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) e4[u] = nx[u];
            if (i0 + 16 < n_scan / NW) {
#pragma unroll
                for (int u = 0; u < 16; ++u) nx[u] = *reinterpret_cast<const f32x4*>(E + (long)entry_of(NW * (i0 + 16 + u) + wave) * D + 4 * lane);
            }
            float pd[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) pd[u] = z4[0] * e4[u][0] + z4[1] * e4[u][1] + z4[2] * e4[u][2] + z4[3] * e4[u][3];
            // 16 wave sums by a halving butterfly (17 shuffles instead of 96): after the masks 32, 16, 8, 4 the lane holds
            // the partial of entry u = 8 b5 + 4 b4 + 2 b3 + b2 (b = lane bits) over its 16-lane class; masks 2, 1 finish it
#pragma unroll
            for (int lvl = 0; lvl < 4; ++lvl) {
                const int w = 32 >> lvl, n = 8 >> lvl;
                const bool up = (lane & w) != 0;
#pragma unroll
                for (int u = 0; u < n; ++u) {
                    const float send = up ? pd[u] : pd[u + n];
                    const float keep = up ? pd[u + n] : pd[u];
                    pd[u] = keep + __shfl_xor(send, w);
                }
            }
            float tot = pd[0] + __shfl_xor(pd[0], 2);
            tot += __shfl_xor(tot, 1);
            if ((lane & 3) == 0) {
                const int u = ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
                const int e = NW * (i0 + u) + wave;                    // scan index; s_ap is indexed by it
                s_ap[e] = (zz_a + ee_s[entry_of(e)]) - 2.0f * tot;
            }
        }
        __syncthreads();
        float mn = INFINITY;
#pragma unroll
        for (int u = 0; u < K / NT; ++u)
            if (tid + NT * u < n_scan) mn = fminf(mn, s_ap[tid + NT * u]);
#pragma unroll
        for (int w = 32; w > 0; w >>= 1) mn = fminf(mn, __shfl_xor(mn, w));
        if (lane == 0) s_v[wave] = mn;
        const int base_o = s_tot[0];
        __syncthreads();
        float mn_all = s_v[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mn_all = fminf(mn_all, s_v[w]);
        const float thr2 = mn_all + s_oeps[o];
#pragma unroll
        for (int u = 0; u < K / NT; ++u) {
            const int e = tid + NT * u;
            if (e < n_scan && s_ap[e] <= thr2) {
                const int pos = atomicAdd(&s_tot[0], 1);
                if (pos < RF_PAIRS) { s_row[pos] = (uint16_t)ov; s_k[pos] = (uint16_t)entry_of(e); }
            }
        }
        if (!both && tid < s_okeep[4 * o]) {                      // the complete half's own candidates
            const int pos = atomicAdd(&s_tot[0], 1);
            if (pos < RF_PAIRS) { s_row[pos] = (uint16_t)ov; s_k[pos] = s_okeep[4 * o + 1 + tid]; }
        }
        __syncthreads();
        if (tid == 0) {
            const int c_o = s_tot[0] - base_o;
            if (c_o < 1 || c_o > 64 || base_o + c_o > RF_PAIRS) {   // (c_o < 1 cannot happen for finite data) -> all K entries
                s_over[o] = (uint16_t)(ovf | 0x8000);
                s_tot[0] = base_o;
            } else {
                s_obase[o] = (uint16_t)base_o;
                s_ocnt[o] = (uint16_t)c_o;
            }
        }
        __syncthreads();
    }
    const int total = s_tot[0];
    for (int s0 = 0; s0 < total; s0 += NT / 4) {
        if (s0 + wave * 16 < total) {                             // wave-uniform: this wave has at least one pair
            const int slot = s0 + (tid >> 2), q = tid & 3;
            const bool act = slot < total;
            const int rl = act ? s_row[slot] : 0, k = act ? s_k[slot] : 0;
            float zz, dot;
            chain_pair_x4(z + ((long)blockIdx.x * WG_ROWS + rl) * D, E + (long)k * D, q, act, zz, dot);
            if (act && q == 0) {
                const float t = zz + ee_s[k];
                s_d[slot] = t - 2.0f * dot;
            }
        }
    }
    __syncthreads();
    if (amb && usable) {
        float d = s_d[base];
        int k = s_k[base];
        for (int i = 1; i < c; ++i) {
            const float od = s_d[base + i];
            const int ok2 = s_k[base + i];
            if (dvq_argmin_better(od, ok2, d, k)) { d = od; k = ok2; }
        }
        idx[grow] = k;
    }
    if (tid < n_over && !(s_over[tid] & 0x8000)) {               // rows that went through the second-level filter
        const int b0 = s_obase[tid], cn = s_ocnt[tid];
        float d = s_d[b0];
        int k = s_k[b0];
        for (int i = 1; i < cn; ++i) {
            const float od = s_d[b0 + i];
            const int ok2 = s_k[b0 + i];
            if (dvq_argmin_better(od, ok2, d, k)) { d = od; k = ok2; }
        }
        idx[(long)blockIdx.x * WG_ROWS + (s_over[tid] & 0x1fff)] = k;
    }
    // what is left (NaN/Inf, out-of-range magnitudes, > 64 second-level candidates): all K entries canonically, the whole
    // workgroup per row, one single-lane chain per entry
    for (int o = 0; o < n_over; ++o) {
        if (!(s_over[o] & 0x8000)) continue;
        const long gr = (long)blockIdx.x * WG_ROWS + (s_over[o] & 0x1fff);
        const float* zr = z + gr * D;
        float bv = INFINITY;
        int bi = 0x7fffffff;
        for (int kk = tid; kk < K; kk += NT) {
            float zz2, dot2;
            chain_pair(zr, E + (long)kk * D, zz2, dot2);
            const float t = zz2 + ee_s[kk];
            const float dd = t - 2.0f * dot2;
            if (dvq_argmin_better(dd, kk, bv, bi)) { bv = dd; bi = kk; }
        }
        s_v[tid] = bv;
        s_i[tid] = bi;
        __syncthreads();
        for (int w = NT / 2; w > 0; w >>= 1) {
            if (tid < w && dvq_argmin_better(s_v[tid + w], s_i[tid + w], s_v[tid], s_i[tid])) {
                s_v[tid] = s_v[tid + w];
                s_i[tid] = s_i[tid + w];
            }
            __syncthreads();
        }
        if (tid == 0) idx[gr] = s_i[0];
        __syncthreads();
    }
    if (dbg && tid == 0) {
        const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
        dbg[blockIdx.x * 8 + 0] = t0; dbg[blockIdx.x * 8 + 1] = t1; dbg[blockIdx.x * 8 + 2] = t2; dbg[blockIdx.x * 8 + 3] = t3;
        dbg[blockIdx.x * 8 + 4] = (unsigned long long)total; dbg[blockIdx.x * 8 + 5] = (unsigned long long)n_over;
        dbg[blockIdx.x * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_ID
        dbg[blockIdx.x * 8 + 7] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // XCC_ID
    }
}

struct FastScratch {
    char* dbg;
    long n_wg;
    size_t bytes;
};

FastScratch plan(int64_t M, void* ws) {
    FastScratch s;
    s.n_wg = (M + WG_ROWS - 1) / WG_ROWS;
    s.dbg = (char*)ws;                                           // per-workgroup phase stamps (DVQ_VQ_DBG only)
    s.bytes = dvq_round_up((size_t)s.n_wg * 64, 256);
    return s;
}

}  // namespace

extern "C" int dvq_vq_fast_supported(int Kq, int Dq) { return Kq == K && Dq == D; }

extern "C" size_t dvq_vq_pack_bytes(int Kq, int Dq) { return dvq_vq_fast_supported(Kq, Dq) ? PK_BYTES : 0; }

extern "C" int dvq_vq_pack(const float* E, int Kq, int Dq, void* packed, size_t packed_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_pack: the fast path supports K=%d, D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(E && packed && dvq_aligned16(E) && dvq_aligned16(packed), "vq_pack: null/unaligned pointer");
    DVQ_REQUIRE(packed_bytes >= PK_BYTES, "vq_pack: buffer %zu < %zu bytes", packed_bytes, PK_BYTES);
    hipStream_t st = (hipStream_t)stream;
    char* pk = (char*)packed;
    DVQ_LAUNCH(vq_pack_norm_kernel, dim3(1), dim3(K), 0, st, E, (float*)(pk + PK_OFF_EE), (PackHeader*)pk);
    DVQ_CHECK_LAUNCH("vq_pack_norm");
    DVQ_LAUNCH(vq_pack_img_kernel, dim3((K * D + 255) / 256), dim3(256), 0, st, E, (const PackHeader*)pk,
                       (_Float16*)(pk + PK_OFF_IMG));
    DVQ_CHECK_LAUNCH("vq_pack_img");
    DVQ_LAUNCH(vq_pack_err_kernel, dim3(1), dim3(K), 0, st, E, (const _Float16*)(pk + PK_OFF_IMG), (PackHeader*)pk);
    DVQ_CHECK_LAUNCH("vq_pack_err");
    return DVQ_OK;
}

extern "C" size_t dvq_vq_fast_workspace_bytes(int64_t M, int Kq, int Dq) {
    if (M <= 0 || !dvq_vq_fast_supported(Kq, Dq)) return 256;
    return plan(M, nullptr).bytes;
}

extern "C" int dvq_vq_argmin_fast(const float* z, const float* E, const void* packed, int64_t M, int Kq, int Dq,
                                  int64_t* idx, unsigned long long* slow_rows, void* workspace, size_t workspace_bytes,
                                  dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_argmin_fast: supports K=%d, D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(M >= 0 && M < (1L << 31), "vq_argmin_fast: bad M");
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
    static DvqOncePerDevice attr_once;
    if (attr_once.first()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) {
            dvq_set_error("vq_argmin_fast: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const char* pk = (const char*)packed;
    DVQ_PROF("vq_argmin_fast", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 2 + (double)M * 8, st);
    DVQ_LAUNCH(vq_filter_kernel, dim3((unsigned)s.n_wg), dim3(NT), LDS_BYTES, st, z, E, (long)M, pk, idx,
                       slow_rows, getenv("DVQ_VQ_DBG") ? (unsigned long long*)s.dbg : nullptr);
    DVQ_CHECK_LAUNCH("vq_filter");
    return DVQ_OK;
}
