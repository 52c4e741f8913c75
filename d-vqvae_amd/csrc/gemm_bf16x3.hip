// fp32-accurate GEMM on the bf16 matrix cores ("bf16x3"): every fp32 operand is split EXACTLY into three bf16
// pieces  a = a1 + a2 + a3  (8 + 8 + 8 significant bits, truncation split: a1 = top 16 bits of a, a2 = top 16 bits of
// a - a1, a3 = a - a1 - a2, all exact), and the product is evaluated as the six partial products of weight >= 2^-24
//      a*b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)          (dropped terms <= 2^-23 |a b|)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 products are exact in fp32, so the result has the
// accuracy of an fp32 GEMM (1e-7 relative) while the matrix pipe runs 6 x 32 = 192 cycles per 32x32x16 block instead
// of 8 x 64 = 512 for v_mfma_f32_32x32x2_f32: 2.67x fewer MFMA cycles.  No scaling is needed (bf16 has fp32's
// exponent range); non-finite inputs give NaN.
//
// Same tiling, block->tile map, multi-source K loop and epilogues as gemm_f32.hip (128x128 tile, 4 waves 2x2, BK = 32,
// LDS rows of 32 bf16 padded to 80 B: ds_read_b128 conflict-free; ONE 60 KiB stage, 2 workgroups per CU whose
// matrix / staging phases interleave; global -> register prefetch two K-tiles ahead).
// Activations are split on the fly while they are staged (global fp32 -> registers -> 3 LDS planes); weights come
// pre-split from the packer when GemmSrc::Wp is set (3 bf16 planes), else they are split on the fly too.
// The accumulation order is still independent of the M tiling (batched == loop, bitwise).
// Three kernels share the arithmetic: gemm_bf16x3_kernel (register-staged, any alignment), gemm_bf16x3_dma_kernel (LDS-DMA of
// the fp32 activation tile and the pre-split weight planes, 128 x 128) and gemm_bf16x3_wide_kernel (128 x 256, eight waves,
// activations split once on their way into the LDS; the default where N % 256 == 0 and N >= 512).
#include "dvq_internal.h"
#include "gemm_common.h"
#include <vector>

// Diagnostics (per-block phase stamps, loop ablations: tools/gemm_phase_stamps.py) are compiled in only with
// -DDVQ_GEMM_DIAG (make EXTRA=-DDVQ_GEMM_DIAG): in the shipped build the predicates below are constant false.
#ifdef DVQ_GEMM_DIAG
#define DVQ_ABL_IS(p, v) ((p).dbg_abl == (v))
#define DVQ_CLK(p) ((p).dbg_clk)
#else
#define DVQ_ABL_IS(p, v) false
#define DVQ_CLK(p) ((unsigned long long*)nullptr)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = GEMM_BM, BN = GEMM_BN, BK = 32;
constexpr int ROWB = 80;                                   // bytes per LDS row (64 B of data + 16 B pad: 5r mod 16 distinct)
constexpr int PLANE_B = 128 * ROWB;                        // one plane of one operand: 128 rows
constexpr int STAGE_B = 6 * PLANE_B;                       // A planes 0..2, W planes 0..2
constexpr size_t SMEM_BYTES = STAGE_B;                     // 61 440 B, single stage: 2 workgroups per CU

// exact 3-way split of two floats; returns the three packed bf16 pairs (element 0 in the low half)
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned a0 = __float_as_uint(x0) & 0xffff0000u, a1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = x0 - __uint_as_float(a0), r1 = x1 - __uint_as_float(a1);
    const unsigned b0 = __float_as_uint(r0) & 0xffff0000u, b1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(b0), s1 = r1 - __uint_as_float(b1);
    p1 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    p2 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

struct Staged {
    f32x4 a[4];          // A: 4 x 4 consecutive k (rows r0 + 32 i, column chunk c4)
    f32x4 w[4];          // W (fp32 path)
    uint4 wp[6];         // W (pre-split path): 6 x 16 B
};

// Per-thread cursor over the flattened (source, K-tile) sequence.  Source parameters are read once per source and the
// per-thread global pointers advance by BK afterwards (no kernarg reads / address rebuilds in the steady state).
template <bool WPLANES>
struct Cursor {
    int s, k_left;
    const float* a_ptr[4];
    const float* w_ptr[4];
    const uint16_t* wp_ptr[6];
    unsigned ok;          // bit i: A row i valid; bit 4+i: W row i valid; bit 8+i: W plane chunk i valid

    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m0, int n0, int tid) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        const int c4 = tid & 7, r0 = tid >> 3;
        ok = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + r0 + 32 * i;
            const bool v = m < p.M;
            ok |= v ? (1u << i) : 0u;
            a_ptr[i] = src.A + (v ? m : 0) * src.lda + c4 * 4;
        }
        if constexpr (WPLANES) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {             // 128 rows x 4 chunks x 3 planes = 1536 chunks of 16 B
                const int id = tid + 256 * i;
                const int plane = id >> 9, row = (id >> 2) & 127, ch = id & 3;
                const bool v = n0 + row < p.N;
                ok |= v ? (1u << (8 + i)) : 0u;
                wp_ptr[i] = src.Wp + plane * src.wp_plane + (long)(v ? n0 + row : 0) * src.ldw + ch * 8;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + r0 + 32 * i;
                const bool v = n < p.N;
                ok |= v ? (1u << (4 + i)) : 0u;
                w_ptr[i] = src.W + (long)(v ? n : 0) * src.ldw + c4 * 4;
            }
        }
    }
    __device__ __forceinline__ bool valid() const { return k_left > 0; }
    // issue the loads of the current tile into t, then step to the next tile
    __device__ __forceinline__ void fetch(const GemmParams& p, long m0, int n0, int tid, Staged& t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t.a[i] = (ok >> i & 1u) ? *reinterpret_cast<const f32x4*>(a_ptr[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            a_ptr[i] += BK;
        }
        if constexpr (WPLANES) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                t.wp[i] = (ok >> (8 + i) & 1u) ? *reinterpret_cast<const uint4*>(wp_ptr[i]) : uint4{0u, 0u, 0u, 0u};
                wp_ptr[i] += BK;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                t.w[i] = (ok >> (4 + i) & 1u) ? *reinterpret_cast<const f32x4*>(w_ptr[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
                w_ptr[i] += BK;
            }
        }
        k_left -= BK;
        if (k_left <= 0) open(p, s + 1, m0, n0, tid);
    }
};

__device__ __forceinline__ void split_quad(const f32x4& v, uint2 (&out)[3]) {
    unsigned lo1, lo2, lo3, hi1, hi2, hi3;
    split3_pair(v[0], v[1], lo1, lo2, lo3);
    split3_pair(v[2], v[3], hi1, hi2, hi3);
    out[0] = uint2{lo1, hi1};
    out[1] = uint2{lo2, hi2};
    out[2] = uint2{lo3, hi3};
}

// split the staged registers exactly into bf16 planes and write them to the LDS stage
template <bool WPLANES>
__device__ __forceinline__ void store_stage(char* stage, int tid, const Staged& t) {
    const int c4 = tid & 7, r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint2 pk[3];
        split_quad(t.a[i], pk);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint2*>(stage + pl * PLANE_B + (r0 + 32 * i) * ROWB + c4 * 8) = pk[pl];
    }
    if constexpr (WPLANES) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = tid + 256 * i;
            const int plane = id >> 9, row = (id >> 2) & 127, ch = id & 3;
            *reinterpret_cast<uint4*>(stage + (3 + plane) * PLANE_B + row * ROWB + ch * 16) = t.wp[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 pk[3];
            split_quad(t.w[i], pk);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint2*>(stage + (3 + pl) * PLANE_B + (r0 + 32 * i) * ROWB + c4 * 8) = pk[pl];
        }
    }
}

// one K-tile of 32: two k16 steps of 12 fragment reads + 24 MFMAs
__device__ __forceinline__ void compute_stage(const char* stage, int wm, int wn, int r, int h, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const char* As = stage + (wm * 64 + r) * ROWB + (2 * ks + h) * 16;
        const char* Ws = stage + 3 * PLANE_B + (wn * 64 + r) * ROWB + (2 * ks + h) * 16;
        bf16x8 a[2][3], w[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                a[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * PLANE_B + i * 32 * ROWB);
                w[i][pl] = *reinterpret_cast<const bf16x8*>(Ws + pl * PLANE_B + i * 32 * ROWB);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                f32x16 c = acc[i][jn];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], w[jn][0], c, 0, 0, 0);     // a3 b1
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], w[jn][2], c, 0, 0, 0);     // a1 b3
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], w[jn][1], c, 0, 0, 0);     // a2 b2
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], w[jn][0], c, 0, 0, 0);     // a2 b1
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], w[jn][1], c, 0, 0, 0);     // a1 b2
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], w[jn][0], c, 0, 0, 0);     // a1 b1
                acc[i][jn] = c;
            }
    }
}

template <int EPI, bool WPLANES>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + BN - 1) / BN;
    const long tiles_m = (p.M + BM - 1) / BM;
    const long b = blockIdx.x;
    const long j = b >> 3;
    const long mt = (j / tiles_n) * 8 + (b & 7);
    const int nt = (int)(j % tiles_n);
    if (mt >= tiles_m) return;
    const long m0 = mt * BM;
    const int n0 = nt * BN;

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    // Single LDS stage (two workgroups per CU interleave their phases); global -> register prefetch runs two K-tiles
    // ahead: register sets R0 / R1 alternate, the tile computed in iteration t was loaded during iteration t-2.
    Cursor<WPLANES> cur;
    cur.open(p, 0, m0, n0, tid);
    Staged R0, R1;
    cur.fetch(p, m0, n0, tid, R0);                       // tile 0
    store_stage<WPLANES>(smem_c, tid, R0);
    bool have_next = cur.valid();                        // tile t+1 exists (held in R0 for even t, R1 for odd t)
    if (have_next) cur.fetch(p, m0, n0, tid, R0);        // tile 1
    __syncthreads();
    while (true) {
        // ---- even step: R0 holds tile t+1, R1 receives tile t+2
        const bool have_next2 = cur.valid();
        if (have_next2) cur.fetch(p, m0, n0, tid, R1);
        compute_stage(smem_c, wm, wn, r, h, acc);
        if (!have_next) break;
        __syncthreads();                                 // everybody has read the stage
        store_stage<WPLANES>(smem_c, tid, R0);
        __syncthreads();
        // ---- odd step: R1 holds tile t+1, R0 receives tile t+2
        have_next = cur.valid();
        if (have_next) cur.fetch(p, m0, n0, tid, R0);
        compute_stage(smem_c, wm, wn, r, h, acc);
        if (!have_next2) break;
        __syncthreads();
        store_stage<WPLANES>(smem_c, tid, R1);
        __syncthreads();
    }
    __syncthreads();
    gemm_epilogue<EPI>(p, acc, m0, n0, mt, nt, tid, reinterpret_cast<float*>(smem_c));
}

// ================================================================================================================
// LDS-DMA variant (pre-split weights): nothing goes through registers on the way in.  Per K-tile of 32 every wave issues
// 4 + 6 global_load_lds (1 KiB each): the fp32 activation tile [128][32] and the three bf16 weight planes [128][32],
// both as lane-linear images whose 16-byte chunks are XOR-swizzled on the SOURCE address (fp32 rows: chunk ^ ((row>>1)&7),
// bf16 rows: chunk ^ ((row>>2)&3)) so that the fragment ds_read_b128 are bank-conflict free.  The activation fragment is
// split exactly into its three bf16 pieces in registers at read time -- vector work that hides under the wave's own MFMAs --
// so there are no LDS stores and one barrier per K-tile (two stages, the next tile's DMA flies during the MFMAs).
constexpr int D_A_BYTES = 128 * 128;                       // fp32 [128 rows][32 k]
constexpr int D_WPL_BYTES = 128 * 64;                      // one bf16 plane [128 rows][32 k]
constexpr int D_STAGE = D_A_BYTES + 3 * D_WPL_BYTES;       // 40 960 B
constexpr size_t D_SMEM = 2 * D_STAGE;                     // 81 920 B: two workgroups per CU

struct DmaCursor {
    int s, k_left;
    const float* a_ptr[4];
    const uint16_t* w_ptr[3][2];

    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m0, int n0, int wave, int lane) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                      // 8 rows x 128 B per DMA
            const int row = 32 * wave + 8 * g + (lane >> 3);
            long m = m0 + row;
            if (m >= p.M) m = p.M - 1;                     // clamped rows only feed outputs the epilogue masks
            a_ptr[g] = src.A + m * src.lda + 4 * ((lane & 7) ^ ((row >> 1) & 7));
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {                      // 16 rows x 64 B per DMA
            const int row = 32 * wave + 16 * g + (lane >> 2);
            int n = n0 + row;
            if (n >= p.N) n = p.N - 1;
            const uint16_t* base = src.Wp + (long)n * src.ldw + 8 * ((lane & 3) ^ ((row >> 2) & 3));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) w_ptr[pl][g] = base + pl * src.wp_plane;
        }
    }
    __device__ __forceinline__ bool valid() const { return k_left > 0; }
    __device__ __forceinline__ void issue(const GemmParams& p, long m0, int n0, int wave, int lane, char* stage) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_ptr[g],
                                             (__attribute__((address_space(3))) void*)(stage + (32 * wave + 8 * g) * 128), 16, 0, 0);
            a_ptr[g] += BK;
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (DVQ_ABL_IS(p, 6)) { w_ptr[pl][g] += BK; continue; }          // diag: no weight pieces
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[pl][g],
                                                 (__attribute__((address_space(3))) void*)(stage + D_A_BYTES + pl * D_WPL_BYTES +
                                                                                          (32 * wave + 16 * g) * 64), 16, 0, 0);
                w_ptr[pl][g] += BK;
            }
        k_left -= BK;
        if (k_left <= 0) open(p, s + 1, m0, n0, wave, lane);
    }
};

__device__ __forceinline__ void split_frag(const f32x4& lo, const f32x4& hi, bf16x8 (&out)[3]) {
    unsigned p[3][4];
    split3_pair(lo[0], lo[1], p[0][0], p[1][0], p[2][0]);
    split3_pair(lo[2], lo[3], p[0][1], p[1][1], p[2][1]);
    split3_pair(hi[0], hi[1], p[0][2], p[1][2], p[2][2]);
    split3_pair(hi[2], hi[3], p[0][3], p[1][3], p[2][3]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl] = __builtin_bit_cast(bf16x8, uint4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_dma_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + BN - 1) / BN;
    const long tiles_m = (p.M + BM - 1) / BM;
    const long b = blockIdx.x;
    const long j = b >> 3;
    const long mt = (j / tiles_n) * 8 + (b & 7);
    const int nt = (int)(j % tiles_n);
    if (mt >= tiles_m) return;
    const long m0 = mt * BM;
    const int n0 = nt * BN;

    constexpr bool SWAP = EPI != EPI_COLMAX;   // store epilogues use the transposed accumulator layout (gemm_common.h)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    if (DVQ_CLK(p) && blockIdx.x == 0 && tid == 0) {
        p.dbg_clk[0] = __builtin_amdgcn_s_memtime();
        p.dbg_clk[1] = __builtin_amdgcn_s_memrealtime();
    }
    if (DVQ_CLK(p) && tid == 0) p.dbg_clk[8 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    DmaCursor cur;
    cur.open(p, 0, m0, n0, wave, lane);
    cur.issue(p, m0, n0, wave, lane, smem_c);
    int stage = 0;
    // fragment addressing (constant over the loop)
    int a_off[2], a_sw[2], w_off[2], w_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int arow = wm * 64 + i * 32 + r, wrow = wn * 64 + i * 32 + r;
        a_off[i] = arow * 128;
        a_sw[i] = (arow >> 1) & 7;
        w_off[i] = D_A_BYTES + wrow * 64;
        w_sw[i] = (wrow >> 2) & 3;
    }
    while (true) {
        dvq_dma_barrier();                                 // the DMA of this tile has landed for every wave (vmcnt(0) + barrier);
                                                           // everybody is done reading the other stage
        const bool more = cur.valid();
        if (more) {
            if (DVQ_ABL_IS(p, 5)) { cur.k_left -= BK; if (cur.k_left <= 0) cur.open(p, cur.s + 1, m0, n0, wave, lane); }   // diag: no DMA
            else cur.issue(p, m0, n0, wave, lane, smem_c + (stage ^ 1) * D_STAGE);
        }
        const char* st = smem_c + stage * D_STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 w[2][3];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    w[jn][pl] = *reinterpret_cast<const bf16x8*>(st + w_off[jn] + (DVQ_ABL_IS(p, 3) ? 0 : pl * D_WPL_BYTES + 16 * ((2 * ks + h) ^ w_sw[jn])));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c0 = 4 * ks + 2 * h;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(st + a_off[i] + 16 * (c0 ^ a_sw[i]));
                const f32x4 hi = *reinterpret_cast<const f32x4*>(st + a_off[i] + 16 * ((c0 + 1) ^ a_sw[i]));
                bf16x8 a[3];
                if (DVQ_ABL_IS(p, 4)) {                       // diag: no split arithmetic
                    a[0] = __builtin_bit_cast(bf16x8, lo); a[1] = __builtin_bit_cast(bf16x8, hi); a[2] = a[0];
                } else split_frag(lo, hi, a);
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    f32x16 c = acc[i][jn];
                    if constexpr (SWAP) {       // weights as operand A: lanes <-> rows, registers <-> columns (16-byte stores)
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][0], a[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][2], a[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][1], a[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][0], a[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][1], a[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][0], a[0], c, 0, 0, 0);
                    } else {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], w[jn][0], c, 0, 0, 0);     // a3 b1
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[jn][2], c, 0, 0, 0);     // a1 b3
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[jn][1], c, 0, 0, 0);     // a2 b2
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], w[jn][0], c, 0, 0, 0);     // a2 b1
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[jn][1], c, 0, 0, 0);     // a1 b2
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[jn][0], c, 0, 0, 0);     // a1 b1
                    }
                    acc[i][jn] = c;
                }
            }
        }
        if (!more) break;
        stage ^= 1;
    }
    if (DVQ_CLK(p) && blockIdx.x == 0 && tid == 0) {
        p.dbg_clk[2] = __builtin_amdgcn_s_memtime();
        p.dbg_clk[3] = __builtin_amdgcn_s_memrealtime();
    }
    if (DVQ_CLK(p) && tid == 0) p.dbg_clk[8 + 4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    if constexpr (SWAP) {
        gemm_epilogue_t<EPI>(p, acc, m0, n0, nt, tid);
    } else {
        __syncthreads();
        gemm_epilogue<EPI>(p, acc, m0, n0, mt, nt, tid, reinterpret_cast<float*>(smem_c));
    }
    if (DVQ_CLK(p)) {
        __syncthreads();
        if (tid == 0) p.dbg_clk[8 + 4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int EPI>
int launch_dma(const GemmParams& p, hipStream_t stream) {
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_dma_kernel<EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)D_SMEM);
        });
        if (e != hipSuccess) {
            dvq_set_error("gemm_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const long tiles_m = (p.M + BM - 1) / BM;
    const long tiles_n = (p.N + BN - 1) / BN;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    static const char* const names[] = {"gemm_bias", "gemm_resid", "gemm_gate", "gemm_colmax", "gemm_argmin"};
    double ksum = 0;
    for (int s = 0; s < p.nsrc; ++s) ksum += p.src[s].K;
    static unsigned long long* clk_buf = nullptr;
    GemmParams q = p;
#ifdef DVQ_GEMM_DIAG
    if (getenv("DVQ_GEMM_CLK")) {
        if (!clk_buf) (void)hipMalloc(&clk_buf, 64 + 32 * 65536);
        q.dbg_clk = clk_buf;
    }
#endif
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        DVQ_LAUNCH((gemm_bf16x3_dma_kernel<EPI>), dim3((unsigned)grid), dim3(256), D_SMEM, stream, q);
    }
    if (q.dbg_clk) {
        unsigned long long hbuf[4];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(hbuf, clk_buf, 32, hipMemcpyDeviceToHost);
        const double cyc = (double)(hbuf[2] - hbuf[0]), ref = (double)(hbuf[3] - hbuf[1]);
        fprintf(stderr, "[dvq clk] block 0 main loop: %.0f memtime ticks, %.0f x 10 ns -> memtime ticks at %.1f MHz, span %.1f us\n", cyc, ref,
                ref > 0 ? cyc / ref * 100.0 : 0.0, ref * 0.01);
        if (grid <= 65536) {                                 // per-block phases (10 ns ticks relative to the first start)
            std::vector<unsigned long long> hb((size_t)grid * 4);
            (void)hipMemcpy(hb.data(), clk_buf + 8, (size_t)grid * 32, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, tend = 0;
            for (long b = 0; b < grid; ++b) { if (hb[4 * b] < t0) t0 = hb[4 * b]; if (hb[4 * b + 2] > tend) tend = hb[4 * b + 2]; }
            double lo = 0, ep = 0, st_second = 0; long n2 = 0;
            for (long b = 0; b < grid; ++b) {
                lo += (double)(hb[4 * b + 1] - hb[4 * b]); ep += (double)(hb[4 * b + 2] - hb[4 * b + 1]);
                if (b >= 512) { st_second += (double)(hb[4 * b] - t0); ++n2; }
            }
            fprintf(stderr, "[dvq clk] %ld blocks: mean loop %.1f us, mean epilogue %.1f us, first start..last end %.1f us, mean start of blocks >= 512: %.1f us\n",
                    grid, lo / grid * 0.01, ep / grid * 0.01, (double)(tend - t0) * 0.01, n2 ? st_second / n2 * 0.01 : 0.0);
        }
    }
    DVQ_CHECK_LAUNCH("gemm_bf16x3_dma");
    return DVQ_OK;
}

// ================================================================================================================
// Wide variant for N % 256 == 0 (the gated PixelCNN GEMMs: N = 512): tile 128 x 256, eight waves as 2 x 4 of 64 x 64, one
// workgroup per CU.  The fp32 activation tile goes through registers and is split ONCE per element into its three bf16
// planes before it reaches the LDS (44 vector instructions per thread and K-tile instead of 176 per wave in the 128 x 128
// kernel, where every wave splits the fragments it reads); the weight planes arrive by LDS-DMA, 6 pieces per wave and K-tile
// for the same 48 MFMAs per wave (10 in the 128 x 128 kernel: its 4 activation pieces are gone).  Both operands sit in the LDS
// as bf16 planes with 64-byte rows, chunk c of row r at c ^ ((r >> 2) & 3).  Same accumulation order as the 128 x 128 kernel:
// bit-identical results (tests/test_gpu_parity.py::test_gemm_tile_variants_agree_bitwise).
constexpr int W_A_PL = 128 * 64;                           // one activation plane [128][32] bf16
constexpr int W_W_PL = 256 * 64;                           // one weight plane [256][32] bf16
constexpr int W_STAGE = 3 * W_A_PL + 3 * W_W_PL;           // 73 728 B
constexpr size_t W_SMEM = 2 * W_STAGE;                     // 147 456 B: one workgroup per CU

struct WideCursor {
    int s, k_left;
    const float* a_ptr;
    const uint16_t* w_ptr[6];

    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m0, int n0, int tid, int wave, int lane) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        {
            long m = m0 + (tid >> 2);
            if (m >= p.M) m = p.M - 1;                     // clamped rows only feed outputs the epilogue masks
            a_ptr = src.A + m * src.lda + 8 * (tid & 3);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {                      // 48 pieces of 16 rows x 64 B, six per wave
            const int id = wave * 6 + i;
            const int pl = id >> 4, rb = id & 15;
            const int row = rb * 16 + (lane >> 2);
            w_ptr[i] = src.Wp + pl * src.wp_plane + (long)(n0 + row) * src.ldw + 8 * ((lane & 3) ^ ((row >> 2) & 3));
        }
    }
    __device__ __forceinline__ bool valid() const { return k_left > 0; }
    struct ARegs { f32x4 lo, hi; };
    __device__ __forceinline__ ARegs load_a() const {
        ARegs r;
        r.lo = *reinterpret_cast<const f32x4*>(a_ptr);
        r.hi = *reinterpret_cast<const f32x4*>(a_ptr + 4);
        return r;
    }
    __device__ __forceinline__ void issue_w(char* stage, int wave) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = wave * 6 + i;
            const int pl = id >> 4, rb = id & 15;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                             (__attribute__((address_space(3))) void*)(stage + 3 * W_A_PL + pl * W_W_PL + rb * 1024), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void advance(const GemmParams& p, long m0, int n0, int tid, int wave, int lane) {
        a_ptr += BK;
#pragma unroll
        for (int i = 0; i < 6; ++i) w_ptr[i] += BK;
        k_left -= BK;
        if (k_left <= 0) open(p, s + 1, m0, n0, tid, wave, lane);
    }
};

// split the thread's eight activation values and write the three 16-byte pieces
__device__ __forceinline__ void wide_store_a(char* stage, int tid, const WideCursor::ARegs& a) {
    bf16x8 pl[3];
    split_frag(a.lo, a.hi, pl);
    const int row = tid >> 2, q = tid & 3;
    char* dst = stage + row * 64 + 16 * (q ^ ((row >> 2) & 3));
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<bf16x8*>(dst + i * W_A_PL) = pl[i];
}

template <int EPI, bool DEPHASE>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3_wide_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
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
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.f;

    WideCursor cur;
    cur.open(p, 0, m0, n0, tid, wave, lane);
    WideCursor::ARegs areg = cur.load_a();
    cur.issue_w(smem_c, wave);
    cur.advance(p, m0, n0, tid, wave, lane);
    wide_store_a(smem_c, tid, areg);
    bool more = cur.valid();
    if (more) areg = cur.load_a();                        // tile 1's activations: stored into the other stage during tile 0
    int a_off[2], a_sw[2], w_off[2], w_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int arow = wm * 64 + i * 32 + r, wrow = wn * 64 + i * 32 + r;
        a_off[i] = arow * 64;
        a_sw[i] = (arow >> 2) & 3;
        w_off[i] = 3 * W_A_PL + wrow * 64;
        w_sw[i] = (wrow >> 2) & 3;
    }
    int stage = 0;
    while (true) {
        dvq_dma_barrier();                                 // this stage is complete (weight DMA landed, activation planes written);
                                                           // everybody is done reading the other stage
        const char* st = smem_c + stage * W_STAGE;
        char* nx = smem_c + (stage ^ 1) * W_STAGE;
        // The two waves of a SIMD (w and w + 4) are in lock step (one workgroup per CU, one barrier per K-tile): waves 0..3
        // feed the next stage at the top of the tile, waves 4..7 between the two k-steps, so that one of the pair runs MFMAs
        // while the other sits in the DMA issue (an LDS-DMA piece holds the issuing wave for ~150 cycles).
        auto feed = [&]() {
            wide_store_a(nx, tid, areg);                   // loaded a K-tile ago; before the DMA issue: a wait for these loads
            cur.issue_w(nx, wave);                         // after it would also wait for the DMA (vmcnt counts in order)
            cur.advance(p, m0, n0, tid, wave, lane);
            if (cur.valid()) areg = cur.load_a();
        };
        if (more && (wave < 4 || !DEPHASE)) feed();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (DEPHASE && ks == 1 && more && wave >= 4) feed();
            bf16x8 w[2][3], a[2][3];
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    w[jn][pl] = *reinterpret_cast<const bf16x8*>(st + w_off[jn] + pl * W_W_PL + 16 * ((2 * ks + h) ^ w_sw[jn]));
                    a[jn][pl] = *reinterpret_cast<const bf16x8*>(st + a_off[jn] + pl * W_A_PL + 16 * ((2 * ks + h) ^ a_sw[jn]));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    f32x16 c = acc[i][jn];           // weights as operand A: lanes <-> rows, registers <-> columns (16-byte stores)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][0], a[i][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][2], a[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][1], a[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][0], a[i][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][1], a[i][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[jn][0], a[i][0], c, 0, 0, 0);
                    acc[i][jn] = c;
                }
        }
        if (!more) break;
        more = cur.valid();
        stage ^= 1;
    }
    gemm_epilogue_t_at<EPI>(p, acc, m0, n0, tid, wm, wn);
}

template <int EPI>
int launch_wide(const GemmParams& p, hipStream_t stream) {
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_wide_kernel<EPI, false>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_SMEM);
            const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_wide_kernel<EPI, true>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)W_SMEM);
            return e0 != hipSuccess ? e0 : e1;
        });
        if (e != hipSuccess) {
            dvq_set_error("gemm_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const long tiles_m = (p.M + 127) / 128;
    const long tiles_n = p.N / 256;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    static const char* const names[] = {"gemm_bias", "gemm_resid", "gemm_gate"};
    double ksum = 0;
    for (int s = 0; s < p.nsrc; ++s) ksum += p.src[s].K;
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        if (dvq_knobs().gemm_dephase) DVQ_LAUNCH((gemm_bf16x3_wide_kernel<EPI, true>), dim3((unsigned)grid), dim3(512), W_SMEM, stream, p);
        else DVQ_LAUNCH((gemm_bf16x3_wide_kernel<EPI, false>), dim3((unsigned)grid), dim3(512), W_SMEM, stream, p);
    }
    DVQ_CHECK_LAUNCH("gemm_bf16x3_wide");
    return DVQ_OK;
}

// ================================================================================================================
// Skinny variant for small M (the reference's own call pattern: GenNet.gen with B = 1 per call, 1 / 20 / 49 / 100 grasps per
// object, gen_diverse_grasp_ho3d.py:212-236).  With M <= 128 the tiled kernels above run one workgroup per 128 or 256 output
// columns, each streaming its whole weight panel through ONE CU (125 us per gated GEMM at M = 1: 4 workgroups busy).  Here every
// WAVE owns one 32 x 32 output block over the full K: weights go global -> registers (three bf16 planes), the activation
// fragment is split in registers, and the six partial products run in EXACTLY the order of the tiled kernels -- the same MFMA
// sequence on the same operands, so the result is bit-identical to the batched path (tests: skinny == tiled bitwise, gen
// batched == loop of B = 1 calls).  A workgroup is two waves: two adjacent column blocks, which for the gate epilogue are a
// tanh block and its sigmoid partner (exchanged through 4 KB of LDS).  Grid = (N / 64) x (M / 32) workgroups.
//
// Loads are shaped in whole 128-byte lines: a lane (row r, half h) fetches 64 contiguous bytes of its weight row per plane
// (four k-steps of 16) and 128 bytes of its activation row with back-to-back 16-byte loads, so every line is touched by
// consecutive instructions and crosses L2 -> L1 once.  The 16-byte pieces then sit in the wrong lane half for two of the four
// k-steps; v_permlane32_swap puts them where the MFMA operand layout wants them (lane half h <-> k = 8h .. 8h+7).
// Measured (tools/gemm_skinny_bench.py, M = 1, N = 1024, K = 2048, launches issued from Python): 30.5 us against 125 us for
// the tiled kernel, the same on L2-warm and on rotating panels (not memory-bound: a k-step costs ~370 cycles -- six dependent
// MFMAs = 192, and ~55 vector instructions (split 44, swaps 10) that the in-order wave issues between them).  Tried: one
// k-step per load (32 bytes of each of 32 lines per instruction: 4x the L1 fill traffic, 590 cycles per k-step), 12 k-steps
// in flight (no change), four waves sharing the split through the LDS with a barrier per chunk (49.8 us: the round trip
// serialises), prefetch helper workgroups sweeping the panel into the Infinity Cache (-7 %, kept).
struct SkinnyChunk {
    uint4 w[3][4];      // after load: piece 4h+q of the row's 128 bytes; after fix(): [0],[2],[1],[3] = k-steps 0,1,2,3
    f32x4 a[8];         // after load: piece 8h+q of the row's 256 bytes; after fix(): (a[0],a[1]),(a[4],a[5]),(a[2],a[3]),(a[6],a[7])
};

__device__ __forceinline__ void swap_halves(uint4& x, uint4& y) {      // x.hi <-> y.lo (per dword)
    unsigned* px = reinterpret_cast<unsigned*>(&x);
    unsigned* py = reinterpret_cast<unsigned*>(&y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(px[i], py[i], false, false);
        px[i] = r[0];
        py[i] = r[1];
    }
}
__device__ __forceinline__ void swap_halves(f32x4& x, f32x4& y) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[i]), __float_as_uint(y[i]), false, false);
        x[i] = __uint_as_float(r[0]);
        y[i] = __uint_as_float(r[1]);
    }
}

struct SkinnyCursor {
    int s, k_left;
    const float* a_ptr;
    const uint16_t* w_ptr[3];

    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m, int n, int h) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        a_ptr = src.A + m * src.lda + 32 * h;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w_ptr[pl] = src.Wp + pl * src.wp_plane + (long)n * src.ldw + 32 * h;
    }
    __device__ __forceinline__ void load(const GemmParams& p, long m, int n, int h, SkinnyChunk& t) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int q = 0; q < 4; ++q) t.w[pl][q] = *reinterpret_cast<const uint4*>(w_ptr[pl] + 8 * q);
            w_ptr[pl] += 64;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) t.a[q] = *reinterpret_cast<const f32x4*>(a_ptr + 4 * q);
        a_ptr += 64;
        k_left -= 64;
        if (k_left <= 0) open(p, s + 1, m, n, h);
    }
};

__device__ __forceinline__ void skinny_fix(SkinnyChunk& t) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        swap_halves(t.w[pl][0], t.w[pl][1]);     // -> k-steps 0 and 2
        swap_halves(t.w[pl][2], t.w[pl][3]);     // -> k-steps 1 and 3
    }
    swap_halves(t.a[0], t.a[2]);                 // -> k-step 0 low, k-step 2 low
    swap_halves(t.a[1], t.a[3]);
    swap_halves(t.a[4], t.a[6]);                 // -> k-step 1 low, k-step 3 low
    swap_halves(t.a[5], t.a[7]);
}

__device__ __forceinline__ void skinny_mfma(const uint4& u0, const uint4& u1, const uint4& u2, const bf16x8 (&a)[3], f32x16& c) {
    const bf16x8 w0 = __builtin_bit_cast(bf16x8, u0), w1 = __builtin_bit_cast(bf16x8, u1), w2 = __builtin_bit_cast(bf16x8, u2);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a[2], c, 0, 0, 0);       // the tiled kernels' order (weights as operand A)
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, a[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, a[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, a[0], c, 0, 0, 0);
}
// one dependent chain of six MFMAs; the in-order wave issues nothing between them unless it is PLACED there: the split of the
// next k-step's activation fragment (44 vector instructions) goes into the six gaps, 1 MFMA : 8 VALU
#define DVQ_SK_INTERLEAVE()                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) {                \
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);              \
        __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);              \
    }
// four k-steps of a chunk, in k order; `a0` = the split fragment of its first k-step (prepared under the previous chunk's last
// MFMAs), `next` = the chunk that follows (its lane fix-up and first split are prepared under this chunk's last MFMAs)
__device__ __forceinline__ void skinny_chunk(SkinnyChunk& t, SkinnyChunk& next, bf16x8 (&a0)[3], f32x16& c) {
    bf16x8 a1[3], a2[3], a3[3];
    split_frag(t.a[4], t.a[5], a1);
    skinny_mfma(t.w[0][0], t.w[1][0], t.w[2][0], a0, c);
    DVQ_SK_INTERLEAVE();
    split_frag(t.a[2], t.a[3], a2);
    skinny_mfma(t.w[0][2], t.w[1][2], t.w[2][2], a1, c);
    DVQ_SK_INTERLEAVE();
    split_frag(t.a[6], t.a[7], a3);
    skinny_mfma(t.w[0][1], t.w[1][1], t.w[2][1], a2, c);
    DVQ_SK_INTERLEAVE();
    skinny_fix(next);
    split_frag(next.a[0], next.a[1], a0);
    skinny_mfma(t.w[0][3], t.w[1][3], t.w[2][3], a3, c);
#pragma unroll
    for (int i_ = 0; i_ < 6; ++i_) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, 14, 0);
    }
}
// the same without a successor
__device__ __forceinline__ void skinny_chunk_last(SkinnyChunk& t, bf16x8 (&a0)[3], f32x16& c) {
    bf16x8 a1[3], a2[3], a3[3];
    split_frag(t.a[4], t.a[5], a1);
    skinny_mfma(t.w[0][0], t.w[1][0], t.w[2][0], a0, c);
    split_frag(t.a[2], t.a[3], a2);
    skinny_mfma(t.w[0][2], t.w[1][2], t.w[2][2], a1, c);
    split_frag(t.a[6], t.a[7], a3);
    skinny_mfma(t.w[0][1], t.w[1][1], t.w[2][1], a2, c);
    skinny_mfma(t.w[0][3], t.w[1][3], t.w[2][3], a3, c);
}

// Helper workgroups (blockIdx.x >= n_work): a gated GEMM at M <= 32 occupies 16 of the 256 CUs.  The idle CUs sweep the launch's
// weight planes once, in whole 1 KiB wave reads, which brings them into the memory-side Infinity Cache just ahead of the
// compute waves (the data itself is discarded).
__device__ __forceinline__ void skinny_prefetch(const GemmParams& p, int helper, int n_helpers, int tid) {
    unsigned sink = 0;
    for (int s = 0; s < p.nsrc; ++s) {
        const GemmSrc& src = p.src[s];
        const long bytes = (long)p.N * src.ldw * 2;                      // one plane's rows [0, N)
        for (int pl = 0; pl < 3; ++pl) {
            const char* base = reinterpret_cast<const char*>(src.Wp + pl * src.wp_plane);
            for (long off = ((long)helper * 128 + tid) * 16; off < bytes; off += (long)n_helpers * 2048 * 4) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long o = off + (long)u * n_helpers * 2048;
                    v[u] = o < bytes ? *reinterpret_cast<const uint4*>(base + o) : uint4{0u, 0u, 0u, 0u};
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) sink ^= v[u].x ^ v[u].w;
            }
        }
    }
    if (sink == 0x9e3779b9u && p.M < 0) p.out[0] = 0.f;                   // never true: keeps the loads alive
}

// Epilogues of the skinny kernels: lane <-> row m, registers 4g..4g+3 <-> four consecutive columns nb0 + 8g + 4h (gemm_common.h,
// gemm_epilogue_t_at: the same arithmetic in the same order).  `xch`: 4 KB of LDS for the gate's tanh / sigmoid exchange.
template <int EPI>
__device__ __forceinline__ void skinny_epilogue(const GemmParams& p, const f32x16& c, float* xch, long m0, int n0, int nb0, int r, int h,
                                                int lane, int wave, bool m_ok) {
    long m;
    m = m0 + r;
    if constexpr (EPI == EPI_BIAS || EPI == EPI_RESID) {
        if (!m_ok) return;
        const bool vec_ok = (p.N % 4 == 0) && (p.ldo % 4 == 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = nb0 + 8 * g + 4 * h;
            if (n >= p.N) continue;
            f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
            if (vec_ok && n + 3 < p.N) {
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                if constexpr (EPI == EPI_RESID) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.ldr + n);
                if (p.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    if (n + cc >= p.N) continue;
                    float x = v[cc] + (p.bias ? p.bias[n + cc] : 0.f);
                    if constexpr (EPI == EPI_RESID) x += p.resid[m * p.ldr + n + cc];
                    if (p.relu) x = fmaxf(x, 0.f);
                    p.out[m * p.ldo + n + cc] = x;
                }
            }
        }
    } else if constexpr (EPI == EPI_GATE) {
        // wave 0 holds the tanh channels na = n0 + ..., wave 1 their sigmoid partners nb = na + 32 (gate-packed order)
        f32x4 mine[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            mine[g] = f32x4{c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
            const int n = nb0 + 8 * g + 4 * h;
            if (p.bias) mine[g] += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.pre && m_ok) *reinterpret_cast<f32x4*>(p.pre + m * p.ldpre + n) = mine[g];
        }
        if (wave == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(xch + (g * 64 + lane) * 4) = mine[g];
        }
        __syncthreads();
        if (wave == 0 && m_ok) {
            const float* crow = p.cls ? p.cls + (long)p.label[m] * p.N : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = 8 * g + 4 * h;
                const int na = n0 + nl, nb = na + 32;
                const int co = (n0 >> 1) + nl;                            // natural output channels
                f32x4 a = mine[g];
                f32x4 gg = *reinterpret_cast<const f32x4*>(xch + (g * 64 + lane) * 4);
                if (crow) {
                    a += *reinterpret_cast<const f32x4*>(crow + na);
                    gg += *reinterpret_cast<const f32x4*>(crow + nb);
                }
                f32x4 o;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) o[qq] = tanhf(a[qq]) * sigmoidf_(gg[qq]);
                *reinterpret_cast<f32x4*>(p.out + m * p.ldo + co) = o;
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(128) void gemm_bf16x3_skinny_kernel(const GemmParams p, int n_col_wgs, int n_work) {
    __shared__ __attribute__((aligned(16))) float xch[64 * 16];          // gate: the sigmoid block's 16 values per lane
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= n_work) {
        skinny_prefetch(p, (int)blockIdx.x - n_work, (int)gridDim.x - n_work, tid);
        return;
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long m0 = (long)((int)blockIdx.x / n_col_wgs) * 32;
    const int n0 = ((int)blockIdx.x % n_col_wgs) * 64;                   // two 32-column blocks per workgroup
    const int nb0 = n0 + 32 * wave;                                      // this wave's block
    long m = m0 + r;
    const bool m_ok = m < p.M;
    if (!m_ok) m = p.M - 1;                                              // clamped rows / columns only feed masked outputs
    int nrow = nb0 + r;
    if (nrow >= p.N) nrow = p.N - 1;

    f32x16 c;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    SkinnyCursor cur;
    cur.open(p, 0, m, nrow, h);
    // two chunks (eight k-steps) in flight behind the one being multiplied; every K is a multiple of 64 here (launch check).
    // The steady state has no branch around a load: with one the compiler's wait insertion falls back to vmcnt(0) before every
    // use, i.e. one full memory latency per chunk.
    int chunks = 0;
    for (int s2 = 0; s2 < p.nsrc; ++s2) chunks += p.src[s2].K >> 6;
    SkinnyChunk q0, q1, q2;
    bf16x8 a0[3];
    if (chunks >= 3) {
        cur.load(p, m, nrow, h, q0);
        cur.load(p, m, nrow, h, q1);
        cur.load(p, m, nrow, h, q2);
        skinny_fix(q0);
        split_frag(q0.a[0], q0.a[1], a0);
        int t = 0;
        for (; t + 6 <= chunks; t += 3) {
            // q0's successor q1 is already in flight; q0 itself is re-loaded only AFTER q1's fix-up has read ... nothing of q0
            skinny_chunk(q0, q1, a0, c); cur.load(p, m, nrow, h, q0);
            skinny_chunk(q1, q2, a0, c); cur.load(p, m, nrow, h, q1);
            // q2's successor is the q0 just loaded: its fix-up waits for that load (two chunks of MFMAs behind it)
            skinny_chunk(q2, q0, a0, c); cur.load(p, m, nrow, h, q2);
        }
        const int rem = chunks - t - 3;                                  // 0 .. 2 chunks not yet loaded; q0 is fixed and split
        skinny_chunk(q0, q1, a0, c); if (rem > 0) cur.load(p, m, nrow, h, q0);
        skinny_chunk(q1, q2, a0, c); if (rem > 1) cur.load(p, m, nrow, h, q1);
        if (rem > 0) {
            skinny_chunk(q2, q0, a0, c);
            if (rem > 1) { skinny_chunk(q0, q1, a0, c); skinny_chunk_last(q1, a0, c); }
            else skinny_chunk_last(q0, a0, c);
        } else skinny_chunk_last(q2, a0, c);
    } else {
        if (chunks > 0) cur.load(p, m, nrow, h, q0);
        if (chunks > 1) cur.load(p, m, nrow, h, q1);
        if (chunks > 0) { skinny_fix(q0); split_frag(q0.a[0], q0.a[1], a0); }
        if (chunks > 1) { skinny_chunk(q0, q1, a0, c); skinny_chunk_last(q1, a0, c); }
        else if (chunks > 0) skinny_chunk_last(q0, a0, c);
    }
    skinny_epilogue<EPI>(p, c, xch, m0, n0, nb0, r, h, lane, wave, m_ok);
}

// ---- LDS-staged variant (round 3).  The register variant above fetches a lane's 128-byte line in four 16-byte pieces with four
// instructions, each touching 32 different lines: 640 line accesses per 64-k chunk and wave, 1 280 per CU -- and a chunk took
// 4 x 320 = 1 280 cycles whatever the split arithmetic or the prefetch depth: one cache line per clock, the vector memory
// path's tag rate.  Here the loads are COALESCED (eight lanes per weight line, sixteen per activation row: 4x fewer line
// accesses) and go through the LDS, whose reads produce the MFMA operand layout (conflict-free through an XOR of the 16-byte
// piece index with (row >> 1) & 7).  Weights: private to the wave that multiplies them (source-side swizzle, lane-linear
// writes, no synchronisation).  Activations: the two waves of the workgroup multiply the SAME rows, so each loads and splits
// half of the chunk (16 rows) into bf16 planes in a shared slot -- half the split arithmetic per wave, none in the MFMA
// loop -- with one two-wave barrier per chunk.  A single in-order wave per SIMD can issue ~48 instructions in the 192 cycles of
// a k-step's six dependent MFMAs: the loop has 6 MFMAs + 6 LDS reads per k-step and ~35 more per k-step for the next chunk.
typedef unsigned sk2_u4 __attribute__((ext_vector_type(4)));   // a native vector (HIP's uint4 is a struct: its copies become memcpy calls
typedef unsigned sk2_u2 __attribute__((ext_vector_type(2)));   // that keep a load -> LDS-store buffer in scratch memory)
constexpr int SK2_PL = 32 * 128;                            // one plane of one operand: 32 rows x 64 k bf16
constexpr int SK2_SLOT = 3 * SK2_PL;                        // 12 288
constexpr int SK2_OFF_A = 2 * 2 * SK2_SLOT;                 // two waves x two weight slots, then two shared activation slots
constexpr int SK2_OFF_X = SK2_OFF_A + 2 * SK2_SLOT;
constexpr int SK2_SMEM = SK2_OFF_X + 4096;                  // 77 824 B

constexpr int SK2_NP = 2;                                   // staging waves per column block: each takes 4 / SK2_NP load instructions
constexpr int SK2_NI = 4 / SK2_NP;
struct Sk2W { sk2_u4 w[3][SK2_NI]; };                       // its share of a chunk's weights as loaded: instruction, this lane's piece
struct Sk2A { f32x4 a[SK2_NI]; };                           // its share of the block's half of the activations

struct Sk2Cursor {                                          // wave-uniform source state of the weight and of the activation stream
    int s, k_left, s_a, k_a;
    const float* a_base;
    const uint16_t* w_base[3];
    int ldw, lda;

    __device__ __forceinline__ void open(const GemmParams& p, int src_i) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        ldw = (int)src.ldw;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w_base[pl] = src.Wp + pl * src.wp_plane;
    }
    __device__ __forceinline__ void open_a(const GemmParams& p, int src_i) {
        s_a = src_i;
        if (s_a >= p.nsrc) { k_a = 0; return; }
        const GemmSrc& src = p.src[s_a];
        k_a = src.K;
        a_base = src.A;
        lda = (int)src.lda;
    }
    // weights: instruction i covers rows 8i .. 8i+7 of the wave's 32, eight lanes per 128-byte line; the lane fetches the piece
    // that belongs at its lane-linear LDS position: logical piece = (lane & 7) ^ f(row), f(row) = (row >> 1) & 7
    __device__ __forceinline__ void load_w(const GemmParams& p, int nb0, int part, int lane, Sk2W& t) {
        const int pe = (lane & 7) ^ ((lane >> 4) & 3);      // f(8i + (lane >> 3)) = (lane >> 4) + 4 (i & 1)
#pragma unroll
        for (int ii = 0; ii < SK2_NI; ++ii) {
            const int i = SK2_NI * part + ii;
            const int n = min(nb0 + 8 * i + (lane >> 3), p.N - 1);          // clamped rows / columns only feed masked outputs
            const int off = n * ldw + 8 * (pe ^ (4 * (i & 1)));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) t.w[pl][ii] = *reinterpret_cast<const sk2_u4*>(w_base[pl] + off);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) w_base[pl] += 64;
        k_left -= 64;
        if (k_left <= 0) open(p, s + 1);
    }
    // activations, half `wave` of the rows (16), this staging wave's instructions: rows 16 wave + 4i .. + 3, sixteen lanes per 256 bytes
    __device__ __forceinline__ void load_a(const GemmParams& p, long m0, int wave, int part, int lane, Sk2A& t) {
#pragma unroll
        for (int ii = 0; ii < SK2_NI; ++ii) {
            const int i = SK2_NI * part + ii;
            const long m = min(m0 + 16 * wave + 4 * i + (lane >> 4), p.M - 1);
            t.a[ii] = *reinterpret_cast<const f32x4*>(a_base + m * lda + 4 * (lane & 15));
        }
        a_base += 64;
        k_a -= 64;
        if (k_a <= 0) open_a(p, s_a + 1);
    }
};

__device__ __forceinline__ void sk2_store_w(char* slot, int part, int lane, const Sk2W& t) {       // lane-linear: 1 KiB per instruction
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int ii = 0; ii < SK2_NI; ++ii) *reinterpret_cast<sk2_u4*>(slot + pl * SK2_PL + 1024 * (SK2_NI * part + ii) + 16 * lane) = t.w[pl][ii];
}
// exact three-way split of the lane's four floats per instruction, 8 bytes per plane into the shared slot
__device__ __forceinline__ void sk2_store_a(char* slot, int wave, int part, int lane, const Sk2A& t) {
#pragma unroll
    for (int ii = 0; ii < SK2_NI; ++ii) {
        const int row = 16 * wave + 4 * (SK2_NI * part + ii) + (lane >> 4);
        unsigned pa[3], pb[3];
        split3_pair(t.a[ii][0], t.a[ii][1], pa[0], pa[1], pa[2]);
        split3_pair(t.a[ii][2], t.a[ii][3], pb[0], pb[1], pb[2]);
        char* dst = slot + row * 128 + 16 * (((lane & 15) >> 1) ^ ((row >> 1) & 7)) + 8 * (lane & 1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<sk2_u2*>(dst + pl * SK2_PL) = sk2_u2{pa[pl], pb[pl]};
    }
}

struct Sk2Frag { sk2_u4 w[3], a[3]; };                     // one k-step in the MFMA operand layout, both operands as planes
__device__ __forceinline__ void sk2_read(const char* wslot, const char* aslot, int j, int r, int h, Sk2Frag& f) {
    const int o = r * 128 + 16 * ((2 * j + h) ^ ((r >> 1) & 7));
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        f.w[pl] = *reinterpret_cast<const sk2_u4*>(wslot + pl * SK2_PL + o);
        f.a[pl] = *reinterpret_cast<const sk2_u4*>(aslot + pl * SK2_PL + o);
    }
}
__device__ __forceinline__ void sk2_mfma(const Sk2Frag& f, f32x16& c) {
    const bf16x8 a[3] = {__builtin_bit_cast(bf16x8, f.a[0]), __builtin_bit_cast(bf16x8, f.a[1]), __builtin_bit_cast(bf16x8, f.a[2])};
    skinny_mfma(__builtin_bit_cast(uint4, f.w[0]), __builtin_bit_cast(uint4, f.w[1]), __builtin_bit_cast(uint4, f.w[2]), a, c);
}
// the four k-steps of one chunk; the reads of k-step j + 1 are issued before the six MFMAs of k-step j
__device__ __forceinline__ void sk2_chunk(const char* wslot, const char* aslot, int r, int h, f32x16& c) {
    Sk2Frag f0, f1;
    sk2_read(wslot, aslot, 0, r, h, f0);
    sk2_read(wslot, aslot, 1, r, h, f1);
    sk2_mfma(f0, c);
    sk2_read(wslot, aslot, 2, r, h, f0);
    sk2_mfma(f1, c);
    sk2_read(wslot, aslot, 3, r, h, f1);
    sk2_mfma(f0, c);
    sk2_mfma(f1, c);
}

template <int EPI>
__global__ __launch_bounds__(128 + 128 * SK2_NP) void gemm_bf16x3_skinny2_kernel(const GemmParams p, int n_col_wgs, int n_work) {
    extern __shared__ __attribute__((aligned(16))) char sk2_smem[];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= n_work) {
        if (tid < 128) skinny_prefetch(p, (int)blockIdx.x - n_work, (int)gridDim.x - n_work, tid);
        return;
    }
    // Waves 0 and 1 multiply (two adjacent 32-column blocks: one dependent MFMA chain each, on SIMDs of their own); waves 2 .. stage
    // for them: wave 2 + w + 2 part loads its share of block w's weights and of half w of the activations, splits, writes the LDS.
    // A single in-order wave that did both spent 1 900 cycles per chunk: its ~220 staging instructions do not overlap with its
    // own chain of dependent MFMAs unless every one of them is placed by hand.  One barrier per chunk for all waves.  What is left
    // is the chain itself: a dependent v_mfma_f32_32x32x16 (same accumulator) issues every ~41 cycles, not every 32 -- measured
    // 0.47 us per 24-MFMA chunk with two or four staging waves and with either consumer schedule.
    const int lane = tid & 63;
    const int wave4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave4 >= 2;
    const int wave = wave4 & 1;                                          // the column block this wave multiplies or stages
    const int part = (wave4 >> 1) - 1;                                   // staging waves: which of the block's load instructions
    const int r = lane & 31, h = lane >> 5;
    const long m0 = (long)((int)blockIdx.x / n_col_wgs) * 32;
    const int n0 = ((int)blockIdx.x % n_col_wgs) * 64;
    const int nb0 = n0 + 32 * wave;
    const bool m_ok = m0 + r < p.M;
    char* w0 = sk2_smem + wave * 2 * SK2_SLOT;                           // block `wave`'s weight slots
    char* w1 = w0 + SK2_SLOT;
    char* a0 = sk2_smem + SK2_OFF_A;                                     // the workgroup's activation slots
    char* a1 = a0 + SK2_SLOT;
    float* xch = reinterpret_cast<float*>(sk2_smem + SK2_OFF_X);
    int chunks = 0;
    for (int s2 = 0; s2 < p.nsrc; ++s2) chunks += p.src[s2].K >> 6;     // every K is a multiple of 64 here (launch check)

    if (producer) {
        // chunk t + 1 is written while chunk t is multiplied; its loads were issued two iterations earlier
        Sk2Cursor cur;
        cur.open(p, 0);
        cur.open_a(p, 0);
        Sk2W wa, wb;
        Sk2A aa, ab;
        if (chunks > 0) {
            cur.load_w(p, nb0, part, lane, wa); cur.load_a(p, m0, wave, part, lane, aa);                        // chunk 0
            if (chunks > 1) { cur.load_w(p, nb0, part, lane, wb); cur.load_a(p, m0, wave, part, lane, ab); }    // chunk 1
            sk2_store_w(w0, part, lane, wa);
            sk2_store_a(a0, wave, part, lane, aa);
            if (chunks > 2) { cur.load_w(p, nb0, part, lane, wa); cur.load_a(p, m0, wave, part, lane, aa); }    // chunk 2
        }
        __syncthreads();                                                 // chunk 0 is in its slots
        int t = 0;
        for (; t + 4 < chunks; t += 2) {                                 // no branch around a load in the steady state
            sk2_store_w(w1, part, lane, wb); sk2_store_a(a1, wave, part, lane, ab);
            cur.load_w(p, nb0, part, lane, wb); cur.load_a(p, m0, wave, part, lane, ab);
            __syncthreads();
            sk2_store_w(w0, part, lane, wa); sk2_store_a(a0, wave, part, lane, aa);
            cur.load_w(p, nb0, part, lane, wa); cur.load_a(p, m0, wave, part, lane, aa);
            __syncthreads();
        }
        const int rem = chunks - t;                                      // 0 .. 4 chunks left; one barrier per chunk, as the consumers
        if (rem > 0) {
            if (rem > 1) { sk2_store_w(w1, part, lane, wb); sk2_store_a(a1, wave, part, lane, ab); }
            if (rem > 3) { cur.load_w(p, nb0, part, lane, wb); cur.load_a(p, m0, wave, part, lane, ab); }       // chunk t + 3
            __syncthreads();
        }
        if (rem > 1) {
            if (rem > 2) { sk2_store_w(w0, part, lane, wa); sk2_store_a(a0, wave, part, lane, aa); }
            __syncthreads();
        }
        if (rem > 2) {
            if (rem > 3) { sk2_store_w(w1, part, lane, wb); sk2_store_a(a1, wave, part, lane, ab); }
            __syncthreads();
        }
        if (rem > 3) __syncthreads();
        if constexpr (EPI == EPI_GATE) __syncthreads();                  // the consumers' exchange barrier
        return;
    }
    f32x16 c;
#pragma unroll
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    __syncthreads();                                                     // chunk 0 is in its slots
    // Software pipeline over the chunks: all four k-steps of chunk t are in registers when its MFMAs start, so the barrier that
    // hands slot t back to the staging waves and publishes chunk t + 1 is taken FIRST, and chunk t + 1 is read under chunk t's
    // 24 MFMAs (k-step by k-step: the wave is in-order).
    Sk2Frag f[4], g[4];
    if (chunks > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sk2_read(w0, a0, j, r, h, f[j]);
    }
    for (int t = 0; t < chunks; t += 2) {
        __syncthreads();                                                 // chunk t + 1 written, chunk t read
        if (t + 1 < chunks) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { sk2_read(w1, a1, j, r, h, g[j]); sk2_mfma(f[j], c); }
            __syncthreads();
            if (t + 2 < chunks) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { sk2_read(w0, a0, j, r, h, f[j]); sk2_mfma(g[j], c); }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) sk2_mfma(g[j], c);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) sk2_mfma(f[j], c);
        }
    }
    skinny_epilogue<EPI>(p, c, xch, m0, n0, nb0, r, h, lane, wave, m_ok);
}

constexpr long SKINNY_MAX_M = 256;     // above this the tiled kernels win (every 32-row block re-reads the weight panel from L2)

template <int EPI>
int launch_skinny(const GemmParams& p, hipStream_t stream) {
    static const char* const names[] = {"gemm_bias", "gemm_resid", "gemm_gate"};
    double ksum = 0;
    for (int s = 0; s < p.nsrc; ++s) ksum += p.src[s].K;
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        const int n_col = (int)((p.N + 63) / 64), n_work = n_col * (int)((p.M + 31) / 32);
        // helpers only where most of the chip would idle AND the panel is worth it (>= 1 MB of planes)
        const bool big = 6.0 * p.N * ksum >= 1.0e6;
        const int helpers = (dvq_knobs().gemm_skinny_prefetch && big && n_work <= 64) ? 192 : 0;
        if (dvq_knobs().gemm_skinny == 2) {                               // the register-staged variant (A/B runs)
            DVQ_LAUNCH((gemm_bf16x3_skinny_kernel<EPI>), dim3((unsigned)(n_work + helpers)), dim3(128), 0, stream, p, n_col, n_work);
        } else {
            static DvqOncePerDevice attr_once;
            const hipError_t e = attr_once.run([] {
                return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_skinny2_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SK2_SMEM);
            });
            if (e != hipSuccess) {
                dvq_set_error("gemm_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                return DVQ_ELAUNCH;
            }
            DVQ_LAUNCH((gemm_bf16x3_skinny2_kernel<EPI>), dim3((unsigned)(n_work + helpers)), dim3(128 + 128 * SK2_NP), SK2_SMEM, stream, p, n_col, n_work);
        }
    }
    DVQ_CHECK_LAUNCH("gemm_bf16x3_skinny");
    return DVQ_OK;
}

__global__ void split_bf16x3_kernel(const float* __restrict__ w, long n, uint16_t* __restrict__ planes) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    const unsigned a = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(a);
    const unsigned b = __float_as_uint(r) & 0xffff0000u;
    const float t = r - __uint_as_float(b);
    planes[i] = (uint16_t)(a >> 16);
    planes[n + i] = (uint16_t)(b >> 16);
    planes[2 * n + i] = (uint16_t)(__float_as_uint(t) >> 16);
}

template <int EPI, bool WPLANES>
int launch(const GemmParams& p, hipStream_t stream) {
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16x3_kernel<EPI, WPLANES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BYTES);
        });
        if (e != hipSuccess) {
            dvq_set_error("gemm_bf16x3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const long tiles_m = (p.M + BM - 1) / BM;
    const long tiles_n = (p.N + BN - 1) / BN;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    static const char* const names[] = {"gemm_bias", "gemm_resid", "gemm_gate", "gemm_colmax", "gemm_argmin"};
    double ksum = 0;
    for (int s = 0; s < p.nsrc; ++s) ksum += p.src[s].K;
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        DVQ_LAUNCH((gemm_bf16x3_kernel<EPI, WPLANES>), dim3((unsigned)grid), dim3(256), SMEM_BYTES, stream, p);
    }
    DVQ_CHECK_LAUNCH("gemm_bf16x3");
    return DVQ_OK;
}

}  // namespace

// Called by dvq_launch_gemm (gemm_f32.hip) after argument validation.  K of every source must be a multiple of 16.
int dvq_launch_gemm_bf16x3(const GemmParams& p, GemmEpilogue epi, hipStream_t stream) {
#ifdef DVQ_GEMM_DIAG
    if (const char* e = getenv("DVQ_GEMM_ABL")) const_cast<GemmParams&>(p).dbg_abl = atoi(e);
    static const bool use_dma = !(getenv("DVQ_GEMM_NODMA") && getenv("DVQ_GEMM_NODMA")[0] == '1');
#else
    constexpr bool use_dma = true;
#endif
    bool planes = true;                                              // pre-split weights: all-or-nothing per launch
    for (int s = 0; s < p.nsrc; ++s) planes = planes && p.src[s].Wp != nullptr;
    if (planes && use_dma) {
        bool aligned = true;                       // the DMA moves 16-byte chunks: every pointer and row stride must allow it
        for (int s = 0; s < p.nsrc; ++s)
            aligned = aligned && ((reinterpret_cast<uintptr_t>(p.src[s].Wp) & 15) == 0) && (p.src[s].wp_plane % 8 == 0) &&
                      (p.src[s].ldw % 8 == 0);
        if (aligned && dvq_knobs().gemm_skinny && p.M <= SKINNY_MAX_M) {
            bool a_ok = true;                      // 16-byte activation loads; the gate's two blocks need N % 64 == 0
            for (int s = 0; s < p.nsrc; ++s)
                a_ok = a_ok && ((reinterpret_cast<uintptr_t>(p.src[s].A) & 15) == 0) && (p.src[s].lda % 4 == 0) && (p.src[s].K % 64 == 0);
            if (a_ok) switch (epi) {
                case EPI_BIAS: return launch_skinny<EPI_BIAS>(p, stream);
                case EPI_RESID: return launch_skinny<EPI_RESID>(p, stream);
                case EPI_GATE: if (p.N % 64 == 0) return launch_skinny<EPI_GATE>(p, stream); break;
                default: break;
            }
        }
        // 128 x 256 tiles where they fill the chip (N = 256 leaves one tile column: the 128 x 128 kernel is faster there)
        const bool use_wide = dvq_knobs().gemm_wide != 0;
        if (aligned && use_wide && p.N % 256 == 0 && p.N >= 512) {
            bool a_ok = true;                      // 16-byte activation loads
            for (int s = 0; s < p.nsrc; ++s)
                a_ok = a_ok && ((reinterpret_cast<uintptr_t>(p.src[s].A) & 15) == 0) && (p.src[s].lda % 4 == 0);
            if (a_ok) switch (epi) {
                case EPI_BIAS: return launch_wide<EPI_BIAS>(p, stream);
                case EPI_RESID: return launch_wide<EPI_RESID>(p, stream);
                case EPI_GATE: return launch_wide<EPI_GATE>(p, stream);
                default: break;
            }
        }
        if (aligned) {
            switch (epi) {
                case EPI_BIAS: return launch_dma<EPI_BIAS>(p, stream);
                case EPI_RESID: return launch_dma<EPI_RESID>(p, stream);
                case EPI_GATE: return launch_dma<EPI_GATE>(p, stream);
                case EPI_COLMAX: return launch_dma<EPI_COLMAX>(p, stream);
                default: break;
            }
        }
    }
    switch (epi) {
        case EPI_BIAS: return planes ? launch<EPI_BIAS, true>(p, stream) : launch<EPI_BIAS, false>(p, stream);
        case EPI_RESID: return planes ? launch<EPI_RESID, true>(p, stream) : launch<EPI_RESID, false>(p, stream);
        case EPI_GATE: return planes ? launch<EPI_GATE, true>(p, stream) : launch<EPI_GATE, false>(p, stream);
        case EPI_COLMAX: return planes ? launch<EPI_COLMAX, true>(p, stream) : launch<EPI_COLMAX, false>(p, stream);
        default: break;
    }
    dvq_set_error("gemm_bf16x3: epilogue %d is not available on the split-bf16 path", (int)epi);
    return DVQ_EINVAL;
}

extern "C" int dvq_split_bf16x3(const float* w, int64_t n, uint16_t* planes, dvq_stream_t stream) {
    DVQ_REQUIRE(n >= 0 && (n == 0 || (w && planes)), "split_bf16x3: null pointer");
    if (n == 0) return DVQ_OK;
    DVQ_LAUNCH(split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (long)n, planes);
    DVQ_CHECK_LAUNCH("split_bf16x3");
    return DVQ_OK;
}
