// fp32-class GEMM on the fp16 matrix cores with THREE products per fp32 product ("f16x2", the default of the path's GEMMs).
//
//   weights (packer, once):  w~ = w * 2^t_n (one power of two per output row n: max_k |w~| in [2^14, 2^15)),
//                            w1 = fp16(w~), w2 = fp16((w~ - w1) * 2^11)                      (w~ - w1 is exact in fp32)
//   activations (staging):   a1 = fp16(a),  a2 = fp16((a - a1) * 2^11)
//   two fp32 accumulators:   hi += a1 w1 ;  lo += a1 w2 ; lo += a2 w1        (v_mfma_f32_16x16x32_f16: products exact in fp32)
//   epilogue:                out = (hi + lo * 2^-11) * 2^-t_n (+ bias, ...)
//
// fp16 carries 11 significant bits and the second piece is stored scaled by 2^11, so it stays a NORMAL fp16 number wherever
// the first piece is one: 22 bits of every operand down to |a| = 2^-14; below that the absolute error is <= 2^-36.  Dropped:
// a2 w2 and the two representation residuals, together <= 3 * 2^-22 |a w| per product; the cross terms are summed apart from
// the large ones.  Measured against fp64 (tools/microbench/gemm_f16x2_wide.hip): 0.6-0.9e-7 * sum|a w|, a third of the
// six-product split-bf16 kernels' (gemm_bf16x3.hip, kept behind DVQ_GEMM=bf16x3), at half their matrix work.
// Range: |a| must stay below 65 520 (fp16); beyond it the row's outputs are NaN, never a silently wrong number
// (GenNet.gen re-runs such a batch on the split-bf16 kernels, which have fp32's range).
//
// Shape: the 16x16x32 instruction holds a higher clock than 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS
// give-back item 7): measured here 288 against 262 TFLOP/s on the gated PixelCNN shape (M = 16 384, N = 1024, K = 1536).
//
// Tiled kernel: 128 x 256 output tile, eight waves as 2 (M) x 4 (N) of 64 x 64, one workgroup per CU, K-tiles of 32, two LDS
// stages of 48 KiB: both operands as fp16 planes with 64-byte rows, the 16-byte chunk c of row r stored at c ^ ((4 - (r >> 2)) & 3)
// (conflict-free ds_read_b128 in the 16-row fragment shape: the four lane groups of a read hit sixteen different 16-byte bank
// groups).  Activations go global -> registers -> split -> LDS (once per element), weight planes by LDS-DMA with the swizzle on
// the source address.  Weights are MFMA operand A: a lane owns ONE output row m and four consecutive columns per block, so
// every store is 16 bytes and a wave-instruction covers 64-byte row segments.
// The accumulation order of an output element -- k-tiles of 32 in source order; per tile hi: (a1 w1), lo: (a1 w2) then
// (a2 w1) -- is the same in the skinny kernel below: results do not depend on the batch tiling (batched == loop, bitwise).
#include "dvq_internal.h"
#include "gemm_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TM = 128, TN = 256, BK = 32;
constexpr int A_PL = TM * 64;                               // one activation plane [128][32] fp16
constexpr int W_PL = TN * 64;                               // one weight plane [256][32] fp16
constexpr int STAGE = 2 * A_PL + 2 * W_PL;                  // 49 152 B
constexpr size_t SMEM = 2 * STAGE;                          // 98 304 B: one workgroup per CU (eight waves, <= 256 registers)
constexpr float LO_SCALE = 1.0f / 2048.0f;

__device__ __forceinline__ int swz16(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// a -> (fp16(a), fp16((a - fp16(a)) * 2^11)) for eight values.  The second piece is ONE rounding of the exact value
// fma(h, -2048, a * 2048) (v_fma_mixlo / mixhi_f16 take the fp16 piece h as an operand and round the fp32 result to fp16):
// 20 vector instructions per eight values where convert-back, subtract, scale, convert took 32; the same bits (1 M random pairs incl.
// subnormal, out-of-range and non-finite values: tools/microbench, round 4).
__device__ __forceinline__ void split2(const f32x4& lo, const f32x4& hi, h8& p1, h8& p2) {
    unsigned hb[4], lb[4];
    const float m2048 = -2048.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = j < 2 ? lo[2 * j] : hi[2 * j - 4], a1 = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
        hb[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a0, a1}, h2));         // v_cvt_pk_f16_f32, round to nearest even
        const float t0 = a0 * 2048.0f, t1 = a1 * 2048.0f;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lb[j]) : "v"(hb[j]), "v"(m2048), "v"(t0));
        asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb[j]) : "v"(hb[j]), "v"(m2048), "v"(t1));
    }
    p1 = __builtin_bit_cast(h8, uint4{hb[0], hb[1], hb[2], hb[3]});
    p2 = __builtin_bit_cast(h8, uint4{lb[0], lb[1], lb[2], lb[3]});
}

struct TileCursor {
    int s, k_left;
    const float* a_ptr;
    const uint16_t* w_ptr[4];

    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m0, int n0, int tid, int wave, int lane) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        {
            long m = m0 + (tid >> 2);
            if (m >= p.M) m = p.M - 1;                     // clamped rows / columns only feed outputs the epilogue masks
            if (src.arow) m = src.arow[m];
            a_ptr = src.A + m * src.lda + 8 * (tid & 3);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // 32 pieces of 16 rows x 64 B (two planes x 256 rows), four per wave
            const int id = wave * 4 + i;
            const int pl = id >> 4, rb = id & 15;
            const int row = rb * 16 + (lane >> 2);
            int n = n0 + row;
            if (n >= p.N) n = p.N - 1;
            w_ptr[i] = src.Wp + pl * src.wp_plane + (long)n * src.ldw + 8 * ((lane & 3) ^ swz16(row));
        }
    }
    __device__ __forceinline__ bool valid() const { return k_left > 0; }
    __device__ __forceinline__ void issue_w(char* stage, int wave) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = wave * 4 + i;
            const int pl = id >> 4, rb = id & 15;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                             (__attribute__((address_space(3))) void*)(stage + 2 * A_PL + pl * W_PL + rb * 1024), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void advance(const GemmParams& p, long m0, int n0, int tid, int wave, int lane) {
        a_ptr += BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) w_ptr[i] += BK;
        k_left -= BK;
        if (k_left <= 0) open(p, s + 1, m0, n0, tid, wave, lane);
    }
};

// Epilogue of a wave's 64 x 64 block held as 4 x 4 blocks of 16 x 16: hi / lo [jn][i][e] = column n0w + 16 jn + 4 (lane >> 4) + e of
// row m0w + 16 i + (lane & 15).  The arithmetic per element -- (hi + lo * 2^-11) * scale, + bias, ... in this order -- is shared
// with the skinny kernel.
__device__ __forceinline__ f32x4 f16x2_combine(const f32x4& hi, const f32x4& lo, const float* ws, int n, bool vec) {
    f32x4 v = hi + lo * LO_SCALE;
    if (vec) v *= *reinterpret_cast<const f32x4*>(ws + n);
    return v;
}

// ReLU that keeps a NaN (torch.relu does; fmaxf would turn the NaN of an out-of-range row into 0 and hide it from gen()'s check)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }

template <int EPI>
__device__ __forceinline__ void f16x2_store_block(const GemmParams& p, long m, int n, f32x4 v, bool full) {
    if (full) {
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
        if constexpr (EPI == EPI_RESID) v += *reinterpret_cast<const f32x4*>(p.resid + m * p.ldr + n);
        if (p.relu) { v[0] = relu_nan(v[0]); v[1] = relu_nan(v[1]); v[2] = relu_nan(v[2]); v[3] = relu_nan(v[3]); }
        *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = v;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (n + c >= p.N) continue;
            float x = v[c] * p.wscale[n + c] + (p.bias ? p.bias[n + c] : 0.f);
            if constexpr (EPI == EPI_RESID) x += p.resid[m * p.ldr + n + c];
            if (p.relu) x = relu_nan(x);
            p.out[m * p.ldo + n + c] = x;
        }
    }
}

// tanh(a) * sigmoid(g) on the hardware's exp2 / reciprocal (1 ulp each): ten instructions per output where tanhf, expf and an IEEE
// division took ~85 -- the gate epilogue was 8 us of an 86 us tile.  Absolute error <= ~1.5e-7 (both factors are bounded by 1: the
// cancellation in 1 - 2 / (1 + e^2a) near a = 0 costs RELATIVE accuracy there, nothing in absolute terms); saturates to the exact
// limits, a NaN stays a NaN.  Every f16x2 kernel gates through this function (tiled == skinny, bitwise).
__device__ __forceinline__ float gate_act(float a, float g) {
    const float th = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * a)), 1.0f);
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * g));
    return th * sg;
}

// gate: `a` = the four tanh channels at gate-packed index na, `g` = their sigmoid partners at na + 32; natural channel c
__device__ __forceinline__ void f16x2_store_gate(const GemmParams& p, long m, int na, int c, f32x4 a, f32x4 g, const float* crow) {
    const int nb = na + 32;
    if (p.bias) {
        a += *reinterpret_cast<const f32x4*>(p.bias + na);
        g += *reinterpret_cast<const f32x4*>(p.bias + nb);
    }
    if (p.pre) {
        *reinterpret_cast<f32x4*>(p.pre + m * p.ldpre + na) = a;
        *reinterpret_cast<f32x4*>(p.pre + m * p.ldpre + nb) = g;
    }
    if (crow) {
        a += *reinterpret_cast<const f32x4*>(crow + na);
        g += *reinterpret_cast<const f32x4*>(crow + nb);
    }
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = gate_act(a[q], g[q]);
    *reinterpret_cast<f32x4*>(p.out + m * p.ldo + c) = o;
}

// accumulators of a wave's 64 x (16 NJ) block from a stored state (GemmParams::acc_*), or zero
template <int NJ, bool INIT>
__device__ __forceinline__ void f16x2_init_acc(const GemmParams& p, f32x4 (&hi)[NJ][4], f32x4 (&lo)[NJ][4], long m0w, int n0w, int lane) {
    if constexpr (!INIT) {                                  // the plain launch: exactly the code it always was
#pragma unroll
        for (int a = 0; a < NJ; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) { hi[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[a][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        return;
    }
    const int lr = lane & 15, lc = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0w + i * 16 + lr;
        const bool row_ok = p.acc_hi && m < p.M;
        const long r = row_ok ? (p.acc_row ? p.acc_row[m] : m) : 0;
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn) {
            const int n = n0w + jn * 16 + 4 * lc;
            if (row_ok && n < p.N) {                        // N % 4 == 0 with a state (checked by the launcher)
                hi[jn][i] = *reinterpret_cast<const f32x4*>(p.acc_hi + r * p.ldacc + n);
                lo[jn][i] = *reinterpret_cast<const f32x4*>(p.acc_lo + r * p.ldacc + n);
            } else {
                hi[jn][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                lo[jn][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

template <int EPI, int NJ = 4>
__device__ __forceinline__ void f16x2_epilogue(const GemmParams& p, f32x4 (&hi)[NJ][4], f32x4 (&lo)[NJ][4], long m0w, int n0w, int c0w,
                                               int lane) {
    static_assert(NJ == 4 || EPI != EPI_GATE, "the gate pairs a wave's tanh columns with their partners 32 columns further");
    const int lr = lane & 15, lc = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0w + i * 16 + lr;
        if (m >= p.M) continue;
        if constexpr (EPI == EPI_BIAS || EPI == EPI_RESID) {
            const bool vec_ok = (p.N % 4 == 0) && (p.ldo % 4 == 0) && (EPI != EPI_RESID || p.ldr % 4 == 0);
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                const int n = n0w + jn * 16 + 4 * lc;
                if (n >= p.N) continue;
                const bool full = vec_ok && n + 3 < p.N;
                f16x2_store_block<EPI>(p, m, n, f16x2_combine(hi[jn][i], lo[jn][i], p.wscale, n, full), full);
            }
        } else if constexpr (EPI == EPI_STATE) {
#pragma unroll
            for (int jn = 0; jn < NJ; ++jn) {
                const int n = n0w + jn * 16 + 4 * lc;
                if (n >= p.N) continue;
                *reinterpret_cast<f32x4*>(p.out + m * p.ldo + n) = hi[jn][i];
                *reinterpret_cast<f32x4*>(p.pre + m * p.ldpre + n) = lo[jn][i];
            }
        } else if constexpr (EPI == EPI_GATE) {
            if constexpr (NJ == 4) {
            if (n0w >= p.N) continue;                      // N % 128 == 0: a wave's 64 columns are all inside or all outside
            const float* crow = p.cls ? p.cls + (long)p.label[m] * p.N : nullptr;
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                const int nl = jn * 16 + 4 * lc;
                const int na = n0w + nl;
                f16x2_store_gate(p, m, na, c0w + nl, f16x2_combine(hi[jn][i], lo[jn][i], p.wscale, na, true),
                                 f16x2_combine(hi[jn + 2][i], lo[jn + 2][i], p.wscale, na + 32, true), crow);
            }
            }
        }
    }
}

template <int EPI, bool DEPHASE, bool INIT>
__global__ __launch_bounds__(512, 1) void gemm_f16x2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + TN - 1) / TN;
    const long tiles_m = (p.M + TM - 1) / TM;
    const long b = blockIdx.x;
    const long j = b >> 3;                                  // XCD-aware map: the column tiles of a 128-row panel run on one XCD
    const long mt = (j / tiles_n) * 8 + (b & 7);
    const int nt = (int)(j % tiles_n);
    if (mt >= tiles_m) return;
    const long m0 = mt * TM;
    const int n0 = nt * TN;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    f32x4 hi[4][4], lo[4][4];                               // [jn][i]
    f16x2_init_acc<4, INIT>(p, hi, lo, m0 + wm * 64, n0 + wn * 64, lane);

    TileCursor cur;
    cur.open(p, 0, m0, n0, tid, wave, lane);
    const int a_dst = (tid >> 2) * 64 + 16 * ((tid & 3) ^ swz16(tid >> 2));
    f32x4 alo, ahi;
    auto load_a = [&]() {
        alo = *reinterpret_cast<const f32x4*>(cur.a_ptr);
        ahi = *reinterpret_cast<const f32x4*>(cur.a_ptr + 4);
    };
    auto store_a = [&](char* stage) {
        h8 p1, p2;
        split2(alo, ahi, p1, p2);
        *reinterpret_cast<h8*>(stage + a_dst) = p1;
        *reinterpret_cast<h8*>(stage + A_PL + a_dst) = p2;
    };
    load_a();
    cur.issue_w(smem_c, wave);
    cur.advance(p, m0, n0, tid, wave, lane);
    store_a(smem_c);
    bool more = cur.valid();
    if (more) load_a();                                     // tile 1's activations: written into the other stage during tile 0

    const int rd = (lane & 15) * 64 + 16 * ((lane >> 4) ^ swz16(lane & 15));       // fragment read: row lane & 15, chunk lane >> 4
    int stage = 0;
    while (true) {
        dvq_dma_barrier();                                  // this stage is complete (weight DMA landed, activation planes written);
                                                            // everybody is done reading the other stage
        const char* st = smem_c + stage * STAGE;
        char* nx = smem_c + (stage ^ 1) * STAGE;
        // The two waves of a SIMD (w and w + 4) move in lock step: waves 0..3 feed the next stage at the top of the tile, waves 4..7
        // half way, so that one of the pair runs MFMAs while the other sits in the DMA issue.
        auto feed = [&]() {
            store_a(nx);                                    // loaded a K-tile ago; before the DMA issue (vmcnt counts in order)
            cur.issue_w(nx, wave);
            cur.advance(p, m0, n0, tid, wave, lane);
            if (cur.valid()) load_a();
        };
        if (more && (wave < 4 || !DEPHASE)) feed();
        h8 af[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = *reinterpret_cast<const h8*>(st + pl * A_PL + (wm * 64 + i * 16) * 64 + rd);
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
        if (!more) break;
        more = cur.valid();
        stage ^= 1;
    }
    f16x2_epilogue<EPI>(p, hi, lo, m0 + wm * 64, n0 + wn * 64, (n0 >> 1) + wn * 32, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Ping-pong schedule of the same tile (the default): the two waves of a SIMD -- w and w + 4, rows 0..63 and 64..127 of the tile --
// alternate between a LOAD phase (the sixteen fragment reads of K-tile t; conversion + LDS write of this thread's share of tile
// t + 2, whose activations it loaded a period ago; LDS-DMA of tile t + 2's weight planes; global loads of tile t + 3's activations)
// and a COMPUTE phase (the 48 MFMAs of tile t on the fragments in registers, at s_setprio 1), one workgroup barrier per phase,
// waves 4..7 half a period behind: one wave of every SIMD always has MFMAs to issue.  Three LDS stages (144 KiB): what a load
// phase issues has two phases to land, and the stage it writes was last read two phases earlier.  Same operands into the same
// MFMA sequence per output as gemm_f16x2_kernel: bit-identical.  Measured on the gated shape (M 16 384, N 1024, K 1536):
// 300-307 TFLOP/s against 286-293 for the two-stage dephased kernel; in-kernel stamps (tools/microbench/gemm_f16x2_wide.hip):
// load phase ~850 cycles + loop overhead against 803 for the compute phase at an in-kernel clock of 1.74 GHz; moving the split
// into the compute phase's MFMA gaps (1 MFMA : 2 vector instructions) measured SLOWER (273).  Hand-counted waits (activation
// loads as inline asm, vmcnt(4) / vmcnt(6) so that neither stream waits for the other's younger requests) measured 327 on the
// microbench, where the activations stream from HBM, and NOTHING in the benchmark step (gated GEMMs 199.6 -> 200.5 ms: their
// activations were written by the previous launch); the variant tried also let waves 0-3 read a tile whose DMA pieces waves 4-7
// had not waited for yet (one differing result hash in four bench runs) -- not kept: every load phase starts with vmcnt(0).
constexpr int PP_STAGES = 3;
constexpr size_t PP_SMEM = PP_STAGES * STAGE;               // 147 456 B
constexpr size_t PP_SMEM_N128 = PP_STAGES * (2 * A_PL + 2 * 128 * 64);   // 98 304 B (128 x 128 tile)

struct PpCursorA {
    int s, k_left;
    const float* a_ptr;
    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m0, int tid) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        long m = m0 + (tid >> 2);
        if (m >= p.M) m = p.M - 1;                         // clamped rows / columns only feed outputs the epilogue masks
        if (src.arow) m = src.arow[m];
        a_ptr = src.A + m * src.lda + 8 * (tid & 3);
    }
    __device__ __forceinline__ bool valid() const { return k_left > 0; }
    __device__ __forceinline__ void advance(const GemmParams& p, long m0, int tid) {
        a_ptr += BK;
        k_left -= BK;
        if (k_left <= 0) open(p, s + 1, m0, tid);
    }
};

// NJ = 16-column blocks per wave: 4 -> the 128 x 256 tile, 2 -> a 128 x 128 tile (launches whose 128 x 256 tiles would leave half
// the CUs idle: the short-N GEMMs of an 8 192-row batch -- one rank's share of the benchmark batch on eight GPUs)
template <int NJ>
struct PpCursorW {
    int s, k_left;
    const uint16_t* w_ptr[NJ];
    __device__ __forceinline__ void open(const GemmParams& p, int src_i, int n0, int wave, int lane) {
        s = src_i;
        if (s >= p.nsrc) { k_left = 0; return; }
        const GemmSrc& src = p.src[s];
        k_left = src.K;
#pragma unroll
        for (int i = 0; i < NJ; ++i) {                     // 8 NJ pieces of 16 rows x 64 B (two planes x 64 NJ rows), NJ per wave
            const int id = wave * NJ + i;
            const int pl = id / (4 * NJ), rb = id % (4 * NJ);
            const int row = rb * 16 + (lane >> 2);
            int n = n0 + row;
            if (n >= p.N) n = p.N - 1;
            w_ptr[i] = src.Wp + pl * src.wp_plane + (long)n * src.ldw + 8 * ((lane & 3) ^ swz16(row));
        }
    }
    __device__ __forceinline__ void issue(char* stage, int wave) const {
#pragma unroll
        for (int i = 0; i < NJ; ++i) {
            const int id = wave * NJ + i;
            const int pl = id / (4 * NJ), rb = id % (4 * NJ);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)w_ptr[i],
                                             (__attribute__((address_space(3))) void*)(stage + 2 * A_PL + pl * (64 * NJ * 64) + rb * 1024), 16, 0, 0);
        }
    }
    __device__ __forceinline__ void advance(const GemmParams& p, int n0, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < NJ; ++i) w_ptr[i] += BK;
        k_left -= BK;
        if (k_left <= 0) open(p, s + 1, n0, wave, lane);
    }
};

template <int EPI, int NJ, bool INIT>
__global__ __launch_bounds__(512, 1) void gemm_f16x2_pp_kernel(const GemmParams p, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    constexpr int TN = 64 * NJ, W_PL = TN * 64, STAGE = 2 * A_PL + 2 * W_PL;      // this kernel's tile width (shadows the file's)
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + TN - 1) / TN;
    const long tiles_m = (p.M + TM - 1) / TM;
    const long b = blockIdx.x;
    const long j = b >> 3;                                  // XCD-aware map: the column tiles of a 128-row panel run on one XCD
    const long mt = (j / tiles_n) * 8 + (b & 7);
    const int nt = (int)(j % tiles_n);
    if (mt >= tiles_m) return;
    const long m0 = mt * TM;
    const int n0 = nt * TN;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    PpCursorA ca;
    PpCursorW<NJ> cw;
    ca.open(p, 0, m0, tid);
    cw.open(p, 0, n0, wave, lane);
    const int a_dst = (tid >> 2) * 64 + 16 * ((tid & 3) ^ swz16(tid >> 2));
    f32x4 alo, ahi;
    auto load_a = [&]() {
        alo = *reinterpret_cast<const f32x4*>(ca.a_ptr);
        ahi = *reinterpret_cast<const f32x4*>(ca.a_ptr + 4);
        ca.advance(p, m0, tid);
    };
    auto store_a = [&](char* stage) {
        h8 p1, p2;
        split2(alo, ahi, p1, p2);
        *reinterpret_cast<h8*>(stage + a_dst) = p1;
        *reinterpret_cast<h8*>(stage + A_PL + a_dst) = p2;
    };
    // prologue: tiles 0 and 1 staged, tile 2's activations in registers
    load_a();
    cw.issue(smem_c, wave);
    cw.advance(p, n0, wave, lane);
    store_a(smem_c);
    if (T > 1) {
        load_a();
        cw.issue(smem_c + STAGE, wave);
        cw.advance(p, n0, wave, lane);
        store_a(smem_c + STAGE);
    }
    if (T > 2) load_a();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    f32x4 hi[NJ][4], lo[NJ][4];                             // [jn][i]
    f16x2_init_acc<NJ, INIT>(p, hi, lo, m0 + wm * 64, n0 + wn * 16 * NJ, lane);

    const int rd = (lane & 15) * 64 + 16 * ((lane >> 4) ^ swz16(lane & 15));       // fragment read: row lane & 15, chunk lane >> 4
    if (wave >= 4) __builtin_amdgcn_s_barrier();           // the second half runs half a period behind
    int cur_st = 0, nx_st = 2;                              // stage of tile t, of tile t + 2
    for (int t = 0; t < T; ++t) {
        // ---- load phase
        const char* st = smem_c + cur_st * STAGE;
        h8 af[4][2], wf[NJ][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (DVQ_DIAG_ON && (p.dbg_abl & 128)) {     // timing only: no fragment reads at all
                    af[i][pl] = h8{};
                    if (i < NJ) wf[i][pl] = h8{};
                    continue;
                }
                if (DVQ_DIAG_ON && (p.dbg_abl & 64) && pl == 1) {   // timing only: half the fragment bytes (the second pieces = the first)
                    af[i][1] = af[i][0];
                    if (i < NJ) wf[i][1] = wf[i][0];
                    continue;
                }
                af[i][pl] = *reinterpret_cast<const h8*>(st + pl * A_PL + (wm * 64 + i * 16) * 64 + rd);
                if (i < NJ) wf[i][pl] = *reinterpret_cast<const h8*>(st + 2 * A_PL + pl * W_PL + (wn * 16 * NJ + i * 16) * 64 + rd);
            }
        // Everything this wave issued in its previous load phase (a period ago) has landed before it passes this phase's barrier:
        // the DMA pieces of tile t + 1 -- read by everybody from the next load phase on -- and the activations converted below.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t + 2 < T) {
            char* nx = smem_c + nx_st * STAGE;
            // (timing-only ablations of the diagnostics build, DVQ_GEMM_ABL: 2 no MFMAs, 8 no weight DMA, 16 no split + plane store, 32 no
            // activation loads, 64 half the fragment reads, 128 none, 256 no epilogue, 512 one barrier per K-tile; results are garbage;
            // tools/gemm_skeleton.py turns them into nanoseconds per K-tile)
            if (!(DVQ_DIAG_ON && (p.dbg_abl & 16))) store_a(nx);   // tile t + 2
            if (!(DVQ_DIAG_ON && (p.dbg_abl & 8))) cw.issue(nx, wave);
            cw.advance(p, n0, wave, lane);
            if (t + 3 < T && !(DVQ_DIAG_ON && (p.dbg_abl & 32))) load_a();   // tile t + 3
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers (and the plane writes retired) before the phase ends
        __builtin_amdgcn_s_barrier();
        // ---- compute phase: hi: (a1 w1); lo: (a1 w2) then (a2 w1) -- per output the order of gemm_f16x2_kernel
        __builtin_amdgcn_s_setprio(1);
        if (DVQ_DIAG_ON && (p.dbg_abl & 2)) {               // timing only: no matrix work (the operands still have to arrive)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : : "v"(af[i][0]), "v"(af[i][1]), "v"(wf[i < NJ ? i : 0][0]), "v"(wf[i < NJ ? i : 0][1]));
        } else {
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int i = 0; i < 4; ++i) hi[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jn][0], af[i][0], hi[jn][i], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int i = 0; i < 4; ++i) lo[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jn][1], af[i][0], lo[jn][i], 0, 0, 0);
#pragma unroll
        for (int jn = 0; jn < NJ; ++jn)
#pragma unroll
            for (int i = 0; i < 4; ++i) lo[jn][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[jn][0], af[i][1], lo[jn][i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (!(DVQ_DIAG_ON && (p.dbg_abl & 512))) __builtin_amdgcn_s_barrier();   // (512: timing only, the loop with ONE barrier per K-tile)
        cur_st = cur_st == PP_STAGES - 1 ? 0 : cur_st + 1;
        nx_st = nx_st == PP_STAGES - 1 ? 0 : nx_st + 1;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();            // pairs with the late half's extra barrier
    if (DVQ_DIAG_ON && (p.dbg_abl & 256)) {                // timing only: no epilogue (one store so that the accumulators stay alive)
        if (hi[0][0][0] + lo[0][0][0] == 12345.678f) p.out[0] = 1.f;
        return;
    }
    f16x2_epilogue<EPI, NJ>(p, hi, lo, m0 + wm * 64, n0 + wn * 16 * NJ, (n0 >> 1) + wn * 32, lane);
}

// Measured and not kept (round 4): a 128 x 128 tile with four waves and two 32 KiB stages, TWO workgroups per CU, for the short-K
// launches (the PixelCNN's 1x1 residual conv and head, K = 512: 46 us per launch = 187 TFLOP/s; 100 MB of HBM traffic for 8.6 GFLOP).
// Bit-identical, and no faster: residual GEMMs 22.6 -> 23.3 ms per step, gated GEMMs 192 -> 212 ms when forced on everything.  A
// workgroup of those launches moves ~1 MB (activations 256 KB, weight planes 512 KB from L2, residual in, tile out) in 46 us =
// 22 GB/s per CU, the rate one CU sustains from beyond L2 (MI355X_MICROARCH.md: 23-33 GB/s): they are bound by the per-CU memory
// path, not by the overlap of prologue and epilogue, and the smaller tile reads every activation row twice as often.
template <int EPI, bool INIT = false>
int launch_tiled(const GemmParams& p, hipStream_t stream) {
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            const hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_kernel<EPI, false, INIT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM);
            const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_kernel<EPI, true, INIT>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM);
            hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_pp_kernel<EPI, 4, INIT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_SMEM);
            if constexpr (EPI != EPI_GATE)
                if (e2 == hipSuccess) e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_pp_kernel<EPI, 2, INIT>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_SMEM_N128);
            return e0 != hipSuccess ? e0 : (e1 != hipSuccess ? e1 : e2);
        });
        if (e != hipSuccess) {
            dvq_set_error("gemm_f16x2: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const long tiles_m = (p.M + TM - 1) / TM;
    const long tiles_n = (p.N + TN - 1) / TN;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    // 128 x 128 tiles where they take fewer rounds of workgroups per unit of work (one workgroup per CU either way): e.g. M = 8 192,
    // N = 512 is 128 tiles of 128 x 256 -- half the chip -- or 256 of 128 x 128.  DVQ_GEMM_TN=128 / 256 forces one (A/B runs, same bits).
    bool narrow = false;
    if constexpr (EPI != EPI_GATE) {
        const long tiles_n2 = (p.N + 127) / 128, cus = dvq_num_cus();
        const long r256 = (tiles_m * tiles_n + cus - 1) / cus, r128 = (tiles_m * tiles_n2 + cus - 1) / cus;
        narrow = r128 < 2 * r256;
        if (dvq_knobs().gemm_tn == 128) narrow = true;
        if (dvq_knobs().gemm_tn == 256) narrow = false;
    }
    static const char* const names_row[] = {"gemm_bias", "gemm_resid", "gemm_gate", "", "", "gemm_state"};
    static const char* const names_cls[] = {"gemm_bias_cls", "gemm_resid_cls", "gemm_gate_cls", "", "", "gemm_state"};
    const char* const* names = p.prof_cls ? names_cls : names_row;
    double ksum = 0;
    int T = 0;
    for (int s = 0; s < p.nsrc; ++s) { ksum += p.src[s].K; T += p.src[s].K / BK; }
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        const int mode = dvq_knobs().gemm_dephase;          // DVQ_GEMM_DEPHASE: 2 (default) ping-pong, 1 two stages dephased, 0 two stages in lock step
        if (mode == 2 && narrow) {
            if constexpr (EPI != EPI_GATE) {
                const long grid2 = ((tiles_m + 7) / 8) * 8 * ((p.N + 127) / 128);
                DVQ_LAUNCH((gemm_f16x2_pp_kernel<EPI, 2, INIT>), dim3((unsigned)grid2), dim3(512), PP_SMEM_N128, stream, p, T);
            }
        } else if (mode == 2) DVQ_LAUNCH((gemm_f16x2_pp_kernel<EPI, 4, INIT>), dim3((unsigned)grid), dim3(512), PP_SMEM, stream, p, T);
        else if (mode == 1) DVQ_LAUNCH((gemm_f16x2_kernel<EPI, true, INIT>), dim3((unsigned)grid), dim3(512), SMEM, stream, p);
        else DVQ_LAUNCH((gemm_f16x2_kernel<EPI, false, INIT>), dim3((unsigned)grid), dim3(512), SMEM, stream, p);
    }
    DVQ_CHECK_LAUNCH("gemm_f16x2");
    return DVQ_OK;
}

// ================================================================================================================
// Skinny kernel for M <= 256 (the reference's own call pattern: GenNet.gen with B = 1 per call, 1 / 20 / 49 / 100 grasps per object,
// gen_diverse_grasp_ho3d.py:212-236): a B = 1 call is ~470 dependent GEMM launches with M = 1 row each, so what counts is the
// time of ONE small launch, not throughput.  Measured (gated launch, K = 3072, M = 1): 13.5 us whether a wave owns 16, 8 or 4
// columns (32 / 64 / 128 waves), with 24 or 48 KB of loads in flight, with or without 192 helper workgroups sweeping the rows
// into the XCD's L2: not the weight stream but the wave's own instruction stream, ~70 instructions per k-step of which 32 split
// the activations.  One single-wave workgroup per R output columns (R = 16, 8 or 4: the MFMA's sixteen weight rows are R real
// rows and repeats, whose outputs nobody reads) and per 16 rows of M; for the gate the block's rows are R tanh columns
// followed by their R sigmoid partners (R = 8 or 4), exchanged by a lane shuffle in the epilogue.  No LDS, no barrier: the
// wave streams its weight rows HBM -> registers in the MFMA fragment shape (lane = row l & 15, 16-byte chunk l >> 4 of a
// 64-byte k-step), its activation rows in whole 128-byte lines, PF k-steps ahead through a ring of register sets (no branch
// around a load: steps past the end re-load the last tile), splits the activations in registers and runs the tiled kernel's
// MFMA sequence per output -- hi: (a1 w1), lo: (a1 w2) then (a2 w1), k-steps in source order -- so the result is bit-identical
// to the batched path (an output's value does not depend on its position in the 16 x 16 block:
// tests/test_gpu_parity.py::test_skinny_gemm_equals_tiled_kernels_bitwise).
template <int NB>
struct SkinnySlot {
    uint4 w[NB][2];
    f32x4 a[2];
};

template <int NB>
struct SkinnyCursor16 {
    int s, k_left;
    const float* a_ptr;
    const uint16_t* w_ptr[NB][2];

    __device__ __forceinline__ void open(const GemmParams& p, int src_i, long m, const int (&n)[NB], int lc) {
        s = src_i;
        const GemmSrc& src = p.src[s];
        k_left = src.K;
        a_ptr = src.A + (src.arow ? src.arow[m] : m) * src.lda + 8 * lc;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) w_ptr[b][pl] = src.Wp + pl * src.wp_plane + (long)n[b] * src.ldw + 8 * lc;
    }
    __device__ __forceinline__ void load(SkinnySlot<NB>& t) const {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            t.w[b][0] = *reinterpret_cast<const uint4*>(w_ptr[b][0]);
            t.w[b][1] = *reinterpret_cast<const uint4*>(w_ptr[b][1]);
        }
        t.a[0] = *reinterpret_cast<const f32x4*>(a_ptr);
        t.a[1] = *reinterpret_cast<const f32x4*>(a_ptr + 4);
    }
    // step to the next k-step of the launch, if there is one (wave-uniform; no load inside)
    __device__ __forceinline__ void advance(const GemmParams& p, long m, const int (&n)[NB], int lc, bool last) {
        if (last) return;
        k_left -= BK;
        if (k_left > 0) {
            a_ptr += BK;
#pragma unroll
            for (int b = 0; b < NB; ++b) { w_ptr[b][0] += BK; w_ptr[b][1] += BK; }
        } else open(p, s + 1, m, n, lc);
    }
};

// block row j (0..15) of column group g -> output column (gate-packed index for the gate); rows past the real ones repeat row 0
template <int EPI, int R>
__device__ __forceinline__ int skinny_col(int g, int j) {
    if constexpr (EPI == EPI_GATE) {
        constexpr int per64 = 32 / R;                       // groups per 64-column gate block (32 tanh columns + 32 partners)
        const int c0 = 64 * (g / per64) + R * (g % per64);
        return j < R ? c0 + j : (j < 2 * R ? c0 + 32 + (j - R) : c0);
    } else {
        return j < R ? R * g + j : R * g;
    }
}

// NB column groups per wave (2 when the launch has several row groups: they share the activation split, the larger part of a
// k-step's instructions)
template <int EPI, int R, int NB, int PF>
__device__ __forceinline__ void skinny_body(const GemmParams& p, int T, int groups) {
    const int lane = threadIdx.x;
    const int lr = lane & 15, lc = lane >> 4;
    const long mb0 = (long)blockIdx.y * 16;
    long m = mb0 + lr;
    if (m >= p.M) m = p.M - 1;                              // clamped rows / columns only feed masked outputs
    int g[NB], n[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        g[b] = (int)blockIdx.x * NB + b;
        if (g[b] >= groups) g[b] = groups - 1;              // an odd group count: the last wave computes its group twice
        n[b] = skinny_col<EPI, R>(g[b], lr);
        if (n[b] >= p.N) n[b] = p.N - 1;
    }
    f32x4 hi[NB], lo[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {                          // zero, or the stored state of this lane's four columns of row mb0 + lr
        hi[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        lo[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nc0 = skinny_col<EPI, R>(g[b], 4 * lc);
        const bool col_live = EPI == EPI_GATE ? 4 * lc < 2 * R : (4 * lc < R && nc0 < p.N);
        if (p.acc_hi && col_live && mb0 + lr < p.M) {
            const long r = p.acc_row ? p.acc_row[mb0 + lr] : mb0 + lr;
            hi[b] = *reinterpret_cast<const f32x4*>(p.acc_hi + r * p.ldacc + nc0);
            lo[b] = *reinterpret_cast<const f32x4*>(p.acc_lo + r * p.ldacc + nc0);
        }
    }
    SkinnyCursor16<NB> cur;
    cur.open(p, 0, m, n, lc);
    SkinnySlot<NB> slot[PF];
    int issued = 0;                                         // index of the k-step the cursor points at (the last one repeats)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        cur.load(slot[u]);
        cur.advance(p, m, n, lc, issued + 1 >= T);
        issued = issued + 1 < T ? issued + 1 : T;
    }
    for (int t = 0; t < T; t += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (t + u < T) {
                h8 a1, a2;
                split2(slot[u].a[0], slot[u].a[1], a1, a2);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const h8 w1 = __builtin_bit_cast(h8, slot[u].w[b][0]), w2 = __builtin_bit_cast(h8, slot[u].w[b][1]);
                    hi[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, a1, hi[b], 0, 0, 0);
                    lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, a1, lo[b], 0, 0, 0);
                    lo[b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, a2, lo[b], 0, 0, 0);
                }
            }
            cur.load(slot[u]);
            cur.advance(p, m, n, lc, issued + 1 >= T);
            issued = issued + 1 < T ? issued + 1 : T;
        }
    }
    // epilogue: the tiled kernel's arithmetic per element.  The lane holds block rows 4 lc .. 4 lc + 3 (four consecutive columns)
    // of output row mb0 + lr.
    const long mo = mb0 + lr;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (b > 0 && (int)blockIdx.x * NB + b >= groups) break;
        const int nc = skinny_col<EPI, R>(g[b], 4 * lc);    // first of the lane's four columns
        if constexpr (EPI == EPI_GATE) {
            const bool live = 4 * lc < 2 * R && mo < p.M;   // N % 128 == 0: every group is whole
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (live) {
                v = f16x2_combine(hi[b], lo[b], p.wscale, nc, true);
                if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + nc);
                if (p.pre) *reinterpret_cast<f32x4*>(p.pre + mo * p.ldpre + nc) = v;
                if (p.cls) v += *reinterpret_cast<const f32x4*>(p.cls + (long)p.label[mo] * p.N + nc);
            }
            f32x4 gg;                                       // the sigmoid partners sit R / 4 lane groups further
#pragma unroll
            for (int q = 0; q < 4; ++q) gg[q] = __shfl(v[q], lane + 4 * R, 64);
            if (live && 4 * lc < R) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = gate_act(v[q], gg[q]);
                const int c = 32 * (nc >> 6) + (nc & 31);   // natural channel of gate-packed tanh column nc
                *reinterpret_cast<f32x4*>(p.out + mo * p.ldo + c) = o;
            }
        } else if constexpr (EPI == EPI_STATE) {
            if (4 * lc >= R || mo >= p.M || nc >= p.N) continue;
            *reinterpret_cast<f32x4*>(p.out + mo * p.ldo + nc) = hi[b];
            *reinterpret_cast<f32x4*>(p.pre + mo * p.ldpre + nc) = lo[b];
        } else {
            if (4 * lc >= R || mo >= p.M || nc >= p.N) continue;
            const bool vec_ok = (p.N % 4 == 0) && (p.ldo % 4 == 0) && (EPI != EPI_RESID || p.ldr % 4 == 0);
            const bool full = vec_ok && nc + 3 < p.N;
            f16x2_store_block<EPI>(p, mo, nc, f16x2_combine(hi[b], lo[b], p.wscale, nc, full), full);
        }
    }
}

template <int EPI, int R, int NB, int PF>
__global__ __launch_bounds__(64) void gemm_f16x2_skinny_kernel(const GemmParams p, int T, int groups) {
    skinny_body<EPI, R, NB, PF>(p, T, groups);
}

// Up to three gated GEMMs of one shape that do not depend on each other -- the three columns of a row of the PixelCNN's vertical
// stack (pixelcnn.hip) -- in ONE launch, problem = blockIdx.z: a B = 1 call is a chain of dependent launches of ~9 us each, and
// this takes 90 of its ~470 GEMM launches away.  Every output is computed by the instruction sequence of the single launch.
struct GemmGroup {
    GemmParams p[DVQ_GEMM_GROUP_MAX];
    int T[DVQ_GEMM_GROUP_MAX];
};
template <int R, int NB, int PF>
__global__ __launch_bounds__(64) void gemm_f16x2_skinny_gate_group_kernel(const GemmGroup g, int groups) {
    skinny_body<EPI_GATE, R, NB, PF>(g.p[blockIdx.z], g.T[blockIdx.z], groups);
}

constexpr long SKINNY_MAX_M = 256;     // above this the tiled kernel wins (every row group re-reads the weight panel from L2)
// launches on a class table's rows (pixelcnn.hip: a chain of ~150 dependent launches per call on n_classes rows) are latency-bound
// up to more rows: 512 rows measured 7.0 -> 4.6 ms per call
inline long skinny_max_m(const GemmParams& p) { return p.prof_cls ? 4 * SKINNY_MAX_M : SKINNY_MAX_M; }

template <int EPI, int R>
void launch_skinny_r(const GemmParams& p, int T, int gy, hipStream_t stream) {
    const int groups = EPI == EPI_GATE ? (p.N / 64) * (32 / R) : (p.N + R - 1) / R;
    if (gy >= 3) DVQ_LAUNCH((gemm_f16x2_skinny_kernel<EPI, R, 2, 4>), dim3((unsigned)((groups + 1) / 2), (unsigned)gy), dim3(64), 0, stream, p, T, groups);
    else DVQ_LAUNCH((gemm_f16x2_skinny_kernel<EPI, R, 1, 8>), dim3((unsigned)groups, (unsigned)gy), dim3(64), 0, stream, p, T, groups);
}

template <int EPI>
int launch_skinny(const GemmParams& p, hipStream_t stream) {
    static const char* const names_row[] = {"gemm_bias", "gemm_resid", "gemm_gate", "", "", "gemm_state"};
    static const char* const names_cls[] = {"gemm_bias_cls", "gemm_resid_cls", "gemm_gate_cls", "", "", "gemm_state"};
    const char* const* names = p.prof_cls ? names_cls : names_row;
    double ksum = 0;
    int T = 0;
    for (int s = 0; s < p.nsrc; ++s) { ksum += p.src[s].K; T += p.src[s].K / BK; }
    const int gy = (int)((p.M + 15) / 16);
    // Columns per wave.  One row group (M <= 16): the launch waits for the single instruction stream of a wave (~70 instructions
    // per k-step), whatever its width: 8 + 8 gate columns / 16 plain columns per wave.  More row groups: the widest blocks, two
    // per wave (DVQ_GEMM_SKINNY_COLS = 16 / 8 / 4 forces a width: A/B runs, same bits).
    const int want = dvq_knobs().gemm_skinny_cols;
    int R = 16;
    if (want == 4 || want == 8 || want == 16) R = want;
    if (EPI == EPI_GATE && R == 16) R = 8;
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        if constexpr (EPI == EPI_GATE) {
            if (R == 4) launch_skinny_r<EPI, 4>(p, T, gy, stream); else launch_skinny_r<EPI, 8>(p, T, gy, stream);
        } else {
            if (R == 4) launch_skinny_r<EPI, 4>(p, T, gy, stream);
            else if (R == 8) launch_skinny_r<EPI, 8>(p, T, gy, stream);
            else launch_skinny_r<EPI, 16>(p, T, gy, stream);
        }
    }
    DVQ_CHECK_LAUNCH("gemm_f16x2_skinny");
    return DVQ_OK;
}

template <int R>
void launch_skinny_gate_group_r(const GemmGroup& g, int n, int gy, hipStream_t stream) {
    const int groups = (g.p[0].N / 64) * (32 / R);
    if (gy >= 3) DVQ_LAUNCH((gemm_f16x2_skinny_gate_group_kernel<R, 2, 4>), dim3((unsigned)((groups + 1) / 2), (unsigned)gy, (unsigned)n), dim3(64), 0, stream, g, groups);
    else DVQ_LAUNCH((gemm_f16x2_skinny_gate_group_kernel<R, 1, 8>), dim3((unsigned)groups, (unsigned)gy, (unsigned)n), dim3(64), 0, stream, g, groups);
}

int launch_skinny_gate_group(const GemmParams* ps, int n, hipStream_t stream) {
    GemmGroup g = {};
    double flops = 0, bytes = 0;
    for (int i = 0; i < n; ++i) {
        g.p[i] = ps[i];
        double ksum = 0;
        for (int s = 0; s < ps[i].nsrc; ++s) { ksum += ps[i].src[s].K; g.T[i] += ps[i].src[s].K / BK; }
        flops += 2.0 * (double)ps[i].M * ps[i].N * ksum;
        bytes += ((double)ps[i].M + ps[i].N) * ksum * 4;
    }
    const int gy = (int)((ps[0].M + 15) / 16);
    const int want = dvq_knobs().gemm_skinny_cols;          // as launch_skinny: the same width, hence the same kernel body per output
    {
        DVQ_PROF(ps[0].prof_cls ? "gemm_gate_cls" : "gemm_gate", flops, bytes, stream);
        if (want == 4) launch_skinny_gate_group_r<4>(g, n, gy, stream); else launch_skinny_gate_group_r<8>(g, n, gy, stream);
    }
    DVQ_CHECK_LAUNCH("gemm_f16x2_skinny_group");
    return DVQ_OK;
}

// ---------------------------------------------------------------------------------------------------------------- packer
// row_absmax[n] = max(row_absmax[n], max over o, k of |w[o][n][k]|)   (w: [outer][N][K] dense; one wave per (o, n) row)
__global__ void f16x2_absmax_kernel(const float* __restrict__ w, long rows, int N, int K, float* __restrict__ absmax) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* src = w + row * K;
    float mx = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float v = fabsf(src[k]);
        mx = (v > mx || v != v) ? v : mx;                   // a NaN weight poisons the row's scale (outputs NaN, as in fp32)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(mx, o);
        mx = (ov > mx || ov != ov) ? ov : mx;
    }
    if (lane == 0) {
        // non-negative floats order like their bit patterns; NaN patterns are larger than every finite one
        atomicMax(reinterpret_cast<unsigned*>(absmax + row % N), __float_as_uint(mx));
    }
}

// t_n from the row maximum: the largest power of two with max|w| * 2^t < 2^15  (rows of zeros, Inf or NaN: t = 0)
__device__ __forceinline__ int f16x2_exponent(float amax) {
    if (!(amax > 0.f) || !(amax < INFINITY)) return 0;
    int e;
    (void)frexpf(amax, &e);                                 // amax = f * 2^e, f in [0.5, 1)
    int t = 15 - e;
    if (t > 126) t = 126;                                   // 2^-t must stay a normal fp32 number
    if (t < -126) t = -126;
    return t;
}

__global__ void f16x2_split_kernel(const float* __restrict__ w, long total, int N, int K, const float* __restrict__ absmax,
                                   uint16_t* __restrict__ planes, float* __restrict__ row_scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int n = (int)((i / K) % N);
    const int t = f16x2_exponent(absmax[n]);
    const float ws = ldexpf(w[i], t);
    const _Float16 h1 = (_Float16)ws;
    const float r = ws - (float)h1;
    const _Float16 h2v = (_Float16)(r * 2048.0f);
    planes[i] = __builtin_bit_cast(uint16_t, h1);
    planes[total + i] = __builtin_bit_cast(uint16_t, h2v);
    if (i < N) row_scale[i] = ldexpf(1.0f, -f16x2_exponent(absmax[i]));
}

}  // namespace

// Called by dvq_launch_gemm (gemm_f32.hip) after argument validation, when every source carries f16x2 planes.
static int check_f16x2(const GemmParams& p, GemmEpilogue epi) {
    DVQ_REQUIRE(p.wscale, "gemm_f16x2: planes without row scales");
    for (int s = 0; s < p.nsrc; ++s) {
        const GemmSrc& g = p.src[s];
        DVQ_REQUIRE(g.Wp && dvq_aligned16(g.Wp) && g.wp_plane % 8 == 0 && g.ldw % 8 == 0, "gemm_f16x2: weight planes of source %d are not 16-byte aligned", s);
    }
    bool a_ok = true;                                       // 16-byte activation loads
    for (int s = 0; s < p.nsrc; ++s) a_ok = a_ok && dvq_aligned16(p.src[s].A) && p.src[s].lda % 4 == 0;
    DVQ_REQUIRE(a_ok, "gemm_f16x2: activation rows are not 16-byte aligned");
    if (epi == EPI_GATE)
        DVQ_REQUIRE(dvq_aligned16(p.out) && p.ldo % 4 == 0 && (!p.pre || (dvq_aligned16(p.pre) && p.ldpre % 4 == 0)), "gemm_f16x2: gate outputs are not 16-byte aligned");
    if (epi == EPI_STATE)
        DVQ_REQUIRE(p.out && p.pre && dvq_aligned16(p.out) && dvq_aligned16(p.pre) && p.ldo % 4 == 0 && p.ldpre % 4 == 0 && p.N % 4 == 0,
                    "gemm_f16x2: accumulator-state outputs need N %% 4 == 0 and 16-byte aligned rows");
    if (p.acc_hi)
        DVQ_REQUIRE(epi == EPI_GATE && p.acc_lo && dvq_aligned16(p.acc_hi) && dvq_aligned16(p.acc_lo) && p.ldacc % 4 == 0 && p.N % 4 == 0,
                    "gemm_f16x2: an accumulator state needs both sums, N %% 4 == 0 and 16-byte aligned rows");
    return DVQ_OK;
}

// Called by dvq_launch_gemm_gate_group (gemm_f32.hip) with 2 .. DVQ_GEMM_GROUP_MAX validated gated problems of one (M, N), all on
// fp16 planes.  Returns DVQ_OK after ONE launch, or a negative value when the group has to run as single launches (M beyond the
// skinny kernel, DVQ_GEMM_SKINNY=0).
int dvq_launch_gemm_f16x2_gate_group(const GemmParams* ps, int n, hipStream_t stream) {
    if (!dvq_knobs().gemm_skinny || ps[0].M > skinny_max_m(ps[0])) return -1;
    for (int i = 0; i < n; ++i) DVQ_PROPAGATE(check_f16x2(ps[i], EPI_GATE));
    return launch_skinny_gate_group(ps, n, stream);
}

int dvq_launch_gemm_f16x2(const GemmParams& p, GemmEpilogue epi, hipStream_t stream) {
    DVQ_PROPAGATE(check_f16x2(p, epi));
#ifdef DVQ_DIAG
    if (const char* e = getenv("DVQ_GEMM_ABL")) const_cast<GemmParams&>(p).dbg_abl = atoi(e);
#endif
    if (dvq_knobs().gemm_skinny && p.M <= skinny_max_m(p)) switch (epi) {
        case EPI_BIAS: return launch_skinny<EPI_BIAS>(p, stream);
        case EPI_RESID: return launch_skinny<EPI_RESID>(p, stream);
        case EPI_GATE: return launch_skinny<EPI_GATE>(p, stream);
        case EPI_STATE: return launch_skinny<EPI_STATE>(p, stream);
        default: break;
    }
    switch (epi) {
        case EPI_BIAS: return launch_tiled<EPI_BIAS>(p, stream);
        case EPI_RESID: return launch_tiled<EPI_RESID>(p, stream);
        case EPI_GATE: return p.acc_hi ? launch_tiled<EPI_GATE, true>(p, stream) : launch_tiled<EPI_GATE, false>(p, stream);
        case EPI_STATE: return launch_tiled<EPI_STATE>(p, stream);
        default: break;
    }
    dvq_set_error("gemm_f16x2: epilogue %d is not available on the fp16 split path", (int)epi);
    return DVQ_EINVAL;
}

extern "C" int dvq_f16x2_row_absmax(const float* w, int64_t outer, int N, int K, float* row_absmax, dvq_stream_t stream) {
    DVQ_REQUIRE(outer >= 1 && N >= 1 && K >= 1 && w && row_absmax, "f16x2_row_absmax: bad arguments");
    const long rows = (long)outer * N;
    DVQ_LAUNCH(f16x2_absmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, w, rows, N, K, row_absmax);
    DVQ_CHECK_LAUNCH("f16x2_row_absmax");
    return DVQ_OK;
}

extern "C" int dvq_split_f16x2(const float* w, int64_t outer, int N, int K, const float* row_absmax, uint16_t* planes, float* row_scale,
                               dvq_stream_t stream) {
    DVQ_REQUIRE(outer >= 1 && N >= 1 && K >= 1 && w && row_absmax && planes && row_scale, "split_f16x2: bad arguments");
    const long total = (long)outer * N * K;
    DVQ_LAUNCH(f16x2_split_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, total, N, K, row_absmax, planes,
               row_scale);
    DVQ_CHECK_LAUNCH("split_f16x2");
    return DVQ_OK;
}
