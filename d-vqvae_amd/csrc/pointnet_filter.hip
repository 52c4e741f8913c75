// PointNet trunk, filtered: conv1 -> conv2 -> [conv3 -> max over the points] with conv3 evaluated TWICE at very different
// cost (PointNetEncoder.forward, network/pointnet_encoder.py:147-164; STN3d.forward :30-35):
//
//   1. pn_trunk_filter_kernel: conv1 (vector ALU) and conv2 (split-bf16, six matrix-core products, fp32-accurate) as in
//      pn_trunk_kernel; the fp32 h2 rows [point][128] go to HBM.  conv3 (94 % of the trunk's FLOPs) is then evaluated as ONE
//      fp16 product per term on CENTRED rows: d_p = h2_p - c (c = pn_center_kernel's mean of four rows of the sample; the
//      argmax over the points does not depend on it), d scaled by a per-wave power of two, W3 by a per-channel power of
//      two, both rounded to fp16.  Every lane keeps, per 32 x 32 accumulator block (its 16 points of one channel), the TWO largest
//      approximate scores with the id of their point in the low mantissa bits; per (sample, 256-point tile, channel) the kernel
//      emits the FIVE largest of those 32 values and one flag per 16-point group that may hold further points in range whose
//      ids were not kept.
//   2. pn_exact_kernel: per (sample, channel) the estimate of a point's score is  approx + exact_dot(w_n, c)  with
//      |estimate - exact_dot(w_n, h2_p)| <= E_t = |r_n| max_p |d_p| + |w_n| max_p |rd_p| + C_ID |w_n| max_p |d_p|
//      + 2 DELTA |w_n| max_p |h2_p|, maxima over the tile; r_n = w_n - fp16 image (norm measured by the packer), rd_p =
//      d_p - fp16 image (norm measured by the trunk kernel), C_ID: id bits + matrix-core accumulation, DELTA: rounding of
//      one exact_dot.  Every tracked
//      point whose upper bound reaches the best lower bound is re-evaluated in fp32 (exact_dot: a fixed-order fp32 FMA dot
//      of W3[n,:] and the stored h2 row) and the maximum of THOSE values + bias is the feature -- bit-identical to the
//      maximum of exact_dot over ALL points (tests: DVQ_PN_EXHAUSTIVE=1 evaluates exactly that).  The 16 points of a flagged
//      group are all evaluated.
//
// Matrix-core cost per (32 points x 32 channels x K=128): 8 x v_mfma_f32_32x32x16_f16 instead of 48 bf16 MFMAs; the
// kernel is bound by vector-instruction issue: 2.5 instructions per score (id; max3 / med3 per group of three) since round 6 -- a top-two per 16-point
// group flags about as many points for re-evaluation as the top-three per 32 points of round 4 did.
#include "dvq_internal.h"
#include <vector>

namespace {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 qf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 qf16x2 __attribute__((ext_vector_type(2)));
typedef float qf32x2 __attribute__((ext_vector_type(2)));

// beyond the two measured rounding residuals, relative to |w| max|d|: the 8 id bits 2^-15 (3.05e-5); the matrix core's fp32
// accumulation of the 128 exact products -- at most 128 additions of 2^-23 each even if they truncated (1.53e-5; round to
// nearest: half of that); the product of the two residuals (2^-22); rounded up
constexpr float C_ID = 5.0e-5f;
// |exact_dot(w, h) - w.h| <= DELTA |w| |h|: 8 chained FMAs + 4 butterfly adds = 12 roundings (7.2e-7), + the fp32 add of
// the centre term
constexpr float DELTA = 1.0e-6f;
constexpr float NEG_BIG = -3.0e38f;
constexpr int MAX_TILES = 64;                             // filtered trunk: N <= 16384 points

constexpr int F_STAGE2 = 2 * 64 * 128;                    // conv2: one half of W2's two fp16 planes (2 x 64 rows x 128 B)
// the filter image of a trunk (dvq_pointnet_pack_filter): conv3 [1024][128] fp16 | 1 / scale [1024] | |w_n| [1024] | |w_n - image| [1024]
// | conv2 as two fp16 planes [2][128][64] of w * 2^t_n (the second: the remainder * 2^11) | 2^-t_n [128]
constexpr int IMG_OFF_TI = 1024 * 256, IMG_OFF_WN = IMG_OFF_TI + 4096, IMG_OFF_RN = IMG_OFF_WN + 4096;
constexpr int IMG_OFF_W2 = IMG_OFF_RN + 4096, IMG_OFF_K2 = IMG_OFF_W2 + 2 * 128 * 128, IMG_BYTES = IMG_OFF_K2 + 512;
constexpr int F_STAGE3 = 64 * 256;                        // conv3: 64 channels x 128 k fp16
constexpr int F_OFF_TB = 2 * F_STAGE3;                    // conv3 phase: record ring [4 chunks][4 waves][2 point blocks][2 halves][2][64] fp32 (32 KiB) behind the two W3 stages
constexpr int F_SLOT = 4 * 2 * 2 * 2 * 64;                // floats per chunk slot of the ring
constexpr int F_OFF_W1 = F_OFF_TB + 8 * 4 * 4 * 64 * 4;   // [64][4] fp32
constexpr int F_OFF_B1 = F_OFF_W1 + 64 * 4 * 4;           // [64]
constexpr int F_OFF_B2 = F_OFF_B1 + 64 * 4;               // [128]
constexpr int F_OFF_K2 = F_OFF_B2 + 128 * 4;              // [128] 2^-t_n of conv2's weight rows
constexpr int F_OFF_SC = F_OFF_K2 + 128 * 4;              // [4] 1 / (wave scale)
constexpr int F_OFF_TI = F_OFF_SC + 64;                   // [1024] 1 / (channel scale)
constexpr int F_OFF_CS = F_OFF_TI + 4096;                 // [128] centre of the sample
constexpr int F_OFF_WS = F_OFF_CS + 512;                  // [3][4] per-wave |h|max, |d|max, |rd|max
constexpr int F_OFF_E2 = F_OFF_WS + 64;                   // [1024] 2 E of the tile per channel ([4][1024] in the tail kernel: a tile per wave)
constexpr int F_LDS = F_OFF_E2 + 4096;                    // 76 672 B -> 2 workgroups per CU
constexpr int F_LDS_TAIL = F_LDS + 3 * 4096 + 4 * 512;    // tail kernel: three more e2 tables, one centre per wave (its own sample) behind them
static_assert(F_OFF_W1 >= 2 * F_STAGE2, "the conv3 stages and the triple buffer cover the W2 region");

__device__ __forceinline__ float max_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.0e38f); }
__device__ __forceinline__ float min_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, NEG_BIG); }

// Eight values x and a power of two s -> the two fp16 pieces of the three-product split of x s (csrc/gemm_f16x2.hip's, the second
// piece NOT scaled by 2^11 here): p1 = fp16(x s), p2 = fp16(x s - p1), each ONE rounding of an exact fma (v_fma_mixlo / mixhi_f16) --
// two vector instructions per value, no separate scaling or conversion.  (|p2| <= 2^-11 |p1|: with the point's largest activation in
// [2^14, 2^15) a second piece below fp16's normal range belongs to an activation 2^-17 of the largest; what is lost there is 2^-40 of it.)
__device__ __forceinline__ void q_split2(const float (&v)[8], float s, qf16x8& p1, qf16x8& p2) {
    unsigned hb[4], lb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = v[2 * j], a1 = v[2 * j + 1];
        asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(hb[j]) : "v"(a0), "v"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(hb[j]) : "v"(a1), "v"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lb[j]) : "v"(a0), "v"(s), "v"(hb[j]));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lb[j]) : "v"(a1), "v"(s), "v"(hb[j]));
    }
    p1 = __builtin_bit_cast(qf16x8, uint4{hb[0], hb[1], hb[2], hb[3]});
    p2 = __builtin_bit_cast(qf16x8, uint4{lb[0], lb[1], lb[2], lb[3]});
}

// W2 planes [2][128][64] fp16: rows of 128 B, chunk c of row r lands at c ^ ((r >> 1) & 7); one half = 64 rows of both planes =
// sixteen 1 KiB DMA pieces, four per wave
__device__ __forceinline__ void w2_issue(const uint16_t* __restrict__ planes, int row0, char* stage, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int id = wave * 4 + i;
        const int pl = id >> 3, rb = id & 7;
        const int row = rb * 8 + (lane >> 3);
        const uint16_t* src = planes + pl * (128L * 64) + (long)(row0 + row) * 64 + 8 * ((lane & 7) ^ ((row >> 1) & 7));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + (pl * 64 + rb * 8) * 128), 16, 0, 0);
    }
}
__device__ __forceinline__ qf16x8 w2_frag(const char* stage, int pl, int row, int chunk) {
    return *reinterpret_cast<const qf16x8*>(stage + (pl * 64 + row) * 128 + 16 * (chunk ^ ((row >> 1) & 7)));
}

// W3 filter image: [1024][128] fp16, rows of 256 B = 16 chunks; in LDS chunk c of row r sits at chunk c ^ (r & 15).
// One 64-channel chunk = 16 KiB: every lane moves 4 x 16 B through registers (global loads at the top of a chunk, LDS writes
// at its end -- an LDS-DMA piece costs the issuing wave ~150 cycles, a load + a write a few).
struct W3Regs { uint4 a, b, c, d; };
__device__ __forceinline__ const uint4* w3_src(const char* __restrict__ w3h, int ch0, int wave, int lane, int i) {
    // uniform 64-bit base + 32-bit lane offset: the load takes the base from scalar registers (no 64-bit vector add per load)
    const char* base = w3h + (long)(ch0 + (wave * 4 + i) * 4) * 256;
    const unsigned off = (unsigned)(lane >> 4) * 256u + 16u * (unsigned)(lane & 15);
    return reinterpret_cast<const uint4*>(base + off);
}
__device__ __forceinline__ W3Regs w3_load(const char* __restrict__ w3h, int ch0, int wave, int lane) {
    W3Regs v;
    v.a = *w3_src(w3h, ch0, wave, lane, 0);
    v.b = *w3_src(w3h, ch0, wave, lane, 1);
    v.c = *w3_src(w3h, ch0, wave, lane, 2);
    v.d = *w3_src(w3h, ch0, wave, lane, 3);
    return v;
}
__device__ __forceinline__ uint4* w3_dst(char* stage, int wave, int lane, int i) {
    const int row = (wave * 4 + i) * 4 + (lane >> 4);
    return reinterpret_cast<uint4*>(stage + row * 256 + 16 * ((lane & 15) ^ (row & 15)));
}
__device__ __forceinline__ void w3_store(char* stage, int wave, int lane, const W3Regs& v) {
    *w3_dst(stage, wave, lane, 0) = v.a;
    *w3_dst(stage, wave, lane, 1) = v.b;
    *w3_dst(stage, wave, lane, 2) = v.c;
    *w3_dst(stage, wave, lane, 3) = v.d;
}
__device__ __forceinline__ qf16x8 w3_frag(const char* stage, int row, int chunk) {
    return *reinterpret_cast<const qf16x8*>(stage + row * 256 + 16 * (chunk ^ (row & 15)));
}

// top three of the union of two descending triples
__device__ __forceinline__ void merge3(float& a1, float& a2, float& a3, float b1, float b2, float b3) {
    const float x = min_nc(a1, b1), y = max_nc(a2, b2), z = min_nc(a2, b2), w = max_nc(a3, b3);
    a1 = max_nc(a1, b1);
    a2 = max_nc(x, y);
    a3 = max_nc(min_nc(x, y), max_nc(z, w));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Points are dealt to the tiles round robin and to the 256 slots of a tile (wave, block, lane) through a multiplicative
// permutation: neighbours in the cloud's order -- often neighbours in space, i.e. near ties -- land in different waves.
// ``deal`` tiles share the first 256 * deal points this way.  A cloud with 1 .. 32 points beyond a multiple of 256 (the 778 hand
// vertices: 3 * 256 + 10) gets them as a TAIL tile (index deal) of ONE 32-point block instead of a fourth full tile that is
// three quarters padding: point 256 * deal + (slot & 31) (callers fold indices >= N back with % N, as for every padding slot).
__device__ __forceinline__ int point_of_slot(int tile, int slot, int deal) {
    return tile < deal ? ((slot * 67) & 255) * deal + tile : 256 * deal + (slot & 31);
}
// id bits of a tracked score: [3:0] accumulator register, [4] point block, [5] lane half, [7:6] wave -> slot inside the tile
__device__ __forceinline__ int slot_of_id(unsigned id) {
    const int e = id & 15, pb = (id >> 4) & 1, h = (id >> 5) & 1, w = (id >> 6) & 3;
    return w * 64 + pb * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
}

// Wave reductions without ds_bpermute (__shfl_xor: an LDS-crossbar instruction, an lgkmcnt wait and an address register per pattern):
// lanes l and l ^ 32 through v_permlane32_swap, rows through DPP, the four rows through v_readlane.
__device__ __forceinline__ float g_half_sum(float v) {    // v[l] + v[l ^ 32], in every lane
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float g_half_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
template <int CTRL>
__device__ __forceinline__ float g_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float g_wave_max(float v) {    // uniform result
    v = fmaxf(v, g_dpp<0xB1>(v));
    v = fmaxf(v, g_dpp<0x4E>(v));
    v = fmaxf(v, g_dpp<0x141>(v));
    v = fmaxf(v, g_dpp<0x140>(v));                         // every lane: the maximum of its row of 16
    v = g_half_max(v);                                     // rows 0 | 2, 1 | 3
    return fmaxf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)));
}
__device__ __forceinline__ float g_wave_sum(float v) {    // uniform result
    v += g_dpp<0xB1>(v);
    v += g_dpp<0x4E>(v);
    v += g_dpp<0x141>(v);
    v += g_dpp<0x140>(v);
    v = g_half_sum(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
}
// TAIL = false: one workgroup per (sample, dealt tile), blockIdx.x = sample * deal + tile (no workgroup for a tail tile: launched
// and left at once they would all sit on two of the eight XCDs -- blockIdx % 4 == 3 -- and idle a quarter of the chip).  TAIL = true: the tail tiles of FOUR samples per workgroup, one per wave (one 32-point block each;
// the staged W2 / W3 images are shared, everything per sample is per wave: centre, scales, records).
template <int C, bool TAIL>
__global__ __launch_bounds__(256, 2) void pn_trunk_filter_kernel(const float* __restrict__ pc, const float* __restrict__ trans,
                                                                 int N, int Npad, int tiles, int deal, long B, const float* __restrict__ W1,
                                                                 const float* __restrict__ b1,
                                                                 const float* __restrict__ b2, const char* __restrict__ w3f,
                                                                 float* __restrict__ h2buf, f32x4* __restrict__ part, qf32x2* __restrict__ part2,
                                                                 unsigned* __restrict__ tstat, const float* __restrict__ cbuf,
                                                                 int abl_arg /* timing diagnostics only (DVQ_PN_ABL, -DDVQ_DIAG builds) */) {
    const int abl = DVQ_DIAG_ON ? abl_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char fl[];
    float* tb = reinterpret_cast<float*>(fl + F_OFF_TB);
    float* w1s = reinterpret_cast<float*>(fl + F_OFF_W1);
    float* b1s = reinterpret_cast<float*>(fl + F_OFF_B1);
    float* b2s = reinterpret_cast<float*>(fl + F_OFF_B2);
    float* scs = reinterpret_cast<float*>(fl + F_OFF_SC);
    float* tis = reinterpret_cast<float*>(fl + F_OFF_TI);
    float* wst = reinterpret_cast<float*>(fl + F_OFF_WS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // The lane half enters the LDS addresses of the small tables through an opaque copy: knowing h in {0, 1}, the compiler turns
    // "base + 32 h + constant" into "(base | 32 h) | constant", cannot fold the constant into the instruction's offset field any more
    // and keeps one address REGISTER per constant -- 32 of them for conv1's weights alone, live across both point blocks.
    int h_op = h;
    asm("" : "+v"(h_op));
    constexpr int NPB = TAIL ? 1 : 2;                       // 32-point blocks per wave
    const long b_raw = TAIL ? (long)blockIdx.x * 4 + wave : (long)(blockIdx.x / deal);
    const bool live = !TAIL || b_raw < B;                  // tail: the last workgroup's surplus waves work on sample B - 1, store nothing
    const long b = live ? b_raw : B - 1;
    const int tile = TAIL ? deal : (int)(blockIdx.x % deal);
    const long rec = b * tiles + tile;                     // (sample, tile) record
    float* cs = reinterpret_cast<float*>(fl + (TAIL ? F_LDS + 3 * 4096 + 512 * wave : F_OFF_CS));
    float* e2s = reinterpret_cast<float*>(fl + F_OFF_E2) + (TAIL ? 1024 * wave : 0);

    const unsigned long long t_start = (abl & 4096) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long r_start = (abl & 8192) ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz: with 4096 | 8192 the record's word 3 holds the CLOCK
    unsigned long long t_a = 0, t_b = 0, t_c = 0;
    const uint16_t* w2pl = reinterpret_cast<const uint16_t*>(w3f + IMG_OFF_W2);
    w2_issue(w2pl, 0, fl, wave, lane);
    w2_issue(w2pl, 64, fl + F_STAGE2, wave, lane);
    float* k2s = reinterpret_cast<float*>(fl + F_OFF_K2);
    if (tid < 128) k2s[tid] = reinterpret_cast<const float*>(w3f + IMG_OFF_K2)[tid];
    w1s[tid] = W1[tid];
    if (tid < 64) b1s[tid] = b1[tid];
    if (tid < 128) b2s[tid] = b2[tid];
    *reinterpret_cast<f32x4*>(tis + 4 * tid) = *reinterpret_cast<const f32x4*>(w3f + IMG_OFF_TI + 16 * tid);
    if (TAIL) { cs[lane] = cbuf[b * 128 + lane]; cs[64 + lane] = cbuf[b * 128 + 64 + lane]; }
    else if (tid < 128) cs[tid] = cbuf[b * 128 + tid];

    float xin[NPB][4];
    int pidx[NPB];
    bool badpt = false;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        int p = point_of_slot(tile, TAIL ? r : wave * 64 + pb * 32 + r, deal);
        pidx[pb] = p;
        if (p >= N) p %= N;                               // padding slots repeat real points cyclically (a max ignores repeats; a point
                                                          // repeated once costs nothing: both copies carry ids that map back to it)
        const float* src = pc + b * (long)C * N + p;
        float x0 = src[0], x1 = src[N], x2 = src[2L * N];
        const float x3 = (C > 3) ? src[3L * N] : 0.f;
        if (trans) {                                      // xyz @ trans[b]  (pointnet_encoder.py:146)
            const float* t = trans + b * 9;
            const float n0 = fmaf(x2, t[6], fmaf(x1, t[3], x0 * t[0]));
            const float n1 = fmaf(x2, t[7], fmaf(x1, t[4], x0 * t[1]));
            const float n2 = fmaf(x2, t[8], fmaf(x1, t[5], x0 * t[2]));
            x0 = n0; x1 = n1; x2 = n2;
        }
        xin[pb][0] = x0; xin[pb][1] = x1; xin[pb][2] = x2; xin[pb][3] = x3;
        // a non-finite coordinate (after the transform): the reference's features of such a cloud are NaN in every channel -- affine
        // layers and torch.max propagate it.  ReLU as fmaxf(x, 0) squashes it here, and the cloud would go on as a DEGENERATE one
        // (all its points tie: every group flagged, every channel evaluated over all points by one workgroup -- the straggler of its
        // launch).  The tile says so instead (word 3 of its maxima) and pn_exact_kernel writes the NaNs.
        badpt = badpt || !(fabsf(x0) < 3.0e38f) || !(fabsf(x1) < 3.0e38f) || !(fabsf(x2) < 3.0e38f) || !(fabsf(x3) < 3.0e38f);
    }
    dvq_dma_barrier();                                    // W1/b1/b2 visible, W2 planes landed
    // The same wait once more in a form the compiler's counter model sees (vmcnt(0), the other counters untouched).  Without it the
    // FIRST use of the second point block's coordinates -- loaded above, consumed after the first block's sixteen h2 stores -- gets an
    // "s_waitcnt vmcnt(3)": the stores sit in a branch (padding slots store nothing), the compiler takes the smaller count of the two
    // paths, and the wave waits for thirteen of its sixteen stores to be ACKNOWLEDGED before it goes on (round 5: -11 % of the kernel
    // with the stores ablated, all of it this wait).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (abl & 4096) t_a = __builtin_amdgcn_s_memtime();
    float cnorm;                                          // |c| (every wave for itself: no ordering between the waves needed)
    {
        const float cq = fmaf(cs[lane], cs[lane], cs[64 + lane] * cs[64 + lane]);
        cnorm = sqrtf(g_wave_sum(cq)) * 1.0001f;
    }

    // ---- conv1 + conv2, h2 = relu(conv2 + b2) kept in fp32: hv[pb][16 t4 + e].  conv2 runs on the fp16 THREE-product split of
    // csrc/gemm_f16x2.hip since round 5 (six bf16 products before): weights as two fp16 planes of w * 2^t_n (per output row), the
    // activations of a point as two fp16 pieces of h1 * s_p with s_p a power of two that puts the POINT's largest activation in
    // [2^14, 2^15) -- a function of the point alone, so a row's bits do not depend on which points share its wave (tail tile ==
    // full tile, batched == single); acc = a1 w2 + a2 w1 + a1 w1 in ONE fp32 accumulator (second pieces unscaled), h2 = acc / s_p 2^-t_n + b2.
    float hv[NPB][64];
    if (abl & 131072) {                                    // timing only: a "consumer" workgroup -- no conv1 / conv2, rows from thin air
#pragma unroll
        for (int pb = 0; pb < NPB; ++pb)
#pragma unroll
            for (int i = 0; i < 64; ++i) hv[pb][i] = xin[pb][i & 3] * (float)(i + 1);
    } else
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        float v[4][8];
        float amax = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w1s + 32 * h_op + 64 * s + 4 * j);   // row k = 16 s + 8 h + j
                float a = fmaf(xin[pb][0], w[0], (b1s + 8 * h_op)[16 * s + j]);
                a = fmaf(xin[pb][1], w[1], a);
                a = fmaf(xin[pb][2], w[2], a);
                if constexpr (C > 3) a = fmaf(xin[pb][3], w[3], a);
                v[s][j] = fmaxf(a, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) amax = fmaxf(fmaxf(amax, v[s][j]), v[s][j + 1]);   // v_max3_f32 (a NaN is dropped here and reaches the products through the pieces)
        }
        amax = g_half_max(amax);                          // the lane halves hold the two halves of a point's 64 activations
        float s_p = 1.f, r_p = 1.f;
        {
            const int ex = (int)((__float_as_uint(amax) >> 23) & 255u);              // amax in [2^(ex-127), 2^(ex-126))
            if (ex > 20 && ex < 235) {
                s_p = __uint_as_float((unsigned)(127 + 15 - (ex - 126)) << 23);
                r_p = __uint_as_float((unsigned)(127 - 15 + (ex - 126)) << 23);
            }
        }
        qf16x8 h1a[4], h1b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) q_split2(v[s], s_p, h1a[s], h1b[s]);
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const char* st = fl + (t4 >> 1) * F_STAGE2;
            const int row = 32 * (t4 & 1) + r;
            // ONE fp32 accumulator: the eight small products (a1 w2, a2 w1: 2^-11 of the large ones) first, the four large ones after
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            qf16x8 w1f[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                w1f[s] = w2_frag(st, 0, row, 2 * s + h);
                const qf16x8 w2f = w2_frag(st, 1, row, 2 * s + h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f, h1a[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[s], h1b[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[s], h1a[s], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch = 32 * t4 + (e & 3) + 8 * (e >> 2);       // + 4 h
                hv[pb][16 * t4 + e] = fmaxf(fmaf(acc[e] * r_p, (k2s + 4 * h_op)[ch], (b2s + 4 * h_op)[ch]), 0.f);
            }
            if (pidx[pb] < N && live && !(abl & 1)) {     // natural channel order: 4 consecutive channels per 16-byte store
                float* dst = h2buf + ((b * Npad + pidx[pb]) * 128 + 32 * t4 + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(dst + 8 * g) =
                        f32x4{hv[pb][16 * t4 + 4 * g], hv[pb][16 * t4 + 4 * g + 1], hv[pb][16 * t4 + 4 * g + 2], hv[pb][16 * t4 + 4 * g + 3]};
            }
        }
    }
    if (abl & 4096) t_b = __builtin_amdgcn_s_memtime();
    // ---- centre the rows on the sample's centre (pn_center_kernel)
    float dn2 = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(cs + 4 * h_op + 32 * t4 + 8 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int pb = 0; pb < NPB; ++pb) hv[pb][16 * t4 + 4 * g + i] -= c4[i];
            }
        }
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) sq = fmaf(hv[pb][i], hv[pb][i], sq);
        sq = g_half_sum(sq);                              // the two lane halves hold the two halves of a point's channels
        dn2 = fmaxf(dn2, sq);
    }
    dn2 = g_wave_max(dn2);
    // per-wave power-of-two scale: (largest row norm) * s in [2^14, 2^15) -- every element is at most its row's norm
    float s_w = 1.f;
    const float dnorm = sqrtf(dn2) * 1.0001f;
    {
        const int ex = (int)((__float_as_uint(dnorm) >> 23) & 255u);             // dnorm in [2^(ex-127), 2^(ex-126))
        if (ex > 20 && ex < 235) s_w = __uint_as_float((unsigned)(127 + 15 - (ex - 126)) << 23);
    }
    // conv3's A operand, one fp16 plane: step = 2 t4 + q, k order inside a step as conv2's accumulator delivers it;
    // rn2 = largest squared norm of a row's rounding residual (scaled units)
    qf16x8 a3[NPB][8];
    float rn2 = 0.f;
#pragma unroll
    for (int pb = 0; pb < NPB; ++pb) {
        float sq = 0.f;
        const float s_pb = s_w;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            unsigned pk[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                // fp16(d s) and the rounding residual d s - fp16(d s), each one instruction on the exact product (s a power of two)
                const float d0 = hv[pb][8 * st + 2 * j2], d1 = hv[pb][8 * st + 2 * j2 + 1];
                asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(pk[j2]) : "v"(d0), "v"(s_pb));
                asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(pk[j2]) : "v"(d1), "v"(s_pb));
                float r0, r1;
                asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(d0), "v"(s_pb), "v"(pk[j2]));
                asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(d1), "v"(s_pb), "v"(pk[j2]));
                sq = fmaf(r0, r0, sq);
                sq = fmaf(r1, r1, sq);
            }
            a3[pb][st] = __builtin_bit_cast(qf16x8, uint4{pk[0], pk[1], pk[2], pk[3]});
        }
        sq = g_half_sum(sq);
        rn2 = fmaxf(rn2, sq);
    }
    rn2 = g_wave_max(rn2);
    // conv3's first W3 chunk: requested BEFORE the three atomics below -- the wait for these loads then leaves the atomics (younger,
    // in the in-order counter) pending instead of sitting out their round trip (600 .. 3 000 cycles each under load)
    const char* w3h = w3f;
    W3Regs wreg = w3_load(w3h, 0, wave, lane);
    const bool any_bad = __any(badpt);
    if (lane == 0) {              // per-tile maxima; non-negative floats (and NaN, above all of them) order as integers
        const float dmx = sqrtf(dn2), rdm = sqrtf(rn2) / s_w;
        const float hm = (dmx + cnorm) * 1.0001f;          // |h_p| <= |h_p - c| + |c|
        scs[wave] = 1.0f / s_w;
        wst[wave] = hm; wst[4 + wave] = dmx; wst[8 + wave] = rdm;
        if (live) {
            if (any_bad && !(abl & 4096)) atomicMax(tstat + 4 * rec + 3, 1u);
            atomicMax(tstat + 4 * rec + 0, __float_as_uint(hm));
            atomicMax(tstat + 4 * rec + 1, __float_as_uint(dmx));
            atomicMax(tstat + 4 * rec + 2, __float_as_uint(rdm));
        }
    }
    dvq_lds_barrier();                                      // everybody is done with W2 in the stages; scs visible
    if (abl & 4096) t_c = __builtin_amdgcn_s_memtime();

    // ---- conv3, filtered: 16 chunks of 64 channels, one fp16 product, top two scores per channel and 16-point group
    w3_store(fl, wave, lane, wreg);
    {
        // 2 E of this tile for every channel, once, into the LDS (the publishing waves used to fetch the two weight norms of their
        // channels from global memory at the top of every publish: an L2 round trip in front of four idle waves, four times)
        const float* wnorm_g = reinterpret_cast<const float*>(w3f + IMG_OFF_WN);
        const float* rnorm_g = reinterpret_cast<const float*>(w3f + IMG_OFF_RN);
        // the tile's maxima: over its four waves -- over the wave alone where every wave is a tile of its own (TAIL)
        const float hm = (TAIL ? wst[wave] : fmaxf(fmaxf(wst[0], wst[1]), fmaxf(wst[2], wst[3]))) * 1.00001f;
        const float dmx = (TAIL ? wst[4 + wave] : fmaxf(fmaxf(wst[4], wst[5]), fmaxf(wst[6], wst[7]))) * 1.00001f;
        const float rdm = (TAIL ? wst[8 + wave] : fmaxf(fmaxf(wst[8], wst[9]), fmaxf(wst[10], wst[11]))) * 1.00001f;
        constexpr int PER = TAIL ? 16 : 4;                 // channels per lane: a wave fills its own table (TAIL), the workgroup one
#pragma unroll
        for (int i = 0; i < PER; i += 4) {
            float wn[4], rn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = (TAIL ? lane : tid) + (TAIL ? 64 : 256) * (i + u);
                wn[u] = wnorm_g[n];
                rn[u] = rnorm_g[n];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = (TAIL ? lane : tid) + (TAIL ? 64 : 256) * (i + u);
                e2s[n] = 2.0f * fmaf(rn[u], dmx, fmaf(wn[u], rdm, fmaf(C_ID * wn[u], dmx, 2.0f * DELTA * wn[u] * hm)));
            }
        }
    }
    // Per channel of chunk c the publishing wave merges the sorted PAIRS of the sixteen 16-point groups (4 waves x 2 point blocks x 2
    // lane halves) into the tile's FIVE largest id-carrying scores (real units: three + the flags in a 16-byte record, the fourth and
    // fifth in an 8-byte one that pn_exact_kernel reads only where the third is in range) and one flag per group "may hold a point
    // within 2 E of the tile's largest score that is not among the five".  The ring holds four chunks: after chunks 3, 7, 11 and 15
    // every wave publishes one (all four busy at the same time: no wave waits for a publisher at the chunk barriers).
    auto publish = [&](int c) {
        const float* src = tb + (c & 3) * F_SLOT;          // [wave][point block][half][k][channel]: every lane's own sorted pair, as finish() left it
        const float ti = tis[64 * c + lane];
        // all sixteen pairs first (one trip to the LDS), in real units, with the rest of their ids: bits [4:0] point block + register
        // (the chain's), 5 lane half, [7:6] wave
        constexpr int NG = TAIL ? 2 : 16;                  // groups: 4 w + 2 pb + hh (TAIL: the publishing wave's two lane halves, "wave 0")
        float t1[NG], t2[NG];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int w = TAIL ? 0 : gi >> 2, pb = TAIL ? 0 : (gi >> 1) & 1, hh = gi & 1;
            const int ws = TAIL ? wave : w;                // whose pairs (TAIL: one wave = the whole tile)
            const float* q = src + (((ws * 2 + pb) * 2 + hh) * 2) * 64 + lane;
            t1[gi] = q[0];
            t2[gi] = q[64];
        }
        // Run-time check of the hand-over: finish() stamps bits [6:5] of what it stores with (chunk / 4) mod 4, so a value left in
        // this ring slot by an earlier chunk -- a store that did not happen or was not seen -- shows.  One stale input and the record
        // cannot be trusted: every group is flagged (pn_exact_kernel then evaluates the whole tile for this channel) and bit 16 says
        // why (counted: dvq_pointnet_fault_counters).  The pair of a group is ONE LDS store: checking its first value covers both.
        unsigned tagdiff = 0;
#pragma unroll
        for (int gi = 0; gi + 1 < NG; gi += 2)             // v_xor_b32 + v_or3_b32 per two groups
            tagdiff = tagdiff | (__float_as_uint(t1[gi]) ^ (unsigned)(((c >> 2) & 3) << 5)) | (__float_as_uint(t1[gi]) ^ __float_as_uint(t1[gi + 1]));
        const bool suspect = (tagdiff & 0x60u) != 0;
        unsigned tag_mask = ~0xE0u;
        asm volatile("" : "+v"(tag_mask));
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int w = TAIL ? 0 : gi >> 2, hh = gi & 1;
            const float sc = scs[TAIL ? wave : w] * ti;     // a power of two
            unsigned tag;                                   // (w << 6) | (hh << 5) as a SCALAR and the mask in a vector register: one
            asm("s_mov_b32 %0, %1" : "=s"(tag) : "i"((w << 6) | (hh << 5)));   // v_and_or_b32 per value instead of v_and + v_or
            t1[gi] = __uint_as_float((__float_as_uint(t1[gi] * sc) & tag_mask) | tag);
            t2[gi] = __uint_as_float((__float_as_uint(t2[gi] * sc) & tag_mask) | tag);
        }
        const float e2 = e2s[64 * c + lane];               // 2 E of the tile for this channel (filled once, before the loop)
        float c1 = NEG_BIG, c2 = NEG_BIG, c3 = NEG_BIG, c4 = NEG_BIG, c5 = NEG_BIG;   // the tile's FIVE largest id-carrying scores
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            float x = t1[gi];
            c5 = __builtin_amdgcn_fmed3f(c4, c5, x); c4 = __builtin_amdgcn_fmed3f(c3, c4, x);
            c3 = __builtin_amdgcn_fmed3f(c2, c3, x); c2 = __builtin_amdgcn_fmed3f(c1, c2, x); c1 = max_nc(c1, x);
            x = t2[gi];                                    // <= t1[gi] <= c1: never the new largest
            c5 = __builtin_amdgcn_fmed3f(c4, c5, x); c4 = __builtin_amdgcn_fmed3f(c3, c4, x);
            c3 = __builtin_amdgcn_fmed3f(c2, c3, x); c2 = __builtin_amdgcn_fmed3f(c1, c2, x);
        }
        const float thr = c1 - e2;                         // NaN -> no flag here; pn_exact_kernel sees the non-finite bound
        // bit 4 w + 2 pb + h: that 16-point group may hold a point in range that is not among the five: its SECOND is in range (a
        // third could be: the lanes keep two), or its first is in range and was not kept (six in range in the tile).  With t1 >= t2
        // that is "u >= thr" for u = t1 if t1 was not kept (t1 < c5), else t2 -- no branches, no second trip to the LDS.
        unsigned flags = 0;
#pragma unroll
        for (int gi = NG - 1; gi >= 0; --gi) {
            const float u = t1[gi] < c5 ? t1[gi] : t2[gi];
            flags = flags + flags + (unsigned)(u >= thr);
        }
        if (suspect) flags = 0x1FFFFu;
        if ((abl & 32768) && c == 5 && lane == 7) c1 = fabsf(c1) * 1.0e3f + 1.0f;   // diagnostics: a record that lies about its tile
        if (live) {
            part[rec * 1024 + 64 * c + lane] = f32x4{c1, c2, c3, __uint_as_float(flags)};
            part2[rec * 1024 + 64 * c + lane] = qf32x2{c4, c5};
        }
    };
    // one 32-point x 32-channel block: 8 MFMAs; its 16 scores per lane go through the top-two chain (3 vector instructions per
    // score) while the NEXT block's MFMAs run: 1 MFMA (32 cycles of the matrix pipe) per 6 chain instructions
#define F_MFMA_BLOCK(ACC, PB, WF)                                                                              \
    do {                                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) ACC[e] = 0.f;                                           \
        _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                          \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[PB][s], WF[s], ACC, 0, 0, 0);                      \
    } while (0)
    /* CT: the ring tag of the chunk (bits [6:5], uniform): set HERE, with the id, by the one instruction per score that is needed    */ \
    /* anyway -- finish() used to set it on the block's winner with an instruction of its own                                        */
#define F_CHAIN_BLOCK(ACC, PB, M1, M2, CT)                                                                     \
    do {                                                                                                       \
        if (abl & 128) { M1 = ACC[0]; M2 = ACC[15]; } else   /* timing only: no chain */                       \
        {                                                                                                      \
            /* round 6: groups of three -- v_max3 / v_med3 give a group's two largest, merged into the running pair by          */ \
            /* second = med3(M1, g1, max(M2, g2)): 40 instead of 48 instructions per 16 scores, the same pair                    */ \
            float x_[16];                                                                                      \
            _Pragma("unroll") for (int e = 0; e < 16; ++e)                                                     \
                x_[e] = __uint_as_float((__float_as_uint(ACC[e]) & id_mask) | (CT)[16 * (PB) + e]);           \
            M1 = fmaxf(fmaxf(x_[0], x_[1]), x_[2]);                                                            \
            M2 = __builtin_amdgcn_fmed3f(x_[0], x_[1], x_[2]);                                                 \
            _Pragma("unroll") for (int g = 3; g < 15; g += 3) {                                                \
                const float g1_ = fmaxf(fmaxf(x_[g], x_[g + 1]), x_[g + 2]);                                   \
                const float g2_ = max_nc(M2, __builtin_amdgcn_fmed3f(x_[g], x_[g + 1], x_[g + 2]));            \
                M2 = __builtin_amdgcn_fmed3f(M1, g1_, g2_);                                                    \
                M1 = max_nc(M1, g1_);                                                                          \
            }                                                                                                  \
            M2 = __builtin_amdgcn_fmed3f(M1, M2, x_[15]);                                                      \
            M1 = max_nc(M1, x_[15]);                                                                           \
        }                                                                                                      \
    } while (0)
#define F_INTERLEAVE()                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                                 \
        }                                                                                                      \
    } while (0)
    // every lane hands its own sorted pair of a block over (16 points of one channel); the publishing wave merges the sixteen groups
    // of a (tile, channel) -- one lane per channel there
    auto finish = [&](int c, int jn, int pb, float m1, float m2) {
        if ((abl & 65536) && c == 9 && wave == 2 && jn == 0) return;   // diagnostics: a hand-over that does not happen
        float* dst = tb + (c & 3) * F_SLOT + ((wave * 2 + pb) * 2 + h) * 128 + 32 * jn + r;
        dst[0] = m1;                                        // bits [6:5] of both: the chunk's ring tag (F_CHAIN_BLOCK), checked by publish()
        dst[64] = m2;
    };
    // (the mask in a vector register: id | tag is a scalar, and one v_and_or_b32 takes one scalar operand)
    unsigned id_mask = ~127u;
    asm volatile("" : "+v"(id_mask));
    int stage = 0;
    if constexpr (TAIL) {
        // one point block per wave: two MFMA blocks (channels 0..31, 32..63) and two chains per chunk; every wave publishes the
        // eight chunks of ITS sample after chunk 7 and after chunk 15
#pragma unroll 1
        for (int c = 0; c < 16; ++c) {
            dvq_lds_barrier();                              // chunk c is in its stage; the other stage and tb parity are free
            if (c + 1 < 16) wreg = w3_load(w3h, 64 * (c + 1), wave, lane);
            if ((c & 3) == 0 && c > 0) {                  // the ring's four chunks are complete (barrier above)
#pragma unroll 1
                for (int q = c - 4; q < c; ++q) publish(q);
                dvq_lds_barrier();                          // before this chunk's pairs overwrite slot 0
            }
            const char* st = fl + stage * F_STAGE3;
            qf16x8 wf0[8], wf1[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) wf0[s] = w3_frag(st, r, 2 * s + h);
#pragma unroll
            for (int s = 0; s < 8; ++s) wf1[s] = w3_frag(st, 32 + r, 2 * s + h);
            f32x16 accA, accB;
            float a1, a2, q1, q2;
            F_MFMA_BLOCK(accA, 0, wf0);
            F_MFMA_BLOCK(accB, 0, wf1);
            unsigned ctag_c[32];                          // id | ring tag per accumulator register, as scalars (see sid4 below)
#pragma unroll
            for (int e = 0; e < 16; ++e) asm("s_or_b32 %0, %1, %2" : "=s"(ctag_c[e]) : "s"((unsigned)(((c >> 2) & 3) << 5)), "i"(e));
            F_CHAIN_BLOCK(accA, 0, a1, a2, ctag_c);
            F_INTERLEAVE();
            F_CHAIN_BLOCK(accB, 0, q1, q2, ctag_c);
            finish(c, 0, 0, a1, a2);
            finish(c, 1, 0, q1, q2);
            if (c + 1 < 16) w3_store(fl + (stage ^ 1) * F_STAGE3, wave, lane, wreg);
            stage ^= 1;
        }
    } else {
    f32x16 accP;                                          // the chunk's last accumulator block, scored under the next chunk's first MFMAs
    // Four chunks per trip of the rolled loop, the four written out: the ring slot (c & 3), the stage (c & 1), "is a chain pending"
    // and "is this a publishing chunk" are compile-time constants then -- LDS addresses become instruction offsets instead of vector
    // additions per access, the conditions disappear (round 5: the kernel is vector-issue bound).
#pragma unroll 1
    for (int c4 = 0; c4 < ((abl & 2) ? 0 : 16); c4 += 4) {
    // id (point block, register) | ring tag of this trip's four chunks, one SCALAR per accumulator register: the chain's one instruction
    // per score is then v_and_or_b32 (score, mask in a vector register, this scalar).  (Written as "id | tag" in the expression the
    // compiler makes it v_and_b32 + v_or3_b32: two vector instructions per score.)
    unsigned ctag4[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) asm("s_or_b32 %0, %1, %2" : "=s"(ctag4[e]) : "s"((unsigned)(((c4 >> 2) & 3) << 5)), "i"(e));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = c4 + u;
        const int stage = u & 1;
        const bool pending = u != 0;
        if (!(abl & 16384)) dvq_lds_barrier();              // chunk c is in its stage; the other stage and tb parity are free
        if (c + 1 < 16 && !(abl & 1024)) wreg = w3_load(w3h, 64 * (c + 1), wave, lane);
        if (u == 0 && c > 0 && !(abl & 256)) {              // the ring's four chunks are complete (barrier above): one per wave
            publish(c - 4 + wave);
            dvq_lds_barrier();                              // before this chunk's pairs overwrite slot 0
        }
        const char* st = fl + stage * F_STAGE3;
        qf16x8 wf0[8], wf1[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) wf0[s] = w3_frag(st, r, 2 * s + h);
#pragma unroll
        for (int s = 0; s < 8; ++s) wf1[s] = w3_frag(st, 32 + r, 2 * s + h);
        // Four MFMA blocks and four chain blocks per chunk, each chain under the MFMAs of the block that follows its own.  The last
        // chain of a chunk (second point block x channels 32..63: accP) has no successor inside the chunk: it runs under the FIRST
        // MFMA block of the next chunk (8 dependent MFMAs that had nothing to cover them), except before a publish (c = 3, 7, 11, 15).
        f32x16 accA, accB;
        float m1, m2;
        F_MFMA_BLOCK(accA, 0, wf0);
        if (pending) {
            F_CHAIN_BLOCK(accP, 1, m1, m2, ctag4);
            F_INTERLEAVE();
            if (!(abl & 512)) finish(c - 1, 1, 1, m1, m2);
            else if (m1 + m2 == 12345.f) tb[lane] = m1;
        }
        F_MFMA_BLOCK(accB, 1, wf0);
        F_CHAIN_BLOCK(accA, 0, m1, m2, ctag4);
        F_INTERLEAVE();
        if (!(abl & 512)) finish(c, 0, 0, m1, m2);
        else if (m1 + m2 == 12345.f) tb[lane] = m1;
        F_MFMA_BLOCK(accA, 0, wf1);
        F_CHAIN_BLOCK(accB, 1, m1, m2, ctag4);
        F_INTERLEAVE();
        if (!(abl & 512)) finish(c, 0, 1, m1, m2);
        else if (m1 + m2 == 12345.f) tb[lane] = m1;
        F_MFMA_BLOCK(accP, 1, wf1);
        F_CHAIN_BLOCK(accA, 0, m1, m2, ctag4);
        F_INTERLEAVE();
        if (!(abl & 512)) finish(c, 1, 0, m1, m2);
        else if (m1 + m2 == 12345.f) tb[lane] = m1;
        if (u == 3) {                                     // nothing to cover it before a publish
            F_CHAIN_BLOCK(accP, 1, m1, m2, ctag4);
            if (!(abl & 512)) finish(c, 1, 1, m1, m2);
            else if (m1 + m2 == 12345.f) tb[lane] = m1;
        }
        if (c + 1 < 16 && !(abl & 1024)) w3_store(fl + (stage ^ 1) * F_STAGE3, wave, lane, wreg);
    }
    }
    }
#undef F_MFMA_BLOCK
#undef F_CHAIN_BLOCK
#undef F_INTERLEAVE
    dvq_lds_barrier();
    if constexpr (TAIL) {
#pragma unroll 1
        for (int q = 12; q < 16; ++q) publish(q);
    } else if (!(abl & 256)) {
        publish(12 + wave);
    }
    if ((abl & 4096) && tid == 0) {                        // diagnostics: phase durations in units of 64 ticks, 8 bits each
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        auto q = [](unsigned long long d) { d >>= 6; return (unsigned)(d > 255 ? 255 : d); };
        tstat[4 * rec + 3] = q(t_a - t_start) | (q(t_b - t_a) << 8) | (q(t_c - t_b) << 16) | (q((t_end - t_c) >> 3) << 24);
        if (abl & 8192) {                                   // in-kernel clock in MHz: shader cycles per 100 MHz tick over the workgroup's life
            const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
            tstat[4 * rec + 3] = (unsigned)((t_end - t_start) * 100ull / (r_end - r_start + 1));
        }
    }
}

#ifdef DVQ_DIAG   // diagnostics build only (round 6): measured 4.5 % slower than the two-per-CU kernel; kept for tests/ and tools/, not shipped
// ------------------------------------------------------------------------------------------------------------------------
// pn_trunk3_kernel: the full-tile trunk kernel laid out for THREE workgroups per CU (<= 168 registers, <= 53 KB of LDS); same
// arithmetic for conv1 / conv2 / h2, same record format, same run-time checks as pn_trunk_filter_kernel<C, false>.  What differs:
//   * conv3 in 32 chunks of 32 channels (two 8 KB W3 stages instead of two of 16 KB; 32 fragment registers instead of 64; two
//     accumulator blocks in flight instead of three);
//   * the ring holds eight small chunks (32 KB as before); after chunks 7, 15, 23, 31 every wave publishes TWO of them (lanes 0..31
//     one, 32..63 the other: 64 consecutive channels, every lane busy);
//   * 1 / scale and the two weight norms of a publishing lane's channel come from the image in L2, requested a chunk ahead (the
//     8 KB of LDS tables are what did not fit); 2 E is four FMAs per publish;
//   * the two point blocks of a wave go through conv1 / conv2 / centring / conversion one after the other, each with its OWN
//     power-of-two scale (the exact stage never sees the scale: records are published in real units).
constexpr int G_STAGE = 32 * 256;                         // one 32-channel chunk of the W3 image
constexpr int G_OFF_TB = 2 * G_STAGE;                     // ring: [8 small chunks][4 waves][2 point blocks][2 halves][2][32] fp32
constexpr int G_SLOT = 4 * 2 * 2 * 2 * 32;                // floats per small chunk
constexpr int G_OFF_W1 = G_OFF_TB + 8 * G_SLOT * 4;       // 48 KB: [64][4]
constexpr int G_OFF_B1 = G_OFF_W1 + 1024;
constexpr int G_OFF_B2 = G_OFF_B1 + 256;
constexpr int G_OFF_K2 = G_OFF_B2 + 512;
constexpr int G_OFF_CS = G_OFF_K2 + 512;
constexpr int G_OFF_SC = G_OFF_CS + 512;                  // [4][2] 1 / (point block scale)
constexpr int G_OFF_WS = G_OFF_SC + 64;                   // [3][4] per-wave |h|max, |d|max, |rd|max
constexpr int G_LDS = G_OFF_WS + 64;                      // 52 096 B -> 3 workgroups per CU
static_assert(G_OFF_W1 >= 2 * F_STAGE2, "the stages and the ring cover the W2 region");
static_assert(3 * G_LDS <= 160 * 1024, "three workgroups per CU");
constexpr int pn_g_lds() { return G_LDS; }

#ifndef PN3_WGS
#define PN3_WGS 3
#endif
struct W3Half { uint4 a, b; };
__device__ __forceinline__ W3Half g_w3_load(const char* __restrict__ w3h, int ch0, int wave, int lane) {
    W3Half v;
    const int row = wave * 8 + (lane >> 4);
    v.a = *reinterpret_cast<const uint4*>(w3h + (long)(ch0 + row) * 256 + 16 * (lane & 15));
    v.b = *reinterpret_cast<const uint4*>(w3h + (long)(ch0 + row + 4) * 256 + 16 * (lane & 15));
    return v;
}
__device__ __forceinline__ void g_w3_store(char* stage, int wave, int lane, const W3Half& v) {
    const int row = wave * 8 + (lane >> 4);
    *reinterpret_cast<uint4*>(stage + row * 256 + 16 * ((lane & 15) ^ (row & 15))) = v.a;
    *reinterpret_cast<uint4*>(stage + (row + 4) * 256 + 16 * ((lane & 15) ^ ((row + 4) & 15))) = v.b;
}

template <int C>
__global__ __launch_bounds__(256, PN3_WGS) void pn_trunk3_kernel(const float* __restrict__ pc, const float* __restrict__ trans,
                                                           int N, int Npad, int tiles, int deal, long B, const float* __restrict__ W1,
                                                           const float* __restrict__ b1, const float* __restrict__ b2,
                                                           const char* __restrict__ w3f, float* __restrict__ h2buf,
                                                           f32x4* __restrict__ part, qf32x2* __restrict__ part2,
                                                           unsigned* __restrict__ tstat, const float* __restrict__ cbuf,
                                                           int abl_arg /* fault injection only (DVQ_PN_ABL, -DDVQ_DIAG builds) */) {
    const int abl = DVQ_DIAG_ON ? abl_arg : 0;
    extern __shared__ __attribute__((aligned(16))) char fl[];
    float* tb = reinterpret_cast<float*>(fl + G_OFF_TB);
    float* w1s = reinterpret_cast<float*>(fl + G_OFF_W1);
    float* b1s = reinterpret_cast<float*>(fl + G_OFF_B1);
    float* b2s = reinterpret_cast<float*>(fl + G_OFF_B2);
    float* k2s = reinterpret_cast<float*>(fl + G_OFF_K2);
    float* cs = reinterpret_cast<float*>(fl + G_OFF_CS);
    float* scs = reinterpret_cast<float*>(fl + G_OFF_SC);
    float* wst = reinterpret_cast<float*>(fl + G_OFF_WS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // The lane half enters the LDS addresses of the small tables through an opaque copy: knowing h in {0, 1}, the compiler turns
    // "base + 32 h + constant" into "(base | 32 h) | constant", cannot fold the constant into the instruction's offset field any more
    // and keeps one address REGISTER per constant -- 32 of them for conv1's weights alone, live across both point blocks.
    int h_op = h;
    asm("" : "+v"(h_op));
    const float* w1h = w1s + 32 * h_op;
    const float* b1h = b1s + 8 * h_op;
    const float* k2h = k2s + 4 * h_op;
    const float* b2h = b2s + 4 * h_op;
    const float* csh = cs + 4 * h_op;
    const long b = (long)(blockIdx.x / deal);
    const int tile = (int)(blockIdx.x % deal);
    const long rec = b * tiles + tile;                     // (sample, tile) record

    const uint16_t* w2pl = reinterpret_cast<const uint16_t*>(w3f + IMG_OFF_W2);
    w2_issue(w2pl, 0, fl, wave, lane);
    w2_issue(w2pl, 64, fl + F_STAGE2, wave, lane);
    if (tid < 128) k2s[tid] = reinterpret_cast<const float*>(w3f + IMG_OFF_K2)[tid];
    w1s[tid] = W1[tid];
    if (tid < 64) b1s[tid] = b1[tid];
    if (tid < 128) b2s[tid] = b2[tid];
    if (tid < 128) cs[tid] = cbuf[b * 128 + tid];

    float xin[2][4];
    int pidx[2];
    bool badpt = false;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        int p = point_of_slot(tile, wave * 64 + pb * 32 + r, deal);
        pidx[pb] = p;
        if (p >= N) p %= N;                               // padding slots repeat real points cyclically
        const float* src = pc + b * (long)C * N + p;
        float x0 = src[0], x1 = src[N], x2 = src[2L * N];
        const float x3 = (C > 3) ? src[3L * N] : 0.f;
        if (trans) {                                      // xyz @ trans[b]  (pointnet_encoder.py:146)
            const float* t = trans + b * 9;
            const float n0 = fmaf(x2, t[6], fmaf(x1, t[3], x0 * t[0]));
            const float n1 = fmaf(x2, t[7], fmaf(x1, t[4], x0 * t[1]));
            const float n2 = fmaf(x2, t[8], fmaf(x1, t[5], x0 * t[2]));
            x0 = n0; x1 = n1; x2 = n2;
        }
        xin[pb][0] = x0; xin[pb][1] = x1; xin[pb][2] = x2; xin[pb][3] = x3;
        badpt = badpt || !(fabsf(x0) < 3.0e38f) || !(fabsf(x1) < 3.0e38f) || !(fabsf(x2) < 3.0e38f) || !(fabsf(x3) < 3.0e38f);   // see pn_trunk_filter_kernel
    }
    dvq_dma_barrier();                                    // W1/b1/b2/centre visible, W2 planes landed
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // (the compiler's counter model must see it: pn_trunk_filter_kernel)
    float cnorm;                                          // |c|
    {
        const float cq = fmaf(cs[lane], cs[lane], cs[64 + lane] * cs[64 + lane]);
        cnorm = sqrtf(g_wave_sum(cq)) * 1.0001f;
    }

    // ---- per point block: conv1, conv2 (three fp16 products, one accumulator), h2 to HBM, centring, norms, fp16 conversion
    qf16x8 a3[2][8];
    float dn2 = 0.f, rd2 = 0.f;                            // largest squared row norm of d; of the rounding residual in REAL units
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
        // ONE point block at a time: left alone the compiler starts the second block's conv1 under the first block's MFMAs and
        // spills its 32 activations (70 registers to scratch at the 168 this kernel may use)
        asm volatile("" : "+v"(xin[pb][0]), "+v"(xin[pb][1]), "+v"(xin[pb][2]), "+v"(xin[pb][3]) : : "memory");
        float v[4][8];
        float amax = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(w1h + 64 * s + 4 * j);   // row k = 16 s + 8 h + j
                float a = fmaf(xin[pb][0], w[0], b1h[16 * s + j]);
                a = fmaf(xin[pb][1], w[1], a);
                a = fmaf(xin[pb][2], w[2], a);
                if constexpr (C > 3) a = fmaf(xin[pb][3], w[3], a);
                v[s][j] = fmaxf(a, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) amax = fmaxf(fmaxf(amax, v[s][j]), v[s][j + 1]);
        }
        amax = g_half_max(amax);                          // the lane halves hold the two halves of a point's 64 activations
        float s_p = 1.f, r_p = 1.f;
        {
            const int ex = (int)((__float_as_uint(amax) >> 23) & 255u);
            if (ex > 20 && ex < 235) {
                s_p = __uint_as_float((unsigned)(127 + 15 - (ex - 126)) << 23);
                r_p = __uint_as_float((unsigned)(127 - 15 + (ex - 126)) << 23);
            }
        }
        qf16x8 h1a[4], h1b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) q_split2(v[s], s_p, h1a[s], h1b[s]);
        float hv[64];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const char* st = fl + (t4 >> 1) * F_STAGE2;
            const int row = 32 * (t4 & 1) + r;
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const qf16x8 w1f = w2_frag(st, 0, row, 2 * s + h), w2f = w2_frag(st, 1, row, 2 * s + h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f, h1a[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f, h1b[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)                    // (the first plane's fragments once more from the LDS: 16 registers less)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2_frag(st, 0, row, 2 * s + h), h1a[s], acc, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ch = 32 * t4 + (e & 3) + 8 * (e >> 2);       // + 4 h
                hv[16 * t4 + e] = fmaxf(fmaf(acc[e] * r_p, k2h[ch], b2h[ch]), 0.f);
            }
            if (pidx[pb] < N) {                           // natural channel order: 4 consecutive channels per 16-byte store
                float* dst = h2buf + ((b * Npad + pidx[pb]) * 128 + 32 * t4 + 4 * h);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(dst + 8 * g) = f32x4{hv[16 * t4 + 4 * g], hv[16 * t4 + 4 * g + 1], hv[16 * t4 + 4 * g + 2], hv[16 * t4 + 4 * g + 3]};
            }
            // centre these 16 channels at once
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(csh + 32 * t4 + 8 * g);
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[16 * t4 + 4 * g + i] -= c4[i];
            }
        }
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) sq = fmaf(hv[i], hv[i], sq);
        sq = g_half_sum(sq);                              // the two lane halves hold the two halves of a point's channels
        const float dn2_pb = g_wave_max(sq);
        dn2 = fmaxf(dn2, dn2_pb);
        // this point block's power-of-two scale: (largest row norm) * s in [2^14, 2^15)
        float s_w = 1.f;
        {
            const float dnorm = sqrtf(dn2_pb) * 1.0001f;
            const int ex = (int)((__float_as_uint(dnorm) >> 23) & 255u);
            if (ex > 20 && ex < 235) s_w = __uint_as_float((unsigned)(127 + 15 - (ex - 126)) << 23);
        }
        float rq = 0.f;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            unsigned pk[4];
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
                const float d0 = hv[8 * st + 2 * j2], d1 = hv[8 * st + 2 * j2 + 1];
                asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(pk[j2]) : "v"(d0), "v"(s_w));
                asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(pk[j2]) : "v"(d1), "v"(s_w));
                float r0, r1;
                asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(d0), "v"(s_w), "v"(pk[j2]));
                asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(d1), "v"(s_w), "v"(pk[j2]));
                rq = fmaf(r0, r0, rq);
                rq = fmaf(r1, r1, rq);
            }
            a3[pb][st] = __builtin_bit_cast(qf16x8, uint4{pk[0], pk[1], pk[2], pk[3]});
        }
        rq = g_half_sum(rq);
        const float inv = 1.0f / s_w;                     // a power of two
        rd2 = fmaxf(rd2, g_wave_max(rq) * inv * inv);
        if (lane == 0) scs[wave * 2 + pb] = inv;
        // the second half of the first block's fp16 rows waits in the upper half of the ring (idle until chunk 4; the W2 planes end
        // below it) while the second block is worked on: 16 registers the second block's conv2 needs
        uint4* park = reinterpret_cast<uint4*>(fl + 32768 + (wave * 64 + lane) * 64);
        if (pb == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) park[i] = __builtin_bit_cast(uint4, a3[0][4 + i]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a3[0][4 + i] = __builtin_bit_cast(qf16x8, park[i]);
        }
    }
    // conv3's first W3 chunk: requested BEFORE the three atomics below (their round trip would sit in front of it in the in-order counter)
    const char* w3h = w3f;
    W3Half wreg = g_w3_load(w3h, 0, wave, lane);
    const bool any_bad = __any(badpt);
    if (lane == 0) {              // per-tile maxima; non-negative floats (and NaN, above all of them) order as integers
        const float dmx = sqrtf(dn2), rdm = sqrtf(rd2);
        const float hm = (dmx + cnorm) * 1.0001f;          // |h_p| <= |h_p - c| + |c|
        wst[wave] = hm; wst[4 + wave] = dmx; wst[8 + wave] = rdm;
        if (any_bad) atomicMax(tstat + 4 * rec + 3, 1u);
        atomicMax(tstat + 4 * rec + 0, __float_as_uint(hm));
        atomicMax(tstat + 4 * rec + 1, __float_as_uint(dmx));
        atomicMax(tstat + 4 * rec + 2, __float_as_uint(rdm));
    }
    dvq_lds_barrier();                                      // everybody is done with W2 in the stages; scs / wst visible
    g_w3_store(fl, wave, lane, wreg);
    // the tile's maxima (uniform) for 2 E
    const float t_hm = fmaxf(fmaxf(wst[0], wst[1]), fmaxf(wst[2], wst[3])) * 1.00001f;
    const float t_dmx = fmaxf(fmaxf(wst[4], wst[5]), fmaxf(wst[6], wst[7])) * 1.00001f;
    const float t_rdm = fmaxf(fmaxf(wst[8], wst[9]), fmaxf(wst[10], wst[11])) * 1.00001f;
    const float* tinv_g = reinterpret_cast<const float*>(w3f + IMG_OFF_TI);
    const float* wnorm_g = reinterpret_cast<const float*>(w3f + IMG_OFF_WN);
    const float* rnorm_g = reinterpret_cast<const float*>(w3f + IMG_OFF_RN);
    float p_ti = 0.f, p_wn = 0.f, p_rn = 0.f;             // of the channel this lane publishes next (requested a chunk ahead)

    // publish the eight small chunks cbase .. cbase + 7: wave w the two chunks cbase + 2 w, + 2 w + 1 = channels n0 .. n0 + 63
    auto publish = [&](int cbase) {
        const int n = 32 * cbase + 64 * wave + lane;
        const float* src = tb + ((cbase + 2 * wave + (lane >> 5)) & 7) * G_SLOT + (lane & 31);
        float t1[16], t2[16];
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) {                 // group gi = 4 w + 2 pb + hh
            t1[gi] = src[gi * 64];
            t2[gi] = src[gi * 64 + 32];
        }
        // run-time check of the hand-over (pn_trunk_filter_kernel): finish() stamps bits [6:5] with (chunk / 8) mod 4
        const unsigned want = (unsigned)(((cbase >> 3) & 3) << 5);
        unsigned tagdiff = 0;
#pragma unroll
        for (int gi = 0; gi < 16; gi += 2)
            tagdiff = tagdiff | (__float_as_uint(t1[gi]) ^ want) | (__float_as_uint(t1[gi]) ^ __float_as_uint(t1[gi + 1]));
        const bool suspect = (tagdiff & 0x60u) != 0;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) {
            const float sc = scs[gi >> 1] * p_ti;          // 1 / (scale of wave gi >> 2's point block (gi >> 1) & 1) / (channel scale): a power of two
            const unsigned tag = (unsigned)((gi >> 2) << 6) | (unsigned)((gi & 1) << 5);
            t1[gi] = __uint_as_float((__float_as_uint(t1[gi] * sc) & ~0xE0u) | tag);
            t2[gi] = __uint_as_float((__float_as_uint(t2[gi] * sc) & ~0xE0u) | tag);
        }
        const float e2 = 2.0f * fmaf(p_rn, t_dmx, fmaf(p_wn, t_rdm, fmaf(C_ID * p_wn, t_dmx, 2.0f * DELTA * p_wn * t_hm)));
        float c1 = NEG_BIG, c2 = NEG_BIG, c3 = NEG_BIG, c4 = NEG_BIG, c5 = NEG_BIG;
#pragma unroll
        for (int gi = 0; gi < 16; ++gi) {
            float x = t1[gi];
            c5 = __builtin_amdgcn_fmed3f(c4, c5, x); c4 = __builtin_amdgcn_fmed3f(c3, c4, x);
            c3 = __builtin_amdgcn_fmed3f(c2, c3, x); c2 = __builtin_amdgcn_fmed3f(c1, c2, x); c1 = max_nc(c1, x);
            x = t2[gi];
            c5 = __builtin_amdgcn_fmed3f(c4, c5, x); c4 = __builtin_amdgcn_fmed3f(c3, c4, x);
            c3 = __builtin_amdgcn_fmed3f(c2, c3, x); c2 = __builtin_amdgcn_fmed3f(c1, c2, x);
        }
        const float thr = c1 - e2;
        unsigned flags = 0;
#pragma unroll
        for (int gi = 15; gi >= 0; --gi) {
            const float u = t1[gi] < c5 ? t1[gi] : t2[gi];
            flags = flags + flags + (unsigned)(u >= thr);
        }
        if (suspect) flags = 0x1FFFFu;
        if ((abl & 32768) && n == 327) c1 = fabsf(c1) * 1.0e3f + 1.0f;   // diagnostics: a record that lies about its tile
        part[rec * 1024 + n] = f32x4{c1, c2, c3, __uint_as_float(flags)};
        part2[rec * 1024 + n] = qf32x2{c4, c5};
    };
    auto request_norms = [&](int cbase) {                 // for publish(cbase)
        const int n = 32 * cbase + 64 * wave + lane;
        p_ti = tinv_g[n]; p_wn = wnorm_g[n]; p_rn = rnorm_g[n];
    };
#define G_MFMA_BLOCK(ACC, PB)                                                                                  \
    do {                                                                                                       \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) ACC[e] = 0.f;                                           \
        _Pragma("unroll") for (int s = 0; s < 8; ++s)                                                          \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[PB][s], wf[s], ACC, 0, 0, 0);                      \
    } while (0)
#define G_CHAIN_BLOCK(ACC, PB, M1, M2)                                                                         \
    do {                                                                                                       \
        M1 = NEG_BIG; M2 = NEG_BIG;                                                                            \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                       \
            const float x = __uint_as_float((__float_as_uint(ACC[e]) & ~31u) | (unsigned)(16 * (PB) + e));     \
            M2 = __builtin_amdgcn_fmed3f(M1, M2, x);                                                           \
            M1 = max_nc(M1, x);                                                                                \
        }                                                                                                      \
    } while (0)
#define G_INTERLEAVE()                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);                                                 \
        }                                                                                                      \
    } while (0)
    auto finish = [&](int c, int pb, float m1, float m2) {
        if ((abl & 65536) && c == 18 && wave == 2) return;   // diagnostics: a hand-over that does not happen
        float* dst = tb + (c & 7) * G_SLOT + ((wave * 2 + pb) * 2 + h) * 64 + r;
        const unsigned ctag = (unsigned)(((c >> 3) & 3) << 5);
        dst[0] = __uint_as_float((__float_as_uint(m1) & ~0x60u) | ctag);   // the pair is ONE LDS store: the tag of its first value covers both
        dst[32] = m2;
    };
    f32x16 accP;                                          // the chunk's second block, scored under the next chunk's first MFMAs
    // (written out eight chunks per trip -- ring slot, stage and conditions as constants, what pays 1.5 % in the two-per-CU kernel --
    // this loop spills 33 registers instead of 14 and the kernel loses 6 %: it stays rolled)
    bool pending = false;
#pragma unroll 1
    for (int c = 0; c < 32; ++c) {
        dvq_lds_barrier();                                  // chunk c is in its stage; the other stage and ring slot c & 7 are free
        if (c + 1 < 32) wreg = g_w3_load(w3h, 32 * (c + 1), wave, lane);
        if ((c & 7) == 0 && c > 0) {                      // the ring's eight chunks are complete (barrier above): two per wave
            publish(c - 8);
            dvq_lds_barrier();                              // before this chunk's pairs overwrite slot 0
        }
        if ((c & 7) == 6) request_norms(c - 6);            // for the publish two chunks from now
        const char* st = fl + (c & 1) * G_STAGE;
        qf16x8 wf[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[s] = w3_frag(st, r, 2 * s + h);
        f32x16 accA;
        float m1, m2;
        G_MFMA_BLOCK(accA, 0);
        if (pending) {
            G_CHAIN_BLOCK(accP, 1, m1, m2);
            G_INTERLEAVE();
            finish(c - 1, 1, m1, m2);
        }
        G_MFMA_BLOCK(accP, 1);
        G_CHAIN_BLOCK(accA, 0, m1, m2);
        G_INTERLEAVE();
        finish(c, 0, m1, m2);
        pending = (c & 7) != 7;
        if (!pending) {
            G_CHAIN_BLOCK(accP, 1, m1, m2);
            finish(c, 1, m1, m2);
        }
        if (c + 1 < 32) g_w3_store(fl + ((c + 1) & 1) * G_STAGE, wave, lane, wreg);
    }
#undef G_MFMA_BLOCK
#undef G_CHAIN_BLOCK
#undef G_INTERLEAVE
    dvq_lds_barrier();
    publish(24);
}

#endif  // DVQ_DIAG (pn_trunk3_kernel)

// ------------------------------------------------------------------------------------------------------------------------
// exact_dot: THE definition of a conv3 score on this path.  16 lanes per dot, lane j owns k = 4j .. 4j+3 and 64+4j .. 64+4j+3 (fixed
// FMA order), then a 16-lane butterfly (every lane gets the same bits).  (The two 16-byte pieces of a lane are 256 B apart so that the
// sixteen lanes of a load read 256 CONTIGUOUS bytes: with k = 8j .. 8j+7 -- until round 5 -- every load touched four cache lines
// and used half of each, and the exact stage's candidate dots ran at what the CU's vector memory path gives to such gathers.)
template <int CTRL>
__device__ __forceinline__ float q_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float exact_dot_regs(const f32x4& w0, const f32x4& w1, const f32x4& a, const f32x4& c) {
    float s = w0[0] * a[0];
    s = fmaf(w0[1], a[1], s);
    s = fmaf(w0[2], a[2], s);
    s = fmaf(w0[3], a[3], s);
    s = fmaf(w1[0], c[0], s);
    s = fmaf(w1[1], c[1], s);
    s = fmaf(w1[2], c[2], s);
    s = fmaf(w1[3], c[3], s);
    s += q_dpp<0xB1>(s);
    s += q_dpp<0x4E>(s);
    s += q_dpp<0x141>(s);
    s += q_dpp<0x140>(s);
    return s;
}
__device__ __forceinline__ float exact_dot(const f32x4& w0, const f32x4& w1, const float* __restrict__ hrow, int j) {
    return exact_dot_regs(w0, w1, *reinterpret_cast<const f32x4*>(hrow + 4 * j), *reinterpret_cast<const f32x4*>(hrow + 4 * j + 64));
}
// torch.max semantics: a NaN wins
__device__ __forceinline__ float max_nan(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : fmaxf(a, b); }

// order-preserving map float -> unsigned (for atomicMax on LDS)
__device__ __forceinline__ unsigned f2key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// Run-time consistency counters (dvq_pointnet_fault_counters): [0] tile records the trunk kernel marked suspect (a ring value whose
// chunk tag is not the published chunk's: stale or missing input of the merge), [1] channels whose exact maximum lies outside the
// interval the tile records promise (max_t (top_t - E_t) <= max - w.c <= max_t (top_t + E_t)).  Either way the channel is evaluated
// over ALL the points concerned, so the feature is right; a non-zero counter says the filter's bookkeeping was not.
__device__ unsigned long long g_pn_faults[2];

// One workgroup per sample, 16 groups of 16 lanes.
//   phase A (one thread per channel): best lower bound over the tiles; every kept score whose upper bound reaches it
//           becomes a candidate point of the channel (table of 4 per channel, the rest in a list); flagged 16-point groups of tiles
//           in contention become (channel, tile, group) entries;
//   phase B (one group per channel): the weight row once, its candidates' rows together, exact_dot, maximum;
//   phase C (one wave per entry, no barrier): the 16 points of a flagged group; then, whole workgroup per channel, every
//           point for the channels on the "everything" list (DVQ_PN_EXHAUSTIVE / non-finite inputs).
// stats (optional): channels with one candidate, with another count, wave entries, candidates.
constexpr int PAIR_CAP = 1024;
constexpr int FB_CAP = 512;
constexpr int SORT_CAP = 3072;                            // (channel, point) pairs evaluated in point order; more: channel order
__global__ __launch_bounds__(256, 4) void pn_exact_kernel(const f32x4* __restrict__ part, const qf32x2* __restrict__ part2, int tiles, int deal,
                                                       const float* __restrict__ h2buf,
                                                       int N, int Npad, const float* __restrict__ w3, const float* __restrict__ b3,
                                                       const float* __restrict__ wnorm, const float* __restrict__ rnorm,
                                                       const unsigned* __restrict__ tstat, const float* __restrict__ cbuf, int relu, int exhaustive,
                                                       int pair_cap, int fb_cap, float* __restrict__ feat, long ld_feat,
                                                       unsigned long long* __restrict__ stats, int abl_arg) {
    const int abl = DVQ_DIAG_ON ? abl_arg : 0;
    __shared__ unsigned short cand[1024][4];
    __shared__ unsigned char cand_n[1024];
    __shared__ int pair_list[PAIR_CAP];
    __shared__ int fb_list[FB_CAP];
    __shared__ short all_list[1024];
    __shared__ int pair_count, fb_count, all_count;
    __shared__ float fb_part[4][16];
    __shared__ float hm[MAX_TILES], dm[MAX_TILES], rd[MAX_TILES];
    __shared__ unsigned best_k[1024];
    __shared__ int pcnt[1024];                             // pairs per point -> first slot of the point -> fill cursor
    __shared__ unsigned sorted[SORT_CAP];                  // channel | point << 10, grouped by point
    __shared__ int wave_tot[4];
    const int tid = threadIdx.x, g = tid >> 4, j = tid & 15;
    const long b = blockIdx.x;
    float* wcs = reinterpret_cast<float*>(pcnt);           // w_n . c per channel (consistency check): pcnt is dead once the pairs are sorted
    const f32x4 cen_a = *reinterpret_cast<const f32x4*>(cbuf + b * 128 + 4 * j), cen_b = *reinterpret_cast<const f32x4*>(cbuf + b * 128 + 4 * j + 64);
    if (tid == 0) { pair_count = 0; fb_count = 0; all_count = 0; }
#pragma unroll
    for (int i = 0; i < 4; ++i) pcnt[tid + 256 * i] = 0;
    int nonfinite_point = 0;
    if (tid < tiles) {
        const unsigned* ts = tstat + 4 * (b * tiles + tid);
        hm[tid] = __uint_as_float(ts[0]) * 1.00001f;
        dm[tid] = __uint_as_float(ts[1]) * 1.00001f;
        rd[tid] = __uint_as_float(ts[2]) * 1.00001f;
        nonfinite_point = ts[3] != 0 && !(abl & 4096);      // (word 3 holds the phase stamps of the diagnostics build otherwise)
    }
    if (__syncthreads_or(nonfinite_point)) {
        // a cloud with a NaN / Inf coordinate: NaN in every channel, as the reference's affine layers and torch.max make it (the trunk
        // kernel's ReLU squashed it; see there)
        for (int n = tid; n < 1024; n += 256) feat[b * ld_feat + n] = __builtin_nanf("");
        return;
    }
    const float* h2 = h2buf + b * (long)Npad * 128;
    const bool wrap_small = Npad <= 2 * N;                  // a padding slot's index is below 2 N: one subtraction instead of a division
    const f32x4* pt = part + b * (long)tiles * 1024;
    const qf32x2* pt2 = part2 + b * (long)tiles * 1024;     // the fourth and fifth id-carrying scores: read only where the third is in range
    // ---- phase A
    const bool stamps = DVQ_DIAG_ON && stats && (abl & 4096);   // diagnostics: cycles per phase (tid 0's clock), summed into stats[4..7]
    unsigned long long tp0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull, tp1 = 0, tp2 = 0, tp3 = 0;
    unsigned n_single = 0, n_multi = 0, n_cand = 0, n_wave = 0, n_suspect = 0;
    // the interval the records promise for (max - w.c) of channels tid + 256 i; lo > hi: not checked.  Eight scalars updated through
    // selects: the channel loop below stays ROLLED (unrolled it was 12 k instructions, 80 KB of code for a 64 KB instruction cache
    // shared by two CUs) without turning an indexed array into scratch memory.
    float lo_0 = 1.f, lo_1 = 1.f, lo_2 = 1.f, lo_3 = 1.f, hi_0 = 0.f, hi_1 = 0.f, hi_2 = 0.f, hi_3 = 0.f;
    // the next channel's records (the first four tiles' entries -- N <= 1024: all of them) and norms are requested before the current
    // channel is worked on: one exposed trip to memory instead of four
    f32x4 nf0 = pt[tid], nf1 = pt[min(1, tiles - 1) * 1024 + tid], nf2 = pt[min(2, tiles - 1) * 1024 + tid], nf3 = pt[min(3, tiles - 1) * 1024 + tid];
    // (the fourth and fifth scores with them: read where the third is in range, that was a trip to memory inside the tile loop whenever
    // ONE lane of the wave needed it -- most iterations)
    qf32x2 ng0 = pt2[tid], ng1 = pt2[min(1, tiles - 1) * 1024 + tid], ng2 = pt2[min(2, tiles - 1) * 1024 + tid], ng3 = pt2[min(3, tiles - 1) * 1024 + tid];
    float nwn = wnorm[tid], nrn = rnorm[tid];
#pragma unroll 1
    for (int ci = 0; ci < 4; ++ci) {
        const int n = tid + 256 * ci;
        best_k[n] = f2key(NEG_BIG);
        const float wn = nwn, rn = nrn;
        const f32x4 f0 = nf0, f1 = nf1, f2 = nf2, f3 = nf3;
        const qf32x2 g0 = ng0, g1 = ng1, g2 = ng2, g3 = ng3;
        if (ci < 3) {
            const int nx = n + 256;
            nf0 = pt[nx]; nf1 = pt[min(1, tiles - 1) * 1024 + nx]; nf2 = pt[min(2, tiles - 1) * 1024 + nx]; nf3 = pt[min(3, tiles - 1) * 1024 + nx];
            ng0 = pt2[nx]; ng1 = pt2[min(1, tiles - 1) * 1024 + nx]; ng2 = pt2[min(2, tiles - 1) * 1024 + nx]; ng3 = pt2[min(3, tiles - 1) * 1024 + nx];
            nwn = wnorm[nx]; nrn = rnorm[nx];
        }
        float lb, e_all;
        auto bound = [&](int t) { return fmaf(rn, dm[t], fmaf(wn, rd[t], fmaf(C_ID * wn, dm[t], 2.0f * DELTA * wn * hm[t]))); };
        // (tiles beyond the first four: in blocks of four, below)
        // a non-finite bound or top score in ANY tile sends the channel to the "everything" path (fmaxf drops a NaN: tested apart)
        bool nonfinite = false;
        float ub;
        auto fold = [&](float e, float top) {
            nonfinite = nonfinite || !(e < 3.0e38f) || !(fabsf(top) < 3.0e38f);
            e_all = fmaxf(e_all, e);
            lb = fmaxf(lb, top - e);
            ub = fmaxf(ub, top + e);
        };
        {
            const float e0 = bound(0);
            e_all = e0;
            lb = f0[0] - e0;
            ub = f0[0] + e0;
            nonfinite = !(e0 < 3.0e38f) || !(fabsf(f0[0]) < 3.0e38f);
            if (tiles > 1) fold(bound(1), f1[0]);
            if (tiles > 2) fold(bound(2), f2[0]);
            if (tiles > 3) fold(bound(3), f3[0]);
        }
        for (int t0 = 4; t0 < tiles; t0 += 4) {
            float top[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) top[u] = pt[min(t0 + u, tiles - 1) * 1024 + n][0];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < tiles) fold(bound(t0 + u), top[u]);
        }
        int cands = 0;
        bool whole = false;
        const bool all = exhaustive || nonfinite || !(e_all < 3.0e38f) || !(lb > NEG_BIG) || !(lb < 3.0e38f);   // non-finite inputs: evaluate everything
        if (all) {
            all_list[atomicAdd(&all_count, 1)] = (short)n;
            cand_n[n] = 0;
            continue;
        }
        float lo_c = lb, hi_c = ub;
        auto consider = [&](int t, const f32x4& q, const qf32x2& q45) {
            const float et = bound(t);
            const unsigned suspect = (__float_as_uint(q[3]) >> 16) & 1u;   // the trunk kernel did not trust its own merge: it flagged every group
            n_suspect += suspect;
            // the tile's largest score is out of range: so is the rest of it -- unless the record is suspect: then its scores prove
            // nothing about the tile and all its points are evaluated (a bogus top score that RAISES lb is caught by the check below)
            if (!suspect && !(q[0] + et >= lb)) return;
            unsigned flags = __float_as_uint(q[3]) & 0xFFFFu;   // one bit per 16-point group: 4 wave + 2 point block + lane half
            // a kept score in range: its point becomes a candidate of the channel.  (No search for a point that is already one: only
            // padding slots repeat a point, a repeated candidate costs one more dot, and the search was a third of this phase.)
            auto take = [&](float v) {
                int p = point_of_slot(t, slot_of_id(__float_as_uint(v) & 255u), deal);
                if (p >= N) p = wrap_small ? p - N : p % N;  // a padding slot: the real point it repeats
                if (abl & 16) p &= 63;
                if (cands < 4) cand[n][cands] = (unsigned short)p;
                else {
                    const int slot = atomicAdd(&pair_count, 1);
                    if (slot < pair_cap) pair_list[slot] = n | (p << 10);
                    else {                                       // list full (never seen): evaluate its 16-point group instead
                        const unsigned id = __float_as_uint(v);
                        flags |= 1u << (4 * ((id >> 6) & 3u) + 2 * ((id >> 4) & 1u) + ((id >> 5) & 1u));
                    }
                }
                ++cands;
            };
            // descending scores: the ones in range are a prefix of (c1 .. c5)
            const float sc5[5] = {q[0], q[1], q[2], q45[0], q45[1]};
#pragma unroll 1
            for (int k = 0; k < 5; ++k) {
                const float v = k == 0 ? sc5[0] : k == 1 ? sc5[1] : k == 2 ? sc5[2] : k == 3 ? sc5[3] : sc5[4];
                if (!(v + et >= lb)) break;
                take(v);
            }
            while (flags) {
                const int wh = __ffs(flags) - 1;             // 4 * wave + 2 * point block + lane half
                flags &= flags - 1;
                const int slot = atomicAdd(&fb_count, 1);
                if (slot < fb_cap) fb_list[slot] = n | (t << 10) | (wh << 20);
                else if (!whole) {                           // list full (never seen): the channel goes on the "everything" list
                    whole = true;
                    all_list[atomicAdd(&all_count, 1)] = (short)n;
                }
                ++n_wave;
            }
        };
        // one copy of the code above for every tile (rolled: see lo_0 .. hi_3); tiles beyond the first four are loaded in blocks of four
        f32x4 q0 = f0, q1 = f1, q2 = f2, q3 = f3;
        qf32x2 h0 = g0, h1 = g1, h2q = g2, h3 = g3;
#pragma unroll 1
        for (int t0 = 0; t0 < tiles; t0 += 4) {
            if (t0 > 0) {
                q0 = pt[min(t0, tiles - 1) * 1024 + n]; q1 = pt[min(t0 + 1, tiles - 1) * 1024 + n];
                q2 = pt[min(t0 + 2, tiles - 1) * 1024 + n]; q3 = pt[min(t0 + 3, tiles - 1) * 1024 + n];
                h0 = pt2[min(t0, tiles - 1) * 1024 + n]; h1 = pt2[min(t0 + 1, tiles - 1) * 1024 + n];
                h2q = pt2[min(t0 + 2, tiles - 1) * 1024 + n]; h3 = pt2[min(t0 + 3, tiles - 1) * 1024 + n];
            }
#pragma unroll 1
            for (int u = 0; u < 4 && t0 + u < tiles; ++u) {
                const f32x4 q = u == 0 ? q0 : u == 1 ? q1 : u == 2 ? q2 : q3;
                const qf32x2 q45 = u == 0 ? h0 : u == 1 ? h1 : u == 2 ? h2q : h3;
                consider(t0 + u, q, q45);
            }
        }
        cand_n[n] = whole ? 0 : (unsigned char)min(cands, 4);
        if (whole) { lo_c = 1.f; hi_c = 0.f; }              // evaluated in full below: nothing to check
        lo_0 = ci == 0 ? lo_c : lo_0; hi_0 = ci == 0 ? hi_c : hi_0;
        lo_1 = ci == 1 ? lo_c : lo_1; hi_1 = ci == 1 ? hi_c : hi_1;
        lo_2 = ci == 2 ? lo_c : lo_2; hi_2 = ci == 2 ? hi_c : hi_2;
        lo_3 = ci == 3 ? lo_c : lo_3; hi_3 = ci == 3 ? hi_c : hi_3;
        n_single += cands == 1;
        n_multi += cands != 1;
        n_cand += cands;
    }
    dvq_lds_barrier();
    if (stamps) tp1 = __builtin_amdgcn_s_memtime();
    // ---- phase A2: the pairs in POINT order (counting sort in the LDS).  A cloud's 1 024 channels take their maxima at ~100-200
    // distinct points, so ~1 500 candidate pairs name each conv2 row ~10 times: evaluated channel by channel every pair fetched
    // its 512-byte row from HBM again (0.74 MB per cloud, the kernel ran at the HBM roofline); grouped by point a row is
    // fetched once and found in the L1 by the pairs that follow.
    const int npairs = min(pair_count, pair_cap);
    for (int n = tid; n < 1024; n += 256)
        for (int k = 0; k < cand_n[n]; ++k) atomicAdd(&pcnt[cand[n][k] & 1023], 1);     // N > 1024: points 1024 apart share a slot range
    for (int i = tid; i < npairs; i += 256) atomicAdd(&pcnt[(pair_list[i] >> 10) & 1023], 1);
    dvq_lds_barrier();
    int total;
    {
        const int c0 = pcnt[4 * tid], c1 = pcnt[4 * tid + 1], c2 = pcnt[4 * tid + 2], c3 = pcnt[4 * tid + 3];
        int incl = c0 + c1 + c2 + c3;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        dvq_lds_barrier();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += wave_tot[w];
        total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        const int excl = base + incl - (c0 + c1 + c2 + c3);
        pcnt[4 * tid] = excl;
        pcnt[4 * tid + 1] = excl + c0;
        pcnt[4 * tid + 2] = excl + c0 + c1;
        pcnt[4 * tid + 3] = excl + c0 + c1 + c2;
    }
    dvq_lds_barrier();
    const bool by_point = total <= SORT_CAP && !(abl & 64);
    if (by_point) {
        for (int n = tid; n < 1024; n += 256)
            for (int k = 0; k < cand_n[n]; ++k) {
                const int p = cand[n][k];
                sorted[atomicAdd(&pcnt[p & 1023], 1)] = (unsigned)n | ((unsigned)p << 10);
            }
        for (int i = tid; i < npairs; i += 256) {
            const int code = pair_list[i];
            sorted[atomicAdd(&pcnt[(code >> 10) & 1023], 1)] = (unsigned)code;
        }
        dvq_lds_barrier();
        if (stamps) tp2 = __builtin_amdgcn_s_memtime();
        // ---- phase B, point order: every 16-lane group takes a contiguous share of the list, four pairs in flight
        const int per = (total + 15) >> 4, i0 = g * per, i1 = min(total, i0 + per);
        for (int i = i0; i < i1; i += 4) {
            f32x4 w0[4], w1[4], ha[4], hb[4];
            int nn[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned code = sorted[min(i + u, i1 - 1)];
                nn[u] = (int)(code & 1023u);
                // (uniform base + 32-bit offset: the loads take the base from scalar registers, no 64-bit vector add per row)
                const unsigned woff = (code & 1023u) * 512u + 16u * (unsigned)j, hoff = (code >> 10) * 512u + 16u * (unsigned)j;
                w0[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(w3) + woff);
                w1[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(w3) + woff + 256);
                ha[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(h2) + hoff);
                hb[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(h2) + hoff + 256);
            }
            float v[4], wc[4];                              // all eight chains first (independent: they interleave), the LDS updates after
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = exact_dot_regs(w0[u], w1[u], ha[u], hb[u]);
                wc[u] = exact_dot_regs(w0[u], w1[u], cen_a, cen_b);             // the centre term of this channel (check below)
            }
            if (j == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i + u < i1) { atomicMax(&best_k[nn[u]], f2key(v[u])); wcs[nn[u]] = wc[u]; }
            }
        }
    }
    // ---- phase B, channel order (more pairs than the sorted list holds): table; four channels of a group in flight (first
    // candidates), further candidates afterwards
    for (int n0 = g; n0 < ((abl & 64) || by_point ? 0 : 1024); n0 += 64) {
        f32x4 w0[4], w1[4], ha[4], hb[4];
        int cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + 16 * u;
            cn[u] = cand_n[n];
            const float* wr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w3) + ((unsigned)n * 512u + 16u * (unsigned)j));
            w0[u] = *reinterpret_cast<const f32x4*>(wr);
            w1[u] = *reinterpret_cast<const f32x4*>(wr + 64);
            const float* hr = h2 + (long)(cn[u] ? cand[n][0] : 0) * 128 + 4 * j;
            ha[u] = *reinterpret_cast<const f32x4*>(hr);
            hb[u] = *reinterpret_cast<const f32x4*>(hr + 64);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + 16 * u;
            if (cn[u] == 0) continue;
            float best = exact_dot_regs(w0[u], w1[u], ha[u], hb[u]);
            const float wc = exact_dot_regs(w0[u], w1[u], cen_a, cen_b);
            if (j == 0) wcs[n] = wc;
            if (cn[u] > 1) {
                f32x4 xa[3], xb[3];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k < cn[u]) {
                        const float* hr = h2 + (long)cand[n][k] * 128 + 4 * j;
                        xa[k - 1] = *reinterpret_cast<const f32x4*>(hr);
                        xb[k - 1] = *reinterpret_cast<const f32x4*>(hr + 64);
                    }
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k < cn[u]) best = fmaxf(best, exact_dot_regs(w0[u], w1[u], xa[k - 1], xb[k - 1]));
            }
            if (j == 0) atomicMax(&best_k[n], f2key(best));
        }
    }
    // ---- phase B, channel order: overflow list
    for (int i = g; i < (by_point ? 0 : npairs); i += 16) {
        const int code = pair_list[i];
        const int n = code & 1023;
        const float* wr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w3) + ((unsigned)n * 512u + 16u * (unsigned)j));
        const float v = exact_dot(*reinterpret_cast<const f32x4*>(wr), *reinterpret_cast<const f32x4*>(wr + 64), h2 + (long)(code >> 10) * 128, j);
        if (j == 0) atomicMax(&best_k[n], f2key(v));
    }
    if (stamps) tp3 = __builtin_amdgcn_s_memtime();
    // ---- phase C: flagged 16-point groups, one wave of the workgroup per entry, its four 16-lane groups take 4 points each
    const int nfb = (abl & 32) ? 0 : min(fb_count, fb_cap);
    for (int i = tid >> 6; i < nfb; i += 4) {
        const int code = fb_list[i];
        const int n = code & 1023, t = (code >> 10) & 1023, w = (code >> 22) & 3, pb = (code >> 21) & 1, hh = (code >> 20) & 1;
        const float* wr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w3) + ((unsigned)n * 512u + 16u * (unsigned)j));
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 64);
        float best = NEG_BIG;
        {
            f32x4 ha[4], hb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                   // this lane group's 4 of the group's 16 points (accumulator registers e)
                const int e = 4 * (g & 3) + u;
                int p = point_of_slot(t, 64 * w + 32 * pb + 8 * (e >> 2) + 4 * hh + (e & 3), deal);   // tail tile: one block, pb = 0
                if (p >= N) p %= N;
                const float* hr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(h2) + ((unsigned)p * 512u + 16u * (unsigned)j));
                ha[u] = *reinterpret_cast<const f32x4*>(hr);
                hb[u] = *reinterpret_cast<const f32x4*>(hr + 64);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) best = fmaxf(best, exact_dot_regs(w0, w1, ha[u], hb[u]));
        }
        if (j == 0) atomicMax(&best_k[n], f2key(best));
    }
    dvq_lds_barrier();
    // ---- phase C: everything (NaN-propagating maximum, torch.max semantics): the channels on all_list over ALL points, four channels
    // per sweep of the rows (a row's slice is loaded once for the four)
    auto eval_all_list = [&](int count) {
        for (int i = 0; i < count; i += 4) {
            f32x4 w0[4], w1[4];
            float best[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = all_list[min(i + u, count - 1)];
                w0[u] = *reinterpret_cast<const f32x4*>(w3 + n * 128 + 4 * j);
                w1[u] = *reinterpret_cast<const f32x4*>(w3 + n * 128 + 4 * j + 64);
                best[u] = NEG_BIG;
            }
            for (int p = g; p < N; p += 16) {
                const float* hr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(h2) + ((unsigned)p * 512u + 16u * (unsigned)j));
                const f32x4 ha = *reinterpret_cast<const f32x4*>(hr), hb = *reinterpret_cast<const f32x4*>(hr + 64);
#pragma unroll
                for (int u = 0; u < 4; ++u) best[u] = max_nan(best[u], exact_dot_regs(w0[u], w1[u], ha, hb));
                // a NaN stays a NaN (torch.max): four NaN maxima need no more points.  A cloud that is non-finite as a whole -- the row
                // of a grasp whose decoder output left fp16's range, on its way to the per-row fallback of GenNet.gen -- sent ONE
                // workgroup through 1 024 channels x all points, the straggler of its launch (20 ms at the benchmark's batch).
                if (best[0] != best[0] && best[1] != best[1] && best[2] != best[2] && best[3] != best[3]) break;
            }
            if (j == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) fb_part[u][g] = best[u];
            }
            dvq_lds_barrier();
            if (tid < 4 && i + tid < count) {
                float v = fb_part[tid][0];
                for (int k = 1; k < 16; ++k) v = max_nan(v, fb_part[tid][k]);
                best_k[all_list[i + tid]] = (v != v) ? 0xffffffffu : f2key(v);       // NaN: the largest key, decoded below
            }
            dvq_lds_barrier();
        }
    };
    eval_all_list(all_count);
    // ---- consistency: the exact maximum of a channel must lie where its tile records said it would.  |approx + w.c - exact| <= E_t
    // for every point of tile t, so  max_t (top_t - E_t) <= max - w.c <= max_t (top_t + E_t).  A maximum outside that interval means a
    // record did not describe its tile (a wrong score or id; a missing input that was the tile's best shows up in the trunk kernel's
    // own tag check instead): such a channel is evaluated over all points, and counted.
    dvq_lds_barrier();                                       // best_k / wcs complete; all_list free again
    if (tid == 0) all_count = 0;
    dvq_lds_barrier();
    unsigned n_bad = 0;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
        const int n = tid + 256 * ci;
        const float lo_c = ci == 0 ? lo_0 : ci == 1 ? lo_1 : ci == 2 ? lo_2 : lo_3, hi_c = ci == 0 ? hi_0 : ci == 1 ? hi_1 : ci == 2 ? hi_2 : hi_3;
        if (!(lo_c <= hi_c) || (abl & ~(4096 | 8192 | 32768 | 65536 | 524288))) continue;   // (the timing ablations -- of either kernel -- leave maxima that are not maxima)
        const float v = key2f(best_k[n]), wc = wcs[n];
        const float x = v - wc, slack = 4.0e-7f * (fabsf(v) + fabsf(wc));   // the subtraction's own rounding
        if (!(x >= lo_c - slack && x <= hi_c + slack)) {
            all_list[atomicAdd(&all_count, 1)] = (short)n;
            ++n_bad;
        }
    }
    dvq_lds_barrier();
    eval_all_list(all_count);
    if (n_suspect) atomicAdd(&g_pn_faults[0], (unsigned long long)n_suspect);
    if (n_bad) atomicAdd(&g_pn_faults[1], (unsigned long long)n_bad);
    for (int n = tid; n < 1024; n += 256) {
        const unsigned k = best_k[n];
        const float v = (k == 0xffffffffu ? __builtin_nanf("") : key2f(k)) + b3[n];
        feat[b * ld_feat + n] = relu ? (v != v ? v : fmaxf(v, 0.f)) : v;
    }
    if (stamps && tid == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicAdd(stats + 4, tp1 - tp0); atomicAdd(stats + 5, tp2 - tp1); atomicAdd(stats + 6, tp3 - tp2); atomicAdd(stats + 7, t_end - tp3);
    }
    if (stats) {
        if (n_single) atomicAdd(stats + 0, (unsigned long long)n_single);
        if (n_multi) atomicAdd(stats + 1, (unsigned long long)n_multi);
        if (n_wave) atomicAdd(stats + 2, (unsigned long long)n_wave);
        if (n_cand) atomicAdd(stats + 3, (unsigned long long)n_cand);
    }
}

// Centre of a sample: mean of the conv2 rows of the points 0, N/4, N/2, 3N/4, plain fp32 (any vector would do -- it shifts
// every score of a channel by the same w.c -- but one close to the rows makes the fp16 residuals, hence the bounds, small).
// One workgroup takes CENTER_SPB samples: W2 (32 KB) is staged in the LDS once (transposed, conflict-free) for all of them.
constexpr int CENTER_SPB = 4;                             // (8: 80 us per launch of 3 641 samples, 4: 61, 2: 64, 1: 65 -- round 6)
template <int C>
__global__ __launch_bounds__(256) void pn_center_kernel(const float* __restrict__ pc, const float* __restrict__ trans, int N, long B,
                                                        const float* __restrict__ W1, const float* __restrict__ b1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2,
                                                        float* __restrict__ cbuf, unsigned* __restrict__ tstat, int tiles) {
    __shared__ float w2t[64][129];                         // w2t[k][ch] = W2[ch][k] (row stride 129: conflict-free both ways)
    __shared__ float h1c[4][64];
    __shared__ float h2c[4][128];
    const int tid = threadIdx.x, q = tid >> 6, k = tid & 63;
    for (int i = tid; i < 128 * 64; i += 256) w2t[i & 63][i >> 6] = W2[i];
    for (int sb = 0; sb < CENTER_SPB; ++sb) {
        const long b = (long)blockIdx.x * CENTER_SPB + sb;
        if (b >= B) break;
        for (int i = tid; i < 4 * tiles; i += 256) tstat[b * 4 * tiles + i] = 0u;   // the sample's tile records start from zero (the trunk kernel's atomicMax targets): no memset launch
        {
            const int p = (int)(((long)q * N) / 4);
            const float* src = pc + b * (long)C * N + p;
            float x0 = src[0], x1 = src[N], x2 = src[2L * N];
            const float x3 = (C > 3) ? src[3L * N] : 0.f;
            if (trans) {
                const float* t = trans + b * 9;
                const float n0 = fmaf(x2, t[6], fmaf(x1, t[3], x0 * t[0]));
                const float n1 = fmaf(x2, t[7], fmaf(x1, t[4], x0 * t[1]));
                const float n2 = fmaf(x2, t[8], fmaf(x1, t[5], x0 * t[2]));
                x0 = n0; x1 = n1; x2 = n2;
            }
            const float* w = W1 + 4 * k;
            float a = x0 * w[0];
            a = fmaf(x1, w[1], a);
            a = fmaf(x2, w[2], a);
            a = fmaf(x3, w[3], a);
            h1c[q][k] = fmaxf(a + b1[k], 0.f);
        }
        dvq_lds_barrier();                                   // h1c (and, the first time, w2t) complete
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int ch = k + 64 * half;
            float a = 0.f;
            for (int i = 0; i < 64; ++i) a = fmaf(w2t[i][ch], h1c[q][i], a);
            h2c[q][ch] = fmaxf(a + b2[ch], 0.f);
        }
        dvq_lds_barrier();
        if (tid < 128) cbuf[b * 128 + tid] = 0.25f * ((h2c[0][tid] + h2c[1][tid]) + (h2c[2][tid] + h2c[3][tid]));
        // the next sample's h1c writes come after this barrier pair: h2c reads above are done before its second barrier
    }
}

// One wave per conv3 output channel: fp16 image (k permuted to conv2's accumulator order, scaled by a power of two so that
// the row maximum lies in [2^14, 2^15)), 1 / scale, |w| rounded up.
__global__ __launch_bounds__(256) void pn_filter_pack_kernel(const float* __restrict__ w3, _Float16* __restrict__ wh,
                                                             float* __restrict__ tinv, float* __restrict__ wnorm,
                                                             float* __restrict__ rnorm) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float a = w3[n * 128 + lane], c = w3[n * 128 + 64 + lane];
    const float amax = wave_max(fmaxf(fabsf(a), fabsf(c)));
    float sq = fmaf(a, a, c * c);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    float t = 1.f;
    const int ex = (int)((__float_as_uint(amax) >> 23) & 255u);
    if (ex > 20 && ex < 235) t = __uint_as_float((unsigned)(127 + 15 - (ex - 126)) << 23);
    // image position pos = 16 blk + q holds channel 16 blk + perm(q), perm = (0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15)
    float rs = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int pos = 64 * half + lane;
        const int q = pos & 15;
        const int k = (pos & ~15) + ((q & 3) | ((q & 4) << 1) | ((q & 8) >> 1));
        const float v = w3[n * 128 + k] * t;
        const _Float16 hv = (_Float16)v;
        wh[n * 128 + pos] = hv;
        const float res = v - (float)hv;
        rs = fmaf(res, res, rs);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rs += __shfl_xor(rs, o);
    if (lane == 0) {
        tinv[n] = 1.0f / t;
        wnorm[n] = sqrtf(sq) * 1.00001f;
        rnorm[n] = sqrtf(rs) / t * 1.00001f;
    }
}

// One wave per conv2 output channel (128 rows of 64): the two fp16 planes of w * 2^t_n (row maximum in [2^14, 2^15)) -- the second the
// remainder as it is (csrc/gemm_f16x2.hip's packer scales it by 2^11 for a second accumulator; conv2 here has one) -- and 2^-t_n.
__global__ __launch_bounds__(256) void pn_filter_pack_w2_kernel(const float* __restrict__ w2, _Float16* __restrict__ planes,
                                                                float* __restrict__ kinv) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float v = w2[n * 64 + lane];
    const float amax = wave_max(fabsf(v));
    float t = 1.f;
    const int ex = (int)((__float_as_uint(amax) >> 23) & 255u);
    if (ex > 20 && ex < 235) t = __uint_as_float((unsigned)(127 + 15 - (ex - 126)) << 23);
    const float vs = v * t;
    const _Float16 p1 = (_Float16)vs;
    const _Float16 p2 = (_Float16)(vs - (float)p1);
    planes[n * 64 + lane] = p1;
    planes[128 * 64 + n * 64 + lane] = p2;
    if (lane == 0) kinv[n] = 1.0f / t;
}

}  // namespace

size_t dvq_pn_filter_image_bytes() { return (size_t)IMG_BYTES; }

int dvq_launch_pn_filter_pack(const float* w2, const float* w3, void* image, hipStream_t st) {
    char* im = (char*)image;
    DVQ_LAUNCH(pn_filter_pack_kernel, dim3(256), dim3(256), 0, st, w3, reinterpret_cast<_Float16*>(im),
               reinterpret_cast<float*>(im + IMG_OFF_TI), reinterpret_cast<float*>(im + IMG_OFF_WN),
               reinterpret_cast<float*>(im + IMG_OFF_RN));
    DVQ_CHECK_LAUNCH("pn_filter_pack");
    DVQ_LAUNCH(pn_filter_pack_w2_kernel, dim3(32), dim3(256), 0, st, w2, reinterpret_cast<_Float16*>(im + IMG_OFF_W2),
               reinterpret_cast<float*>(im + IMG_OFF_K2));
    DVQ_CHECK_LAUNCH("pn_filter_pack_w2");
    return DVQ_OK;
}

// tiles of 256 points; h2buf [B][Npad][128] fp32, part [B][tiles][1024] float4 (+ float2 behind them), tstat [B][tiles][4] (zeroed by pn_center_kernel).
// Two halves, so that the caller may put them on different streams (pointnet.hip: the exact stage of one launch runs beside the trunk
// kernel of the next): dvq_launch_pn_filter_front = centres + trunk kernel(s), dvq_launch_pn_filter_back = pn_exact_kernel.
static int pn_filter_geometry(int N, int* tiles, int* deal) {
    *tiles = (N + 255) / 256;
    DVQ_REQUIRE(*tiles <= MAX_TILES, "pointnet: the filtered trunk takes at most %d points", MAX_TILES * 256);
    // 1 .. 32 points beyond a multiple of 256 (the 778 MANO vertices: 3 x 256 + 10): a tail tile of one block, four samples per
    // workgroup, instead of a last full tile of padding (DVQ_PN_TAIL=0: the full tile, for A/B runs; same features bit for bit)
    const int over = N - 256 * (*tiles - 1);
    *deal = (*tiles >= 2 && over <= 32 && dvq_knobs().pn_tail) ? *tiles - 1 : *tiles;
    return DVQ_OK;
}
#ifdef DVQ_DIAG
static int pn_abl() { const char* e = getenv("DVQ_PN_ABL"); return e ? atoi(e) : 0; }   // timing-only ablations / phase stamps: diagnostics build only
#else
static constexpr int pn_abl() { return 0; }
#endif

int dvq_launch_pn_filter_front(const float* pc, int C, int N, int Npad, long B, const float* trans, const float* W1, const float* b1,
                               const float* W2, const uint16_t* W2p, const float* b2, const void* w3f, float* h2buf, void* part,
                               unsigned* tstat, float* cbuf, unsigned long long* stats, hipStream_t st) {
    int tiles, deal;
    DVQ_PROPAGATE(pn_filter_geometry(N, &tiles, &deal));
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_trunk_filter_kernel<3, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, DVQ_DIAG_ON ? 100 * 1024 : F_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_trunk_filter_kernel<4, false>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, DVQ_DIAG_ON ? 100 * 1024 : F_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_trunk_filter_kernel<3, true>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_TAIL);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_trunk_filter_kernel<4, true>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_TAIL);
            return e;
        });
        if (e != hipSuccess) {
            dvq_set_error("pointnet: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    qf32x2* part2 = reinterpret_cast<qf32x2*>((char*)part + (size_t)B * tiles * 1024 * 16);   // behind the float4 records (pointnet.hip: 96 B per padded point)
    const long grid = B * deal;
    DVQ_REQUIRE(B * tiles < (1L << 31), "pointnet: grid too large");
    DVQ_REQUIRE(Npad >= N, "pointnet: bad padded row count");
    if (stats && hipMemsetAsync(stats, 0, 64, st) != hipSuccess) {       // statistics runs only; tstat is zeroed by pn_center_kernel
        dvq_set_error("pointnet: hipMemsetAsync failed");
        return DVQ_ELAUNCH;
    }
    {
        DVQ_PROF("pn_center", 2.0 * (double)B * 4 * (4.0 * 64 + 64.0 * 128), (double)B * (64 + 512), st);
        const unsigned cgrid = (unsigned)((B + CENTER_SPB - 1) / CENTER_SPB);
        if (C == 3) DVQ_LAUNCH(pn_center_kernel<3>, dim3(cgrid), dim3(256), 0, st, pc, trans, N, B, W1, b1, W2, b2, cbuf, tstat, tiles);
        else DVQ_LAUNCH(pn_center_kernel<4>, dim3(cgrid), dim3(256), 0, st, pc, trans, N, B, W1, b1, W2, b2, cbuf, tstat, tiles);
    }
    DVQ_CHECK_LAUNCH("pn_center");
    const double pts = (double)B * (deal * 256 + (deal < tiles ? 32 : 0));
    const int abl = pn_abl();
#ifdef DVQ_DIAG
    if ((abl & 262144) && C == 4) {
        // timing only (results INVALID): what splitting the trunk into a producer kernel (conv1 / conv2 / centring / conversion) and a
        // consumer kernel (conv3 loop) would buy if the two ran BESIDE each other: the same grid twice, the halves of the work on two streams
        static hipStream_t side = nullptr;
        static hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (!side) { (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking); (void)hipEventCreateWithFlags(&ev0, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev1, hipEventDisableTiming); }
        DVQ_PROF("pn_trunk", 2.0 * pts * (4.0 * 64 + 64.0 * 128 + 128.0 * 1024), pts * (16 + 512) + (double)grid * 16384, st);
        (void)hipEventRecord(ev0, st);
        (void)hipStreamWaitEvent(side, ev0, 0);
        DVQ_LAUNCH((pn_trunk_filter_kernel<4, false>), dim3((unsigned)grid), dim3(256), F_LDS, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                   b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, (abl & ~262144) | 2);
        DVQ_LAUNCH((pn_trunk_filter_kernel<4, false>), dim3((unsigned)grid), dim3(256), F_LDS, side, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                   b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, (abl & ~262144) | 131072 | 1);
        (void)hipEventRecord(ev1, side);
        (void)hipStreamWaitEvent(st, ev1, 0);
        return DVQ_OK;
    }
#endif
    {
        DVQ_PROF("pn_trunk", 2.0 * pts * (4.0 * 64 + 64.0 * 128 + 128.0 * 1024), pts * (16 + 512) + (double)grid * 16384, st);
        const int lds_main = (abl & 524288) ? 100 * 1024 : F_LDS;   // diagnostics: ONE workgroup per CU (what a wave costs when it has its SIMD to itself)
#ifdef DVQ_DIAG
        // diagnostics build, DVQ_PN_TRUNK3=1: three workgroups per CU (pn_trunk3_kernel) unless a timing diagnostic of the two-per-CU
        // kernel is asked for
        const bool three = dvq_knobs().pn_trunk3 && !(abl & ~(32768 | 65536));
        const int G_LDS = pn_g_lds();
        if (three && C == 3)
            DVQ_LAUNCH((pn_trunk3_kernel<3>), dim3((unsigned)grid), dim3(256), G_LDS, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                       b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, abl);
        else if (three)
            DVQ_LAUNCH((pn_trunk3_kernel<4>), dim3((unsigned)grid), dim3(256), G_LDS, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                       b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, abl);
        else
#endif
        if (C == 3)
            DVQ_LAUNCH((pn_trunk_filter_kernel<3, false>), dim3((unsigned)grid), dim3(256), lds_main, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                       b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, abl);
        else
            DVQ_LAUNCH((pn_trunk_filter_kernel<4, false>), dim3((unsigned)grid), dim3(256), lds_main, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                       b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, abl);
        if (deal < tiles) {
            const unsigned tgrid = (unsigned)((B + 3) / 4);
            if (C == 3)
                DVQ_LAUNCH((pn_trunk_filter_kernel<3, true>), dim3(tgrid), dim3(256), F_LDS_TAIL, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                           b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, abl);
            else
                DVQ_LAUNCH((pn_trunk_filter_kernel<4, true>), dim3(tgrid), dim3(256), F_LDS_TAIL, st, pc, trans, N, Npad, tiles, deal, B, W1, b1,
                           b2, (const char*)w3f, h2buf, (f32x4*)part, part2, tstat, cbuf, abl);
        }
    }
    DVQ_CHECK_LAUNCH("pn_trunk_filter");
    return DVQ_OK;
}

int dvq_launch_pn_filter_back(int N, int Npad, long B, const void* w3f, const float* w3, const float* b3, int relu, const float* h2buf,
                              const void* part, const unsigned* tstat, const float* cbuf, float* feat, long ld_feat, unsigned long long* stats,
                              hipStream_t st) {
    int tiles, deal;
    DVQ_PROPAGATE(pn_filter_geometry(N, &tiles, &deal));
    const qf32x2* part2 = reinterpret_cast<const qf32x2*>((const char*)part + (size_t)B * tiles * 1024 * 16);
    const int abl = pn_abl();
    const DvqKnobs& kn = dvq_knobs();
    const int exhaustive = kn.pn_exhaustive;
    int pair_cap = PAIR_CAP, fb_cap = FB_CAP;              // tests shrink the lists to reach the overflow paths
    if (kn.pn_caps[0] >= 0) {
        pair_cap = kn.pn_caps[0] > PAIR_CAP ? PAIR_CAP : kn.pn_caps[0];
        fb_cap = kn.pn_caps[1] > FB_CAP ? FB_CAP : kn.pn_caps[1];
    }
    {
        DVQ_PROF("pn_exact", 2.0 * (double)B * 1024 * 128, (double)B * (tiles * 16384.0 + 1024.0 * 512 + 4096), st);
        DVQ_LAUNCH(pn_exact_kernel, dim3((unsigned)B), dim3(256), 0, st, (const f32x4*)part, part2, tiles, deal, h2buf, N, Npad, w3, b3,
                   reinterpret_cast<const float*>((const char*)w3f + IMG_OFF_WN),
                   reinterpret_cast<const float*>((const char*)w3f + IMG_OFF_RN), tstat, cbuf, relu, exhaustive, pair_cap, fb_cap,
                   feat, ld_feat, stats, abl);
    }
    DVQ_CHECK_LAUNCH("pn_exact");
    if (stats && (abl & 4096)) {
        const long nrec = B * tiles;                       // a tail tile's record carries its workgroup's stamps too
        std::vector<unsigned> ts((size_t)nrec * 4);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(ts.data(), tstat, ts.size() * 4, hipMemcpyDeviceToHost);
        double a = 0, b2_ = 0, c = 0, d = 0;
        for (long i = 0; i < nrec; ++i) {
            const unsigned v = ts[4 * i + 3];
            a += v & 255; b2_ += (v >> 8) & 255; c += (v >> 16) & 255; d += (v >> 24) & 255;
        }
        if (abl & 8192) {
            double mhz = 0;
            for (long i = 0; i < nrec; ++i) mhz += ts[4 * i + 3];
            fprintf(stderr, "[dvq pn] in-kernel clock of the trunk kernel: %.0f MHz (s_memtime over s_memrealtime, mean over %ld workgroups)\n", mhz / nrec, nrec);
        } else
        fprintf(stderr, "[dvq pn] mean phase ticks per workgroup (s_memtime): start->loaded %.0f, conv1+conv2 %.0f, centre/convert %.0f, conv3 loop %.0f\n",
                a / nrec * 64, b2_ / nrec * 64, c / nrec * 64, d / nrec * 512);
    }
    if (stats) {                                          // diagnostics (DVQ_PN_STATS=1): synchronises
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, stats, sizeof h, hipMemcpyDeviceToHost);
        (void)hipMemset((void*)stats, 0, sizeof h);
        const double tot = (double)B * 1024;
        fprintf(stderr, "[dvq pn] B=%ld N=%d: one candidate %.4f, other counts %.4f of the channels, flagged 16-point groups %.5f per channel; %.3f candidate dots per channel\n",
                B, N, h[0] / tot, h[1] / tot, h[2] / tot, h[3] / tot);
        if (abl & 4096)
            fprintf(stderr, "[dvq pn] exact stage, mean cycles per workgroup: records -> candidates %.0f, sort by point %.0f, candidate dots %.0f, flagged groups + checks + store %.0f\n",
                    (double)h[4] / B, (double)h[5] / B, (double)h[6] / B, (double)h[7] / B);
    }
    return DVQ_OK;
}

int dvq_launch_pn_trunk_filter(const float* pc, int C, int N, int Npad, long B, const float* trans, const float* W1, const float* b1,
                               const float* W2, const uint16_t* W2p, const float* b2, const void* w3f, const float* w3, const float* b3,
                               int relu, float* h2buf, void* part, unsigned* tstat, float* cbuf, float* feat, long ld_feat,
                               unsigned long long* stats, hipStream_t st) {
    DVQ_PROPAGATE(dvq_launch_pn_filter_front(pc, C, N, Npad, B, trans, W1, b1, W2, W2p, b2, w3f, h2buf, part, tstat, cbuf, stats, st));
    return dvq_launch_pn_filter_back(N, Npad, B, w3f, w3, b3, relu, h2buf, part, tstat, cbuf, feat, ld_feat, stats, st);
}

// host side of the consistency counters: [0] suspect tile records, [1] channels outside their records' interval (per device)
int dvq_pn_fault_counters(unsigned long long* out2, int reset) {
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_pn_faults), 16) != hipSuccess) {
        dvq_set_error("pointnet_fault_counters: reading the device counters failed");
        return DVQ_ELAUNCH;
    }
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_pn_faults), z, 16) != hipSuccess) {
            dvq_set_error("pointnet_fault_counters: resetting the device counters failed");
            return DVQ_ELAUNCH;
        }
    }
    return DVQ_OK;
}
