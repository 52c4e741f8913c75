// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain)
// with the path's epilogues fused: bias/ReLU, residual, gated activation (tanh*sigmoid over
// gate-packed channel pairs + class-conditional bias), and the PointNet per-tile column max.
//
//   C[M,N] = sum_s A_s[M,K_s] * W_s[N,K_s]^T          (both operands K-contiguous, like nn.Linear)
//
// Tile: 128x128x32 per 256-thread workgroup (4 waves as 2(M) x 2(N), each 64x64 = 2x2 MFMA tiles),
// LDS rows padded to 36 floats (ds_read_b128 conflict-free: 36*m mod 64 distinct over a lane group),
// global->register->LDS double buffering (one barrier per K-tile), 2 workgroups per CU.
// Accumulation order is canonical: for every output, k runs 0,1,2,... across all sources in order
// (lane half h feeds k = 2t+h at MFMA step t), so results do not depend on the tiling.
// Block -> tile map is XCD-aware: the column tiles of one 128-row panel run on one XCD (blocks b and
// b+8 share an XCD), so the A panel is fetched into one L2 only.
#include "dvq_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
constexpr int TILE_FLOATS = (BM + BN) * LDT;            // one stage: A tile + W tile
constexpr size_t SMEM_BYTES = 2 * TILE_FLOATS * sizeof(float);

struct TileRegs {
    f32x4 a[4];
    f32x4 w[4];
};

__device__ __forceinline__ void load_tile(const GemmParams& p, int s, int k0, long m0, int n0, int tid,
                                          TileRegs& t) {
    const GemmSrc& src = p.src[s];
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        const long m = m0 + row;
        t.a[i] = (m < p.M) ? *reinterpret_cast<const f32x4*>(src.A + m * src.lda + k0 + c4 * 4)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = n0 + row;
        t.w[i] = (n < p.N) ? *reinterpret_cast<const f32x4*>(src.W + (long)n * src.ldw + k0 + c4 * 4)
                           : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

__device__ __forceinline__ void store_tile(float* stage, int tid, const TileRegs& t) {
    const int c4 = tid & 7;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        *reinterpret_cast<f32x4*>(stage + row * LDT + c4 * 4) = t.a[i];
        *reinterpret_cast<f32x4*>(stage + (BM + row) * LDT + c4 * 4) = t.w[i];
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int tiles_n = (p.N + BN - 1) / BN;
    const long tiles_m = (p.M + BM - 1) / BM;
    // XCD-aware map: panel (m-tile) index = 8 * (j / tiles_n) + (b % 8)
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

    // flattened (source, k-tile) iteration
    int s = 0, k0 = 0;
    TileRegs regs;
    load_tile(p, 0, 0, m0, n0, tid, regs);
    store_tile(smem, tid, regs);
    __syncthreads();
    int stage = 0;
    while (true) {
        // advance to the next (source, k-tile)
        int s_next = s, k_next = k0 + BK;
        if (k_next >= p.src[s].K) { s_next = s + 1; k_next = 0; }
        const bool has_next = s_next < p.nsrc;
        if (has_next) load_tile(p, s_next, k_next, m0, n0, tid, regs);

        const float* As = smem + stage * TILE_FLOATS + (wm * 64 + r) * LDT;
        const float* Ws = smem + stage * TILE_FLOATS + (BM + wn * 64 + r) * LDT;
#pragma unroll
        for (int u = 0; u < BK / 4; ++u) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + 4 * u);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + 32 * LDT + 4 * u);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(Ws + 4 * u);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(Ws + 32 * LDT + 4 * u);
            // MFMA step t consumes k = 2t + h: lane half 0 takes the even element, half 1 the odd one
            const float a0e = h ? a0[1] : a0[0], a0o = h ? a0[3] : a0[2];
            const float a1e = h ? a1[1] : a1[0], a1o = h ? a1[3] : a1[2];
            const float w0e = h ? w0[1] : w0[0], w0o = h ? w0[3] : w0[2];
            const float w1e = h ? w1[1] : w1[0], w1o = h ? w1[3] : w1[2];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0e, w0e, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0e, w1e, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1e, w0e, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1e, w1e, acc[1][1], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0o, w0o, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0o, w1o, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1o, w0o, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1o, w1o, acc[1][1], 0, 0, 0);
        }
        if (!has_next) break;
        store_tile(smem + (stage ^ 1) * TILE_FLOATS, tid, regs);
        __syncthreads();
        stage ^= 1;
        s = s_next;
        k0 = k_next;
    }

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

template <int EPI>
int launch(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_BYTES);
        if (e != hipSuccess) {
            dvq_set_error("gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
        attr_set = true;
    }
    const long tiles_m = (p.M + BM - 1) / BM;
    const long tiles_n = (p.N + BN - 1) / BN;
    const long grid = ((tiles_m + 7) / 8) * 8 * tiles_n;
    static const char* const names[] = {"gemm_bias", "gemm_resid", "gemm_gate", "gemm_colmax", "gemm_argmin"};
    double ksum = 0;
    for (int s = 0; s < p.nsrc; ++s) ksum += p.src[s].K;
    {
        DVQ_PROF(names[EPI], 2.0 * (double)p.M * p.N * ksum, ((double)p.M + p.N) * ksum * 4, stream);
        hipLaunchKernelGGL(gemm_f32_kernel<EPI>, dim3((unsigned)grid), dim3(256), SMEM_BYTES, stream, p);
    }
    DVQ_CHECK_LAUNCH("gemm_f32");
    return DVQ_OK;
}

}  // namespace

int dvq_launch_gemm(const GemmParams& p, GemmEpilogue epi, hipStream_t stream) {
    DVQ_REQUIRE(p.nsrc >= 1 && p.nsrc <= DVQ_MAX_SRC, "gemm: nsrc=%d out of range", p.nsrc);
    DVQ_REQUIRE(p.M > 0 && p.N > 0, "gemm: empty problem M=%ld N=%d", p.M, p.N);
    DVQ_REQUIRE(((p.M + BM - 1) / BM + 7) / 8 * 8 * ((p.N + BN - 1) / BN) < (1L << 31), "gemm: grid too large");
    for (int s = 0; s < p.nsrc; ++s) {
        const GemmSrc& g = p.src[s];
        DVQ_REQUIRE(g.A && g.W, "gemm: null operand in source %d", s);
        DVQ_REQUIRE(g.K > 0 && g.K % BK == 0, "gemm: K=%d of source %d is not a multiple of %d", g.K, s, BK);
        DVQ_REQUIRE(dvq_aligned16(g.A) && dvq_aligned16(g.W) && g.lda % 4 == 0 && g.ldw % 4 == 0,
                    "gemm: source %d rows are not 16-byte aligned", s);
    }
    switch (epi) {
        case EPI_BIAS:
            DVQ_REQUIRE(p.out, "gemm: null output");
            return launch<EPI_BIAS>(p, stream);
        case EPI_RESID:
            DVQ_REQUIRE(p.out && p.resid, "gemm: null output/residual");
            return launch<EPI_RESID>(p, stream);
        case EPI_GATE:
            DVQ_REQUIRE(p.out, "gemm: null output");
            DVQ_REQUIRE(p.N % BN == 0, "gemm: gated epilogue needs N %% 128 == 0 (N=%d)", p.N);
            DVQ_REQUIRE(!p.cls || p.label, "gemm: class bias without labels");
            return launch<EPI_GATE>(p, stream);
        case EPI_ARGMIN:
            break;
        case EPI_COLMAX:
            DVQ_REQUIRE(p.partial && p.rows_per_group > 0 && p.rows_per_group % BM == 0 && p.valid_rows > 0,
                        "gemm: bad column-max grouping");
            return launch<EPI_COLMAX>(p, stream);
    }
    if (epi == EPI_ARGMIN) {
        DVQ_REQUIRE(p.row_norm && p.col_norm && p.part_val && p.part_idx, "gemm: null argmin buffers");
        return launch<EPI_ARGMIN>(p, stream);
    }
    dvq_set_error("gemm: unknown epilogue %d", (int)epi);
    return DVQ_EINVAL;
}

extern "C" int dvq_linear(const dvq_gemm_src* src, int nsrc, int64_t M, int N, const float* bias, int act,
                          float* y, int64_t ldy, dvq_stream_t stream) {
    DVQ_REQUIRE(src && nsrc >= 1 && nsrc <= DVQ_MAX_SRC, "dvq_linear: nsrc=%d out of range", nsrc);
    DVQ_REQUIRE(act == DVQ_ACT_NONE || act == DVQ_ACT_RELU, "dvq_linear: unknown activation %d", act);
    if (M == 0) return DVQ_OK;
    GemmParams p = {};
    for (int s = 0; s < nsrc; ++s) p.src[s] = GemmSrc{src[s].x, src[s].w, (long)src[s].ldx, (long)src[s].ldw, src[s].K, 0};
    p.nsrc = nsrc;
    p.M = M;
    p.N = N;
    p.bias = bias;
    p.out = y;
    p.ldo = ldy;
    p.relu = act == DVQ_ACT_RELU;
    return dvq_launch_gemm(p, EPI_BIAS, (hipStream_t)stream);
}
