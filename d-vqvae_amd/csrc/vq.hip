// VQ codebook nearest neighbour: VectorQuantizer.forward(z, istrain=False), network/vqvae/quantizer.py:46-49.
//   d[m,k] = (zz[m] + ee[k]) - 2 * dot(z[m], E[k]),   idx[m] = argmin_k d[m,k]   (first min; NaN wins)
// Canonical evaluation order (shared bit for bit with oracle/vq_canonical.c): zz, ee and dot are
// fp32 fmaf chains over j = 0..D-1 starting from 0; d = fl(fl(zz + ee) - 2*dot).
// Exact kernel: the fp32 MFMA GEMM with A = codebook, W = z (so the reduction over entries is
// lane-local in the accumulator layout) and an (min, argmin) epilogue per 128-entry tile.
#include "dvq_internal.h"

namespace {

__global__ void rownorm_kernel(const float* __restrict__ x, long ld, long rows, int D, float* __restrict__ out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const float* p = x + row * ld;
    float acc = 0.f;
    for (int j = 0; j < D; j += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + j);
        acc = fmaf(v[0], v[0], acc);
        acc = fmaf(v[1], v[1], acc);
        acc = fmaf(v[2], v[2], acc);
        acc = fmaf(v[3], v[3], acc);
    }
    out[row] = acc;
}

__global__ void argmin_finish_kernel(const float* __restrict__ pv, const int* __restrict__ pi, int tiles, long M,
                                     int64_t* __restrict__ idx, float* __restrict__ dmin) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float bv = pv[m];
    int bi = pi[m];
    for (int t = 1; t < tiles; ++t) {
        const float v = pv[t * M + m];
        const int i = pi[t * M + m];
        if (dvq_argmin_better(v, i, bv, bi)) { bv = v; bi = i; }
    }
    idx[m] = bi;
    if (dmin) dmin[m] = bv;
}

struct VqScratch {
    float *zz, *ee, *pv;
    int* pi;
    int tiles;
    size_t bytes;
};

VqScratch plan(int64_t M, int K, void* ws) {
    VqScratch s;
    s.tiles = (K + 127) / 128;
    char* p = (char*)ws;
    auto take = [&](size_t n) { char* q = p; p += dvq_round_up(n, 256); return q; };
    s.zz = (float*)take((size_t)M * 4);
    s.ee = (float*)take((size_t)K * 4);
    s.pv = (float*)take((size_t)s.tiles * M * 4);
    s.pi = (int*)take((size_t)s.tiles * M * 4);
    s.bytes = (size_t)(p - (char*)ws);
    return s;
}

}  // namespace

extern "C" size_t dvq_vq_argmin_workspace_bytes(int64_t M, int K) {
    if (M <= 0 || K <= 0) return 256;
    return plan(M, K, nullptr).bytes;
}

extern "C" int dvq_vq_argmin(const float* z, int64_t ldz, const float* E, int64_t M, int K, int D, int64_t* idx,
                             float* dmin, void* workspace, size_t workspace_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(M >= 0 && K > 0 && D > 0, "vq_argmin: bad shape M=%ld K=%d D=%d", (long)M, K, D);
    DVQ_REQUIRE(D % 32 == 0, "vq_argmin: D=%d must be a multiple of 32", D);
    if (M == 0) return DVQ_OK;
    DVQ_REQUIRE(z && E && idx, "vq_argmin: null pointer");
    DVQ_REQUIRE(ldz >= D && ldz % 4 == 0 && dvq_aligned16(z) && dvq_aligned16(E), "vq_argmin: rows not 16-byte aligned");
    DVQ_REQUIRE(workspace && dvq_aligned16(workspace), "vq_argmin: null/unaligned workspace");
    const VqScratch s = plan(M, K, workspace);
    if (workspace_bytes < s.bytes) {
        dvq_set_error("vq_argmin: workspace %zu < %zu bytes", workspace_bytes, s.bytes);
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    DVQ_PROF("vq_argmin_total", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
    DVQ_LAUNCH(rownorm_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, z, (long)ldz, (long)M, D, s.zz);
    DVQ_CHECK_LAUNCH("rownorm(z)");
    DVQ_LAUNCH(rownorm_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, E, (long)D, (long)K, D, s.ee);
    DVQ_CHECK_LAUNCH("rownorm(E)");
    GemmParams p = {};
    p.src[0] = GemmSrc{E, z, (long)D, (long)ldz, D, 0};
    p.nsrc = 1;
    p.M = K;            // A rows = codebook entries
    p.N = (int)M;       // W rows = z rows
    DVQ_REQUIRE(M < (1L << 31), "vq_argmin: M too large");
    p.row_norm = s.ee;
    p.col_norm = s.zz;
    p.part_val = s.pv;
    p.part_idx = s.pi;
    DVQ_PROPAGATE(dvq_launch_gemm(p, EPI_ARGMIN, st));
    DVQ_LAUNCH(argmin_finish_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, s.pv, s.pi, s.tiles, (long)M,
                       idx, dmin);
    DVQ_CHECK_LAUNCH("argmin_finish");
    return DVQ_OK;
}
