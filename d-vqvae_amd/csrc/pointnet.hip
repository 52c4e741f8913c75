// PointNet encoder (PointNetEncoder.forward, network/pointnet_encoder.py:140-169; STN3d.forward :27-45).
// Layer 1 (C -> 64, K = 3/4) is a VALU kernel that also applies the 3x3 input transform; layers 2/3
// run on the fp32 MFMA GEMM, the 128 -> 1024 layer with the per-sample max fused into its epilogue so
// the [N,1024] activation is never written.  BatchNorm (eval) is folded into the weights by the packer.
#include "dvq_internal.h"

namespace {

// 16 threads per point, 4 output channels each; rows >= N of a sample's padded block are zero-filled.
__global__ void pn_layer1_kernel(const float* __restrict__ pc, int C, int N, int Npad, long B,
                                 const float* __restrict__ trans /* [B,9] or null */, const float* __restrict__ W /* [64,4] */,
                                 const float* __restrict__ bias, float* __restrict__ out /* [B*Npad,64] */) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pt = gid >> 4;
    const int q = (int)(gid & 15);
    if (pt >= B * Npad) return;
    const long b = pt / Npad;
    const int n = (int)(pt % Npad);
    f32x4 y = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const float* src = pc + b * (long)C * N;
        float x0 = src[n], x1 = src[N + n], x2 = src[2 * N + n];
        const float x3 = (C > 3) ? src[3 * N + n] : 0.f;
        if (trans) {   // xyz_new = xyz @ trans[b]  (row vector times matrix, pointnet_encoder.py:146)
            const float* t = trans + b * 9;
            const float n0 = fmaf(x2, t[6], fmaf(x1, t[3], x0 * t[0]));
            const float n1 = fmaf(x2, t[7], fmaf(x1, t[4], x0 * t[1]));
            const float n2 = fmaf(x2, t[8], fmaf(x1, t[5], x0 * t[2]));
            x0 = n0; x1 = n1; x2 = n2;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float* w = W + (q * 4 + o) * 4;
            float v = x0 * w[0];
            v = fmaf(x1, w[1], v);
            v = fmaf(x2, w[2], v);
            v = fmaf(x3, w[3], v);
            y[o] = fmaxf(v + bias[q * 4 + o], 0.f);
        }
    }
    *reinterpret_cast<f32x4*>(out + pt * 64 + q * 4) = y;
}

struct PnScratch {
    float *h1, *h2, *part, *f0, *f1, *f2, *tr;
    long chunk;
    int Npad;
    size_t bytes;
};

PnScratch plan(int64_t B, int N, void* ws) {
    PnScratch s;
    s.Npad = (N + 127) / 128 * 128;
    const size_t per_sample = (size_t)s.Npad * (64 + 128) * 4 + (size_t)(s.Npad / 128) * 1024 * 4 + (1024 + 512 + 256 + 16) * 4;
    const size_t budget = (size_t)6 << 30;
    long chunk = (long)(budget / per_sample);
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    if (chunk < 1) chunk = 1;
    s.chunk = chunk;
    char* p = (char*)ws;
    auto take = [&](size_t n) { char* q = p; p += dvq_round_up(n, 256); return (float*)q; };
    s.h1 = take((size_t)chunk * s.Npad * 64 * 4);
    s.h2 = take((size_t)chunk * s.Npad * 128 * 4);
    s.part = take((size_t)chunk * (s.Npad / 128) * 1024 * 4);
    s.f0 = take((size_t)chunk * 1024 * 4);
    s.f1 = take((size_t)chunk * 512 * 4);
    s.f2 = take((size_t)chunk * 256 * 4);
    s.tr = take((size_t)chunk * 16 * 4);
    s.bytes = (size_t)(p - (char*)ws);
    return s;
}

int dense(const float* x, long ldx, int K, const float* w, const uint16_t* wp, const float* b, long M, int N, int relu,
          float* y, long ldy, hipStream_t st) {
    GemmParams p = {};
    p.src[0] = GemmSrc{x, w, ldx, (long)K, K, 0, wp, (long)N * K};
    p.nsrc = 1;
    p.M = M;
    p.N = N;
    p.bias = b;
    p.out = y;
    p.ldo = ldy;
    p.relu = relu;
    return dvq_launch_gemm(p, EPI_BIAS, st);
}

int trunk(const float* pc, int C, int N, long Bc, const float* trans, const float* w1, const float* b1, const float* w2,
          const uint16_t* w2p, const float* b2, const float* w3, const uint16_t* w3p, const float* b3, int relu3, const PnScratch& s, float* feat, long ld_feat,
          hipStream_t st) {
    const long rows = Bc * s.Npad;
    const long threads = rows * 16;
    {
        DVQ_PROF("pn_layer1", 2.0 * rows * 64 * 4, (double)rows * (16 + 256), st);
        hipLaunchKernelGGL(pn_layer1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, pc, C, N, s.Npad, Bc,
                           trans, w1, b1, s.h1);
    }
    DVQ_CHECK_LAUNCH("pn_layer1");
    DVQ_PROPAGATE(dense(s.h1, 64, 64, w2, w2p, b2, rows, 128, 1, s.h2, 128, st));
    GemmParams p = {};
    p.src[0] = GemmSrc{s.h2, w3, 128, 128, 128, 0, w3p, 1024L * 128};
    p.nsrc = 1;
    p.M = rows;
    p.N = 1024;
    p.bias = b3;
    p.relu = 0;                  // max(relu(x)) == relu(max(x)): the ReLU is applied by the reduction
    p.rows_per_group = s.Npad;
    p.valid_rows = N;
    p.partial = s.part;
    DVQ_PROPAGATE(dvq_launch_gemm(p, EPI_COLMAX, st));
    return dvq_launch_colmax_reduce(s.part, Bc, s.Npad / 128, 1024, relu3, feat, ld_feat, st);
}

}  // namespace

extern "C" size_t dvq_pointnet_workspace_bytes(int64_t B, int N) {
    if (B <= 0 || N <= 0) return 256;
    return plan(B, N, nullptr).bytes;
}

extern "C" int dvq_pointnet_encode(const dvq_pointnet_weights* w, const float* pc, int64_t B, int N, float* feat,
                                   int64_t ld_feat, float* trans_out, void* workspace, size_t workspace_bytes,
                                   dvq_stream_t stream) {
    DVQ_REQUIRE(w && pc && feat, "pointnet_encode: null pointer");
    DVQ_REQUIRE(w->C == 3 || w->C == 4, "pointnet_encode: channel count %d not supported (3 or 4)", w->C);
    DVQ_REQUIRE(B >= 0 && N > 0, "pointnet_encode: bad shape B=%ld N=%d", (long)B, N);
    DVQ_REQUIRE(ld_feat >= 1024, "pointnet_encode: ld_feat=%ld < 1024", (long)ld_feat);
    if (B == 0) return DVQ_OK;
    DVQ_REQUIRE(workspace && dvq_aligned16(workspace), "pointnet_encode: null/unaligned workspace");
    const PnScratch s = plan(B, N, workspace);
    if (workspace_bytes < s.bytes) {
        dvq_set_error("pointnet_encode: workspace %zu < %zu bytes", workspace_bytes, s.bytes);
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int64_t b0 = 0; b0 < B; b0 += s.chunk) {
        const long Bc = (long)((B - b0 < s.chunk) ? (B - b0) : s.chunk);
        const float* pcb = pc + b0 * (long)w->C * N;
        // STN3d: trunk with ReLU on the last layer, then fc1/fc2 (BN folded, ReLU) and fc3 (+identity)
        DVQ_PROPAGATE(trunk(pcb, w->C, N, Bc, nullptr, w->s_w1, w->s_b1, w->s_w2, w->s_w2p, w->s_b2, w->s_w3, w->s_w3p, w->s_b3,
                            1, s, s.f0, 1024, st));
        DVQ_PROPAGATE(dense(s.f0, 1024, 1024, w->s_f1, w->s_f1p, w->s_c1, Bc, 512, 1, s.f1, 512, st));
        DVQ_PROPAGATE(dense(s.f1, 512, 512, w->s_f2, w->s_f2p, w->s_c2, Bc, 256, 1, s.f2, 256, st));
        float* tr = trans_out ? trans_out + b0 * 9 : s.tr;
        DVQ_PROPAGATE(dense(s.f2, 256, 256, w->s_f3, w->s_f3p, w->s_c3, Bc, 9, 0, tr, 9, st));
        // main trunk on the transformed cloud; no ReLU after the last BN (pointnet_encoder.py:162)
        DVQ_PROPAGATE(trunk(pcb, w->C, N, Bc, tr, w->w1, w->b1, w->w2, w->w2p, w->b2, w->w3, w->w3p, w->b3, 0, s,
                            feat + b0 * ld_feat, ld_feat, st));
    }
    return DVQ_OK;
}
