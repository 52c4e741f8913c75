// MANO hand layer (linear blend skinning): the third-party `mano` package call at network/gen_net.py:116-118
// and gen_diverse_grasp_obman.py:252-253, restated from the published smplx-style algorithm
// ("parity unpinned": the package is neither installed nor vendored).
// Three launches per chunk of samples:
//   mano_pose_kernel   one wave per sample: PCA pose -> axis-angle, Rodrigues, pose feature, rest joints, kinematic chain
//                      -> X[b] = [betas(10) | R_1..15 - I (135) | 0 (15)], skinning transforms A[b][16][3x4], joints
//   blendshape GEMM    V[b, 0:2334] = X[b] . Wblend^T + v_template: shape and pose blendshapes of all vertices as ONE
//                      [B,160] x [160,2334] product on the matrix cores (the per-sample form re-read the 1.3 MB of
//                      blendshape bases from L2 for every sample)
//   mano_skin_kernel   per (sample, vertex): T = sum_j w[v][j] A[b][j]; out = T V[b][v] + transl
#include "dvq_internal.h"

namespace {

constexpr int NV = 778, NJ = 16, NB = 10, NP = 45, NPF = 135;
constexpr int XK = 160;                    // GEMM reduction length: 10 + 135, padded to a multiple of 32
constexpr int VLD = 2336;                  // row stride of the blendshape result (2334 padded to a multiple of 4)
constexpr long CHUNK = 16384;              // samples per pass (scratch: 10.8 KB per sample)
constexpr int WPB = 4;                     // waves (samples) per pose block

__global__ __launch_bounds__(64 * WPB) void mano_pose_kernel(dvq_mano_model m, const float* __restrict__ betas, long ldb,
                                                            const float* __restrict__ pose, long ldp,
                                                            const float* __restrict__ gorient, long ldg,
                                                            const float* __restrict__ transl, long ldt, long B,
                                                            float* __restrict__ X, float* __restrict__ A, float* __restrict__ joints) {
    __shared__ float s_beta[WPB][NB];
    __shared__ float s_pose[WPB][NP];
    __shared__ float s_full[WPB][48];
    __shared__ float s_R[WPB][NJ][9];
    __shared__ float s_J[WPB][NJ][3];
    __shared__ float s_G[WPB][NJ][12];   // world transforms, rows of [R | t]
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
    const long b = (long)blockIdx.x * WPB + w;
    const bool live = b < B;
    const long bb = live ? b : B - 1;
    if (t < NB) s_beta[w][t] = betas[bb * ldb + t];
    if (t < NP) s_pose[w][t] = pose[bb * ldp + t];
    __syncthreads();
    if (t < 48) {   // full_pose = [global_orient | hand_pose @ comps] + pose_mean
        float v;
        if (t < 3) {
            v = gorient ? gorient[bb * ldg + t] : 0.f;
        } else {
            v = 0.f;
            for (int i = 0; i < NP; ++i) v = fmaf(s_pose[w][i], m.comps[i * NP + (t - 3)], v);
        }
        s_full[w][t] = v + m.pose_mean[t];
        // rest joints: J = J_regressor @ (v_template + shapedirs . beta), regressor folded at pack time
        float j = m.j_template[t];
        for (int l = 0; l < NB; ++l) j = fmaf(s_beta[w][l], m.j_shapedirs[l * 48 + t], j);
        s_J[w][t / 3][t % 3] = j;
    }
    __syncthreads();
    if (t < NJ) {   // Rodrigues, angle = ||r + 1e-8||
        const float rx = s_full[w][3 * t], ry = s_full[w][3 * t + 1], rz = s_full[w][3 * t + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float ax = rx / angle, ay = ry / angle, az = rz / angle;
        const float s = sinf(angle), c1 = 1.0f - cosf(angle);
        // K = [[0,-az,ay],[az,0,-ax],[-ay,ax,0]];  R = I + s K + (1-c) K^2
        const float K[9] = {0.f, -az, ay, az, 0.f, -ax, -ay, ax, 0.f};
        float K2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
        for (int i = 0; i < 9; ++i) s_R[w][t][i] = ((i % 4 == 0) ? 1.0f : 0.0f) + s * K[i] + c1 * K2[i];
    }
    __syncthreads();
    if (live) {     // the GEMM's left operand
        float* x = X + b * XK;
        for (int u = t; u < XK; u += 64) {
            float v = 0.f;
            if (u < NB) v = s_beta[w][u];
            else if (u < NB + NPF) {
                const int q = u - NB;
                v = s_R[w][1 + q / 9][q % 9] - ((q % 9) % 4 == 0 ? 1.0f : 0.0f);
            }
            x[u] = v;
        }
    }
    if (t == 0) {   // kinematic chain (parents[j] < j), 16 tiny 3x4 products
        for (int j = 0; j < NJ; ++j) {
            const int p = m.parents[j];
            float rel[3];
            for (int i = 0; i < 3; ++i) rel[i] = s_J[w][j][i] - (p >= 0 ? s_J[w][p][i] : 0.f);
            if (p < 0) {
                for (int i = 0; i < 3; ++i) {
                    for (int c = 0; c < 3; ++c) s_G[w][j][i * 4 + c] = s_R[w][j][i * 3 + c];
                    s_G[w][j][i * 4 + 3] = rel[i];
                }
            } else {
                for (int i = 0; i < 3; ++i) {
                    for (int c = 0; c < 3; ++c)
                        s_G[w][j][i * 4 + c] = s_G[w][p][i * 4] * s_R[w][j][c] + s_G[w][p][i * 4 + 1] * s_R[w][j][3 + c] +
                                               s_G[w][p][i * 4 + 2] * s_R[w][j][6 + c];
                    s_G[w][j][i * 4 + 3] = s_G[w][p][i * 4] * rel[0] + s_G[w][p][i * 4 + 1] * rel[1] + s_G[w][p][i * 4 + 2] * rel[2] +
                                           s_G[w][p][i * 4 + 3];
                }
            }
        }
    }
    __syncthreads();
    if (!live) return;
    const float tr = (transl && t < 48) ? transl[b * ldt + t % 3] : 0.f;
    if (joints && t < 48) joints[b * 48 + t] = s_G[w][t / 3][(t % 3) * 4 + 3] + tr;
    // A = G with the rest joint removed: t_A = t_G - R_G J (the translation also takes the sample's `transl`: weights sum to 1)
    for (int u = t; u < NJ * 12; u += 64) {
        const int j = u / 12, e = u % 12, i = e / 4, c = e % 4;
        float v = s_G[w][j][e];
        if (c == 3) v -= s_G[w][j][i * 4] * s_J[w][j][0] + s_G[w][j][i * 4 + 1] * s_J[w][j][1] + s_G[w][j][i * 4 + 2] * s_J[w][j][2];
        A[b * (NJ * 12) + u] = v;
    }
}

// verts[b][v] = (sum_j w[v][j] A[b][j]) . [V[b][v]; 1] + transl[b]
__global__ __launch_bounds__(256) void mano_skin_kernel(const float* __restrict__ weights, const float* __restrict__ V,
                                                        const float* __restrict__ A, const float* __restrict__ transl, long ldt,
                                                        float* __restrict__ verts, int layout) {
    __shared__ float s_A[NJ][12];
    const long b = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < NJ * 12) s_A[threadIdx.x / 12][threadIdx.x % 12] = A[b * (NJ * 12) + threadIdx.x];
    __syncthreads();
    if (v >= NV) return;
    const float* vp = V + b * VLD + 3 * v;
    const float x = vp[0], y = vp[1], z = vp[2];
    float T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    for (int j = 0; j < NJ; ++j) {
        const float wj = weights[v * NJ + j];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = fmaf(wj, s_A[j][i], T[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float o = T[i * 4] * x + T[i * 4 + 1] * y + T[i * 4 + 2] * z + T[i * 4 + 3] + (transl ? transl[b * ldt + i] : 0.f);
        if (layout == 0) verts[(b * NV + v) * 3 + i] = o;
        else verts[(b * 3 + i) * NV + v] = o;
    }
}

}  // namespace

extern "C" size_t dvq_mano_workspace_bytes(int64_t B) {
    const long c = B < CHUNK ? (B > 0 ? B : 1) : CHUNK;
    return dvq_round_up((size_t)c * XK * 4, 256) + dvq_round_up((size_t)c * VLD * 4, 256) + dvq_round_up((size_t)c * NJ * 12 * 4, 256);
}

extern "C" int dvq_mano_forward(const dvq_mano_model* m, const float* betas, int64_t ldb, const float* pose, int64_t ldp,
                                const float* global_orient, int64_t ldg, const float* transl, int64_t ldt, int64_t B,
                                float* verts, int layout, float* joints, void* workspace, size_t workspace_bytes,
                                dvq_stream_t stream) {
    DVQ_REQUIRE(m && betas && pose && verts, "mano_forward: null pointer");
    DVQ_REQUIRE(m->v_template && m->blend_w && m->j_template && m->j_shapedirs && m->weights && m->comps && m->pose_mean,
                "mano_forward: incomplete model");
    DVQ_REQUIRE(layout == 0 || layout == 1, "mano_forward: layout must be 0 ([B,778,3]) or 1 ([B,3,778])");
    DVQ_REQUIRE(B >= 0 && ldb >= 10 && ldp >= 45, "mano_forward: bad strides");
    for (int j = 0; j < 16; ++j) DVQ_REQUIRE(m->parents[j] < j, "mano_forward: parents[%d]=%d is not an ancestor index", j, m->parents[j]);
    if (B == 0) return DVQ_OK;
    DVQ_REQUIRE(workspace && dvq_aligned16(workspace), "mano_forward: null/unaligned workspace");
    if (workspace_bytes < dvq_mano_workspace_bytes(B)) {
        dvq_set_error("mano_forward: workspace %zu < %zu bytes", workspace_bytes, dvq_mano_workspace_bytes(B));
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const long cmax = B < CHUNK ? B : CHUNK;
    char* p = (char*)workspace;
    float* X = (float*)p;
    p += dvq_round_up((size_t)cmax * XK * 4, 256);
    float* V = (float*)p;
    p += dvq_round_up((size_t)cmax * VLD * 4, 256);
    float* A = (float*)p;
    for (long b0 = 0; b0 < B; b0 += CHUNK) {
        const long nb = B - b0 < CHUNK ? B - b0 : CHUNK;
        {
            DVQ_PROF("mano_pose", (double)nb * 2.0e4, (double)nb * (55 + XK + NJ * 12) * 4, st);
            DVQ_LAUNCH(mano_pose_kernel, dim3((unsigned)((nb + WPB - 1) / WPB)), dim3(64 * WPB), 0, st, *m, betas + b0 * ldb, (long)ldb,
                       pose + b0 * ldp, (long)ldp, global_orient ? global_orient + b0 * ldg : nullptr, (long)ldg,
                       transl ? transl + b0 * ldt : nullptr, (long)ldt, nb, X, A, joints ? joints + b0 * 48 : nullptr);
        }
        DVQ_CHECK_LAUNCH("mano_pose");
        GemmParams g = {};
        g.src[0] = GemmSrc{X, m->blend_w, XK, XK, XK, m->planes_kind, m->blend_w_planes, (long)NV * 3 * XK};
        g.wscale = m->blend_w_scale;
        g.nsrc = 1;
        g.M = nb;
        g.N = NV * 3;
        g.bias = m->v_template;
        g.out = V;
        g.ldo = VLD;
        DVQ_PROPAGATE(dvq_launch_gemm(g, EPI_BIAS, st));
        {
            DVQ_PROF("mano_skin", (double)nb * NV * (NJ * 24 + 18), (double)nb * NV * 3 * 8, st);
            DVQ_LAUNCH(mano_skin_kernel, dim3((NV + 255) / 256, (unsigned)nb), dim3(256), 0, st, m->weights, V, A,
                       transl ? transl + b0 * ldt : nullptr, (long)ldt, verts + b0 * NV * 3, layout);
        }
        DVQ_CHECK_LAUNCH("mano_skin");
    }
    return DVQ_OK;
}
