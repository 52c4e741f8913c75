// MANO hand layer (linear blend skinning): the third-party `mano` package call at network/gen_net.py:116-118
// and gen_diverse_grasp_obman.py:252-253, restated from the published smplx-style algorithm
// ("parity unpinned": the package is neither installed nor vendored).
// One 256-thread workgroup per sample; the small per-sample state (pose, rotations, joint chain) lives
// in LDS, the vertex loop streams posedirs/shapedirs (L2-resident, shared by all samples).
#include "dvq_internal.h"

namespace {

constexpr int NV = 778, NJ = 16, NB = 10, NP = 45, NPF = 135;

__global__ __launch_bounds__(256) void mano_kernel(dvq_mano_model m, const float* __restrict__ betas, long ldb,
                                                   const float* __restrict__ pose, long ldp,
                                                   const float* __restrict__ gorient, long ldg,
                                                   const float* __restrict__ transl, long ldt, long B,
                                                   float* __restrict__ verts, int layout, float* __restrict__ joints) {
    __shared__ float s_beta[NB];
    __shared__ float s_pose[NP];
    __shared__ float s_full[48];
    __shared__ float s_R[NJ][9];
    __shared__ float s_pf[NPF];
    __shared__ float s_J[NJ][3];
    __shared__ float s_G[NJ][12];   // world transforms, rows of [R | t]
    __shared__ float s_A[NJ][12];   // skinning transforms
    __shared__ float s_t[3];
    const long b = blockIdx.x;
    const int t = threadIdx.x;
    if (t < NB) s_beta[t] = betas[b * ldb + t];
    if (t < NP) s_pose[t] = pose[b * ldp + t];
    if (t < 3) s_t[t] = transl ? transl[b * ldt + t] : 0.f;
    __syncthreads();
    if (t < 48) {   // full_pose = [global_orient | hand_pose @ comps] + pose_mean
        float v;
        if (t < 3) {
            v = gorient ? gorient[b * ldg + t] : 0.f;
        } else {
            v = 0.f;
            for (int i = 0; i < NP; ++i) v = fmaf(s_pose[i], m.comps[i * NP + (t - 3)], v);
        }
        s_full[t] = v + m.pose_mean[t];
        // rest joints: J = J_regressor @ (v_template + shapedirs . beta), regressor folded at pack time
        float j = m.j_template[t];
        for (int l = 0; l < NB; ++l) j = fmaf(s_beta[l], m.j_shapedirs[l * 48 + t], j);
        s_J[t / 3][t % 3] = j;
    }
    __syncthreads();
    if (t < NJ) {   // Rodrigues, angle = ||r + 1e-8||
        const float rx = s_full[3 * t], ry = s_full[3 * t + 1], rz = s_full[3 * t + 2];
        const float ex = rx + 1e-8f, ey = ry + 1e-8f, ez = rz + 1e-8f;
        const float angle = sqrtf(ex * ex + ey * ey + ez * ez);
        const float ax = rx / angle, ay = ry / angle, az = rz / angle;
        const float s = sinf(angle), c1 = 1.0f - cosf(angle);
        // K = [[0,-az,ay],[az,0,-ax],[-ay,ax,0]];  R = I + s K + (1-c) K^2
        const float K[9] = {0.f, -az, ay, az, 0.f, -ax, -ay, ax, 0.f};
        float K2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) K2[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
        for (int i = 0; i < 9; ++i) s_R[t][i] = ((i % 4 == 0) ? 1.0f : 0.0f) + s * K[i] + c1 * K2[i];
    }
    __syncthreads();
    if (t < NPF) s_pf[t] = s_R[1 + t / 9][t % 9] - ((t % 9) % 4 == 0 ? 1.0f : 0.0f);
    if (t == 0) {   // kinematic chain (parents[j] < j), 16 tiny 3x4 products
        for (int j = 0; j < NJ; ++j) {
            const int p = m.parents[j];
            float rel[3];
            for (int i = 0; i < 3; ++i) rel[i] = s_J[j][i] - (p >= 0 ? s_J[p][i] : 0.f);
            if (p < 0) {
                for (int i = 0; i < 3; ++i) {
                    for (int c = 0; c < 3; ++c) s_G[j][i * 4 + c] = s_R[j][i * 3 + c];
                    s_G[j][i * 4 + 3] = rel[i];
                }
            } else {
                for (int i = 0; i < 3; ++i) {
                    for (int c = 0; c < 3; ++c)
                        s_G[j][i * 4 + c] = s_G[p][i * 4] * s_R[j][c] + s_G[p][i * 4 + 1] * s_R[j][3 + c] + s_G[p][i * 4 + 2] * s_R[j][6 + c];
                    s_G[j][i * 4 + 3] = s_G[p][i * 4] * rel[0] + s_G[p][i * 4 + 1] * rel[1] + s_G[p][i * 4 + 2] * rel[2] + s_G[p][i * 4 + 3];
                }
            }
        }
        for (int j = 0; j < NJ; ++j)      // A = G with the rest joint removed: t_A = t_G - R_G J
            for (int i = 0; i < 3; ++i) {
                for (int c = 0; c < 3; ++c) s_A[j][i * 4 + c] = s_G[j][i * 4 + c];
                s_A[j][i * 4 + 3] = s_G[j][i * 4 + 3] - (s_G[j][i * 4] * s_J[j][0] + s_G[j][i * 4 + 1] * s_J[j][1] + s_G[j][i * 4 + 2] * s_J[j][2]);
            }
    }
    __syncthreads();
    if (joints && t < 48) joints[b * 48 + t] = s_G[t / 3][(t % 3) * 4 + 3] + s_t[t % 3];
    for (int v = t; v < NV; v += 256) {
        float vp[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int e = v * 3 + k;
            float sh = m.v_template[e];
            for (int l = 0; l < NB; ++l) sh = fmaf(s_beta[l], m.shapedirs[l * (NV * 3) + e], sh);
            float po = 0.f;
            for (int q = 0; q < NPF; ++q) po = fmaf(s_pf[q], m.posedirs[q * (NV * 3) + e], po);
            vp[k] = po + sh;
        }
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.f;
        for (int j = 0; j < NJ; ++j) {
            const float wj = m.weights[v * NJ + j];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = fmaf(wj, s_A[j][i], T[i]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float o = T[i * 4] * vp[0] + T[i * 4 + 1] * vp[1] + T[i * 4 + 2] * vp[2] + T[i * 4 + 3] + s_t[i];
            if (layout == 0) verts[(b * NV + v) * 3 + i] = o;
            else verts[(b * 3 + i) * NV + v] = o;
        }
    }
}

}  // namespace

extern "C" int dvq_mano_forward(const dvq_mano_model* m, const float* betas, int64_t ldb, const float* pose, int64_t ldp,
                                const float* global_orient, int64_t ldg, const float* transl, int64_t ldt, int64_t B,
                                float* verts, int layout, float* joints, dvq_stream_t stream) {
    DVQ_REQUIRE(m && betas && pose && verts, "mano_forward: null pointer");
    DVQ_REQUIRE(m->v_template && m->shapedirs && m->posedirs && m->j_template && m->j_shapedirs && m->weights && m->comps &&
                    m->pose_mean, "mano_forward: incomplete model");
    DVQ_REQUIRE(layout == 0 || layout == 1, "mano_forward: layout must be 0 ([B,778,3]) or 1 ([B,3,778])");
    DVQ_REQUIRE(B >= 0 && ldb >= 10 && ldp >= 45, "mano_forward: bad strides");
    for (int j = 0; j < 16; ++j) DVQ_REQUIRE(m->parents[j] < j, "mano_forward: parents[%d]=%d is not an ancestor index", j, m->parents[j]);
    if (B == 0) return DVQ_OK;
    {
        DVQ_PROF("mano_lbs", (double)B * 1.17e6, (double)B * (55 + 2334) * 4, (hipStream_t)stream);
        DVQ_LAUNCH(mano_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, *m, betas, (long)ldb, pose, (long)ldp,
                           global_orient, (long)ldg, transl, (long)ldt, (long)B, verts, layout, joints);
    }
    DVQ_CHECK_LAUNCH("mano_forward");
    return DVQ_OK;
}
