// Microbenchmark (GPU box): the VQ chunk loop's ingredients added one at a time beside v_mfma_f32_32x32x16_f16 with two
// waves per SIMD: (a) 6 vector insts per MFMA (fma + perm + 4 med3), (b) + one ds_read_b128 per MFMA feeding operand A,
// read one iteration ahead, (c) + the result of an OLDER MFMA as the vector instructions' input (as the score absorption
// reads the previous chunk's accumulators); (d) the reads alone; (e) half the reads.  8 waves x 1 KiB per MFMA slot is
// exactly the LDS's 128 B/clk.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
    __syncthreads();
    f16x8 b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)(i * 0.5f);
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    auto frag = [&](int row, int chunk) { return lds + (row & 63) * 512 + 16 * ((chunk & 31) ^ (row & 15)); };   // the kernel's swizzle: conflict-free
    f32x16 accA[2] = {{0}, {0}}, accB[2] = {{0}, {0}};
    float m1 = 1e30f, m2 = 1e30f, m3 = 1e30f, m4 = 1e30f, inv = 0.5f;
    unsigned ids = 0x03020100u;
    f16x8 e0 = *reinterpret_cast<const f16x8*>(frag(r, h)), e1 = *reinterpret_cast<const f16x8*>(frag(32 + r, h));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x16 (&cur)[2] = (it & 1) ? accB : accA;
        f32x16 (&prev)[2] = (it & 1) ? accA : accB;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            f16x8 n0 = e0, n1 = e1;
            if (MODE >= 1 && !(MODE == 4 && (s & 1))) {           // next fragments (MODE 4: every other step, i.e. one read per two MFMAs)
                n0 = *reinterpret_cast<const f16x8*>(frag(r, 2 * (s + 1 + 8 * (it & 1)) + h));
                n1 = *reinterpret_cast<const f16x8*>(frag(32 + r, 2 * (s + 1 + 8 * (it & 1)) + h));
            }
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, b, cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, b, cur[1], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < (MODE == 3 ? 0 : 2); ++q) {
                const float src = (MODE >= 2) ? prev[q][(2 * s) & 15] : (float)(it + s);
                const float sc = fmaf(src, inv, m1);
                const float p = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(sc), ids, 0x07060500u + q));
                m4 = __builtin_amdgcn_fmed3f(m3, m4, p); m3 = __builtin_amdgcn_fmed3f(m2, m3, p);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, p); m1 = __builtin_amdgcn_fmed3f(m1, p, -3e38f);
            }
            e0 = n0; e1 = n1;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = accA[0][0] + accB[1][3] + accA[1][2] + accB[0][5] + m1 + m2 + m3 + m4;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 512;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-60s: %.1f cycles per MFMA per wave (two waves per SIMD; 64 = matrix pipe saturated)\n", name, (double)h / iters / 16.0);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 64);
    run<0>("6 vector insts per MFMA", out, cyc);
    run<1>("+ ds_read_b128 per MFMA (operand A, one step ahead)", out, cyc);
    run<2>("+ vector insts read an older MFMA's accumulators", out, cyc);
    run<3>("ds_read_b128 per MFMA, no vector insts", out, cyc);
    run<4>("ds_read_b128 per TWO MFMAs + 6 vector insts per MFMA", out, cyc);
    return 0;
}
