// Microbenchmark (GPU box): how many vector instructions fit beside one v_mfma_f32_32x32x16_f16 on one SIMD before the
// loop slows down, for one and two waves per SIMD, and per instruction kind (the VQ filter's score absorption is
// fma + perm + 4 x med3 per MFMA).   hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue mfma_valu_coissue.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc0 = {0}, acc1 = {0};
    float m1 = 1e30f, m2 = 1e30f, m3 = 1e30f, m4 = 1e30f, p = threadIdx.x * 1.0f, q = 0.5f;
    unsigned u = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (KIND == 0) {            // four independent med3 chains (top-4 style: each reads the previous p)
                m4 = __builtin_amdgcn_fmed3f(m3, m4, p); m3 = __builtin_amdgcn_fmed3f(m2, m3, p);
                m2 = __builtin_amdgcn_fmed3f(m1, m2, p); m1 = __builtin_amdgcn_fmed3f(m1, p, -3e38f);
                p += 1.0f;
            } else if (KIND == 1) {     // independent fmas
                m1 = fmaf(m1, q, p); m2 = fmaf(m2, q, p); m3 = fmaf(m3, q, p); m4 = fmaf(m4, q, p);
            } else if (KIND == 2) {     // perms
                u = __builtin_amdgcn_perm(u, __float_as_uint(m1), 0x07060500u); m1 += 1.0f;
                u = __builtin_amdgcn_perm(u, __float_as_uint(m2), 0x07060501u); m2 += 1.0f;
            } else {                    // VOP2 min / max
                m1 = fminf(m1, p); m2 = fmaxf(m2, p); m3 = fminf(m3, q); m4 = fmaxf(m4, q);
            }
        }
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        asm volatile("" : "+v"(m1), "+v"(m2), "+v"(m3), "+v"(m4), "+v"(u));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3] + m1 + m2 + m3 + m4 + (float)u;
}

template <int NV, int KIND>
void run(const char* name, int blocks_per_cu, float* out, unsigned long long* cyc) {
    const int iters = 4096;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // 2 MFMAs and (KIND 0/1/3: 4 NV, KIND 2: 4 NV) vector instructions per iteration per wave
    printf("%-8s %d wave(s)/SIMD  NV=%d : %.1f cycles per MFMA per wave (vector insts per MFMA: %d)\n", name, blocks_per_cu, NV,
           (double)h / iters / 2.0, 2 * NV);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4 * 2); hipMalloc(&cyc, 64);
    for (int bp = 1; bp <= 2; ++bp) {
        run<0, 0>("none", bp, out, cyc);
        run<1, 0>("med3", bp, out, cyc); run<2, 0>("med3", bp, out, cyc); run<3, 0>("med3", bp, out, cyc); run<4, 0>("med3", bp, out, cyc);
        run<1, 1>("fma", bp, out, cyc); run<2, 1>("fma", bp, out, cyc); run<3, 1>("fma", bp, out, cyc); run<4, 1>("fma", bp, out, cyc);
        run<2, 2>("perm", bp, out, cyc); run<3, 2>("perm", bp, out, cyc);
        run<2, 3>("minmax", bp, out, cyc); run<3, 3>("minmax", bp, out, cyc); run<4, 3>("minmax", bp, out, cyc);
    }
    return 0;
}
