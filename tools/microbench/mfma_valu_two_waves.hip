// Do a matrix-only wave and a vector-only wave on the SAME SIMD run concurrently?  512-thread workgroups (two waves per SIMD),
// one per CU: waves 0-3 issue dependent v_mfma_f32_32x32x16_f16 chains, waves 4-7 vector instruction chains; each alone, then together.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int KIND>   // MODE bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run vector work.  KIND 0: med3/min chains, 1: independent fma
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.25f * ((lane + i) % 7 - 3)); b[i] = (_Float16)(0.5f * ((lane * 3 + i) % 5 - 2)); }
    f32x16 acc = {0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    float m1 = 1e30f, m2 = 1e30f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (MODE & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            }
    } else {
        if (MODE & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    if (KIND == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float p = __uint_as_float((__float_as_uint(v[e]) & ~31u) | (unsigned)e);
                            m2 = __builtin_amdgcn_fmed3f(m1, m2, p);
                            m1 = __builtin_amdgcn_fmed3f(m1, p, -3e38f);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 0.9999f, -0.5f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.25f);
                    }
                    asm volatile("" : "+v"(m1), "+v"(m2), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
    float s = m1 + m2;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int e = 0; e < 16; ++e) s += acc[e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE, int KIND>
void run(const char* name, float* out, unsigned long long* clk, int iters) {
    hipLaunchKernelGGL((k<MODE, KIND>), dim3(256), dim3(512), 0, 0, out, iters, clk);
    (void)hipDeviceSynchronize();
    unsigned long long h[8];
    (void)hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s matrix wave 0: %8.1f cycles per 16 MFMAs   vector wave 4: %8.1f cycles per block (24 instr)\n", name,
           (double)h[0] / iters, (double)h[4] / iters);
}
int main() {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 8 * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 0>("MFMA waves alone", out, clk, iters);
        run<2, 0>("vector waves alone (med3 chains)", out, clk, iters);
        run<3, 0>("both (med3 chains)", out, clk, iters);
        run<2, 1>("vector waves alone (independent fma)", out, clk, iters);
        run<3, 1>("both (independent fma)", out, clk, iters);
    }
    return 0;
}
