// Are packed-fp32 vector instructions (v_pk_mul_f32 with its op_sel forms) reliable when a second wave shares the SIMD?
// Background: pn_trunk_filter_kernel (two workgroups per CU) published wrong top-three records in lanes 48-63 once per ~1e6
// (tile, channel) pairs; the records were right whenever the compiler emitted no v_pk_* instructions (-fno-slp-vectorize).
// 512-thread workgroups (two waves per SIMD).  Waves 0-3 ("victims") run the instruction sequence of the failing code
// (broadcast forms op_sel:[0,1] / op_sel_hi:[1,0] / op_sel_hi:[0,1], results consumed after 0..3 independent instructions)
// and compare every result with plain v_mul_f32; waves 4-7 ("aggressors") run nothing / MFMA chains / packed multiplies /
// LDS traffic.  Mismatches are counted per lane quarter.
//   hipcc --offload-arch=gfx950 -O3 -o pk_mul_two_waves pk_mul_two_waves.hip && ./pk_mul_two_waves
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float mul1(float a, float b) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float val(unsigned x) {          // a float in [1, 2) from a hash
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    return __uint_as_float(0x3f800000u | (x >> 9));
}

template <int GAP>
__device__ __forceinline__ void victim_step(float ti, f2 s01, f2 s23, f2 q01, f2 q23, f2 q45, f2& r0, f2& r1, f2& r2) {
    f2 tip = {ti, 0.f}, f01, f23;
    // f = ti * s (op_sel_hi:[0,1] broadcasts ti), then q * f with the other two broadcast forms; GAP independent instructions between
    if (GAP == 0)
        asm volatile("v_pk_mul_f32 %0, %5, %6 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %5, %7 op_sel_hi:[0,1]\n"
                     "v_pk_mul_f32 %2, %8, %0 op_sel:[0,1]\n v_pk_mul_f32 %3, %9, %1\n v_pk_mul_f32 %4, %10, %0 op_sel_hi:[1,0]"
                     : "=&v"(f01), "=&v"(f23), "=&v"(r0), "=&v"(r1), "=&v"(r2)
                     : "v"(tip), "v"(s01), "v"(s23), "v"(q01), "v"(q23), "v"(q45));
    else if (GAP == 1)
        asm volatile("v_pk_mul_f32 %0, %5, %6 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %5, %7 op_sel_hi:[0,1]\n s_nop 0\n"
                     "v_pk_mul_f32 %2, %8, %0 op_sel:[0,1]\n v_pk_mul_f32 %3, %9, %1\n v_pk_mul_f32 %4, %10, %0 op_sel_hi:[1,0]"
                     : "=&v"(f01), "=&v"(f23), "=&v"(r0), "=&v"(r1), "=&v"(r2)
                     : "v"(tip), "v"(s01), "v"(s23), "v"(q01), "v"(q23), "v"(q45));
    else
        asm volatile("v_pk_mul_f32 %0, %5, %6 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %5, %7 op_sel_hi:[0,1]\n s_nop 7\n"
                     "v_pk_mul_f32 %2, %8, %0 op_sel:[0,1]\n s_nop 7\n v_pk_mul_f32 %3, %9, %1\n s_nop 7\n v_pk_mul_f32 %4, %10, %0 op_sel_hi:[1,0]\n s_nop 7"
                     : "=&v"(f01), "=&v"(f23), "=&v"(r0), "=&v"(r1), "=&v"(r2)
                     : "v"(tip), "v"(s01), "v"(s23), "v"(q01), "v"(q23), "v"(q45));
}

template <int AGG, int GAP>
__global__ __launch_bounds__(512, 1) void k(unsigned long long* bad, float* sink, int iters) {
    __shared__ float lds[8][64 * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long nbad = 0;
    float keep = 0.f;
    if (wave < 4) {
        for (int it = 0; it < iters; ++it) {
            const unsigned h = (blockIdx.x * 131071u + it) * 64u + lane;
            const float ti = val(h * 11u + 1u);
            const f2 s01 = {val(h * 11u + 2u), val(h * 11u + 3u)}, s23 = {val(h * 11u + 4u), val(h * 11u + 5u)};
            const f2 q01 = {val(h * 11u + 6u), val(h * 11u + 7u)}, q23 = {val(h * 11u + 8u), val(h * 11u + 9u)};
            const f2 q45 = {val(h * 11u + 10u), val(h * 11u + 12u)};
            f2 r0, r1, r2;
            victim_step<GAP>(ti, s01, s23, q01, q23, q45, r0, r1, r2);
            const float f0 = mul1(ti, s01[0]), f1 = mul1(ti, s01[1]), f2_ = mul1(ti, s23[0]), f3 = mul1(ti, s23[1]);
            const bool ok = r0[0] == mul1(q01[0], f1) && r0[1] == mul1(q01[1], f1) && r1[0] == mul1(q23[0], f2_) && r1[1] == mul1(q23[1], f3) &&
                            r2[0] == mul1(q45[0], f0) && r2[1] == mul1(q45[1], f0);
            nbad += ok ? 0 : 1;
        }
        if (nbad) atomicAdd(&bad[lane >> 4], nbad);
    } else if (AGG == 1) {                                   // MFMA chains
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.25f * ((lane + i) % 7 - 3)); b[i] = (_Float16)(0.5f * ((lane * 3 + i) % 5 - 2)); }
        f32x16 acc = {0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        for (int e = 0; e < 16; ++e) keep += acc[e];
    } else if (AGG == 2) {                                   // packed multiplies
        f2 x = {1.0f + lane, 2.0f}, y = {1.0001f, 0.9999f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        keep = x[0] + x[1];
    } else if (AGG == 3) {                                   // LDS traffic
        float x = lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) lds[wave][u * 64 + lane] = x + u;
#pragma unroll
            for (int u = 0; u < 8; ++u) x += lds[wave][u * 64 + ((lane + 1) & 63)];
        }
        keep = x;
    } else if (AGG == 4) {                                   // plain vector work
        float x = lane;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, 1.0001f, 0.5f);
        keep = x;
    }
    sink[blockIdx.x * 512 + threadIdx.x] = keep;
}

template <int AGG, int GAP>
void run(const char* name, unsigned long long* bad, float* sink, int iters) {
    (void)hipMemset(bad, 0, 4 * sizeof(unsigned long long));
    hipLaunchKernelGGL((k<AGG, GAP>), dim3(1024), dim3(512), 0, 0, bad, sink, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
    printf("%-40s gap %d: wrong lane-results by lane quarter %llu %llu %llu %llu  of %.3g per quarter\n", name, GAP, h[0], h[1], h[2], h[3],
           1024.0 * 4 * iters * 16);
}

int main() {
    unsigned long long* bad;
    float* sink;
    (void)hipMalloc(&bad, 64);
    (void)hipMalloc(&sink, 1024 * 512 * 4);
    const int iters = 20000;
    run<0, 0>("other wave: idle", bad, sink, iters);
    run<1, 0>("other wave: MFMA chains", bad, sink, iters);
    run<2, 0>("other wave: packed multiplies", bad, sink, iters);
    run<3, 0>("other wave: LDS traffic", bad, sink, iters);
    run<4, 0>("other wave: vector fma", bad, sink, iters);
    run<1, 1>("other wave: MFMA chains", bad, sink, iters);
    run<2, 1>("other wave: packed multiplies", bad, sink, iters);
    run<1, 2>("other wave: MFMA chains", bad, sink, iters);
    run<2, 2>("other wave: packed multiplies", bad, sink, iters);
    return 0;
}
