// What does one workgroup barrier cost?  (round 5 measured ~180 ns per barrier of the eight waves of gemm_f16x2_pp_kernel from the slope
// of launch time against K with everything but barriers and control removed.)  One workgroup per CU, W waves per workgroup; every wave
// runs REPS x (s_barrier + F filler instructions) and stamps s_memtime around it.  Variants: plain s_barrier; with s_waitcnt before it;
// with the two waves of a SIMD at different priorities; with an LDS write + read between barriers.
//   build: hipcc --offload-arch=gfx950 -O3 barrier_cost.hip -o bin/barrier_cost
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
#include <algorithm>
#define REPS 2000
template <int MODE>
__global__ void k(unsigned long long* out, float* sink, int lds_bytes_unused) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = threadIdx.x;
    const unsigned addr = (threadIdx.x * 4) & 8191;
    if (MODE == 2 && ((threadIdx.x >> 6) & 4)) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        if (MODE == 0) asm volatile("s_barrier" ::: "memory");
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_barrier" ::: "memory");
        if (MODE == 2) asm volatile("s_barrier" ::: "memory");
        if (MODE == 3) asm volatile("ds_write_b32 %1, %0\n s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b32 %0, %1 offset:256\n s_waitcnt lgkmcnt(0)\n s_barrier" : "+v"(a) : "v"(addr) : "memory");
        if (MODE == 4) asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n s_barrier" : "+v"(a) :: "memory");
        if (MODE == 5) asm volatile("s_nop 0" ::: "memory");      // the loop alone
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <int MODE>
void run(const char* what, int lds_bytes) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    hipFuncSetAttribute((const void*)&k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int W : {4, 8, 16}) {
        hipMemset(out, 0, 256 * 16 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * W), lds_bytes, 0, out, sink, 0);
        hipEventRecord(e0);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * W), lds_bytes, 0, out, sink, 0);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < W; ++w) m = std::max(m, h[b * 16 + w]); v.push_back((double)m); }
        std::sort(v.begin(), v.end());
        const double per = (MODE == 3) ? 2.0 : 1.0;
        printf("%-44s %2d waves: %7.1f shader cycles per trip (%.1f per barrier), %.1f ns per trip by the launch time\n", what, W, v[128] / REPS, v[128] / REPS / per,
               ms / 20 * 1e6 / REPS);
    }
}
int main() {
    run<5>("loop alone (s_nop)", 1024);
    run<0>("s_barrier", 1024);
    run<0>("s_barrier, 144 KB of LDS allocated", 144 * 1024);
    run<1>("s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier", 1024);
    run<2>("s_barrier, waves 4-7 at priority 1", 1024);
    run<4>("8 dependent v_add; s_barrier", 1024);
    run<3>("ds_write; wait; barrier; ds_read; wait; barrier", 16 * 1024);
    return 0;
}
