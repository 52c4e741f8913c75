// LDS read bandwidth of one CU for the fragment-read pattern of gemm_f16x2_pp_kernel: every wave reads 16 x ds_read_b128 (1 KB each,
// a contiguous 1 KB block per instruction: 16 rows of 64 B, lane & 15 = row, lane >> 4 = 16-byte chunk, XOR-swizzled), waits, repeats.
// W waves per workgroup (one workgroup per CU).  Ideal at 128 B/clk/CU: W x 16 KB / 128 = 128 W cycles per trip.
//   build: hipcc --offload-arch=gfx950 -O3 lds_read_rate.hip -o bin/lds_read_rate
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
#include <algorithm>
#define REPS 1000
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(unsigned long long* out, float* sink) {
    extern __shared__ char lds[];
    for (int i = threadIdx.x; i < 36 * 1024; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned rd = (lane & 15) * 64 + 16 * ((lane >> 4) ^ ((lane & 15) >> 2 & 3));
    if (MODE == 1) rd = lane * 16;                                   // plain contiguous
    if (MODE == 2) rd = (lane & 15) * 64 + 16 * (lane >> 4);         // no swizzle
    const unsigned base = (unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds + rd + (wave & 3) * 4096;
    f32x4 a0, a1, a2, a3, a4, a5, a6, a7;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:16384\n ds_read_b128 %3, %8 offset:17408\n"
                     "ds_read_b128 %4, %8 offset:8192\n ds_read_b128 %5, %8 offset:9216\n ds_read_b128 %6, %8 offset:32768\n ds_read_b128 %7, %8 offset:33792\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     "ds_read_b128 %0, %8 offset:2048\n ds_read_b128 %1, %8 offset:3072\n ds_read_b128 %2, %8 offset:18432\n ds_read_b128 %3, %8 offset:19456\n"
                     "ds_read_b128 %4, %8 offset:10240\n ds_read_b128 %5, %8 offset:11264\n ds_read_b128 %6, %8 offset:34816\n ds_read_b128 %7, %8 offset:35840\n"
                     "s_waitcnt lgkmcnt(0)\n"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(base) : "memory");
        acc += a0[0] + a7[3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* what) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    hipFuncSetAttribute((const void*)&k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int W : {1, 4, 8}) {
        hipMemset(out, 0, 256 * 16 * 8);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * W), 144 * 1024, 0, out, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < W; ++w) m = std::max(m, h[b * 16 + w]); v.push_back((double)m); }
        std::sort(v.begin(), v.end());
        const double cyc = v[128] / REPS;
        printf("%-28s %d waves: %7.1f cycles per trip of 16 reads per wave = %5.1f B/clk per CU (ideal 128)\n", what, W, cyc, W * 16384.0 / cyc);
    }
}
int main() {
    run<0>("GEMM pattern (swizzled)");
    run<2>("rows of 64 B, no swizzle");
    run<1>("lane * 16 contiguous");
    return 0;
}
