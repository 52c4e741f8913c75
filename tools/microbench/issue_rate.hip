// How many instructions per cycle does ONE SIMD issue from 1, 2 or 4 resident waves, by instruction mix?  (round 6: vq_pipe.hip's
// periods take ~4.35 cycles per instruction summed over a SIMD's four waves -- is that the front end or the kernel?)
// One workgroup per CU, 64 * 4 * W threads (W waves per SIMD); every wave runs REPS x 64 instructions of straight-line code without
// dependences between neighbours and stamps s_memtime around it.   build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o bin/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REPS 64
// 8 independent VALU ops on 8 registers; SALU ops on 8 scalars; LDS reads into 4 registers (waited once per block of 64)
#define V8 "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7\n"
#define S8 "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
#define VS8 "v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %1\n s_add_u32 s21, s21, 1\n v_add_f32 %2, %2, %2\n s_add_u32 s22, s22, 1\n v_add_f32 %3, %3, %3\n s_add_u32 s23, s23, 1\n"
#define VSL8 "v_add_f32 %0, %0, %0\n s_add_u32 s20, s20, 1\n ds_read_b32 %4, %8\n v_add_f32 %1, %1, %1\n s_add_u32 s21, s21, 1\n ds_read_b32 %5, %8 offset:256\n v_add_f32 %2, %2, %2\n s_add_u32 s22, s22, 1\n"
#define NOP8 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"
#define VD8 "v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n v_add_f32 %0, %0, %0\n"
template <int MIX>
__global__ void k(unsigned long long* out, float* sink) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e = a + 4, f = a + 5, g = a + 6, h = a + 7;
    unsigned addr = (threadIdx.x & 63) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REPS; ++r) {
#define BODY(T) asm volatile(T T T T T T T T "s_waitcnt lgkmcnt(0)\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(addr) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc")
        if (MIX == 0) BODY(V8); else if (MIX == 1) BODY(S8); else if (MIX == 2) BODY(VS8); else if (MIX == 3) BODY(VSL8); else if (MIX == 4) BODY(NOP8);
        else BODY(VD8);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
}
template <int MIX>
void run(const char* what) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 16 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    for (int W : {1, 2, 4}) {
        hipMemset(out, 0, 256 * 16 * 8);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(256 * W), 0, 0, out, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int b = 0; b < 256; ++b) { unsigned long long m = 0; for (int w = 0; w < 4 * W; ++w) m = std::max(m, h[b * 16 + w]); v.push_back((double)m); }
        std::sort(v.begin(), v.end());
        const double cyc = v[128], n = (double)REPS * 64 * W;       // instructions per SIMD
        printf("%-34s W=%d waves/SIMD: %8.0f cycles for %6.0f instructions per SIMD -> %.2f cycles per instruction (per wave %.2f)\n", what, W, cyc, n, cyc / n, cyc / (REPS * 64.0));
    }
}
int main() {
    run<0>("VALU (independent)");
    run<5>("VALU (dependent chain)");
    run<1>("SALU");
    run<2>("VALU / SALU alternating");
    run<3>("VALU / SALU / LDS read");
    run<4>("s_nop 0");
    return 0;
}
