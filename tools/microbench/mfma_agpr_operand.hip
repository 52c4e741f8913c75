
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ACLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
template <int F> __device__ __forceinline__ void lda(const f16x8* p) {
    asm volatile("global_load_dwordx4 a[%c1:%c2], %0, off" :: "v"(p), "i"(4 * F), "i"(4 * F + 3) : "memory", ACLOB);
}
template <int F> __device__ __forceinline__ void mf(f32x16& acc, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(acc) : "v"(b), "i"(4 * F), "i"(4 * F + 3));
}
template <int F0, int N> struct LoadAll { static __device__ __forceinline__ void run(const f16x8* p) { lda<F0>(p + F0 * 64); LoadAll<F0 + 1, N - 1>::run(p); } };
template <int F0> struct LoadAll<F0, 0> { static __device__ __forceinline__ void run(const f16x8*) {} };
template <int FILL>
__global__ __launch_bounds__(256, 1) void k(const f16x8* __restrict__ img, const f16x8* __restrict__ zt, float* out, int ntiles, unsigned long long* clk) {
    __shared__ f16x8 lds[2][16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LoadAll<0, 64>::run(img + wave * 64 * 64 + lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x16 acc[4];
    float m0 = 1e30f, m1 = 1e30f;
    float sc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = (float)(lane * 16 + e);
    for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[0][i] = zt[i];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles; ++t) {
        const f16x8* L = lds[0];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        f16x8 bfn = L[lane];
#define STEP(S) { const f16x8 bf = bfn; bfn = L[((S + 1) & 15) * 64 + lane]; \
        mf<0 * 16 + S>(acc[0], bf); if (FILL) { _Pragma("unroll") for (int e = 0; e < FILL; ++e) { m1 = __builtin_amdgcn_fmed3f(m0, m1, sc[(S + e) & 15]); m0 = __builtin_amdgcn_fmed3f(m0, m1, -3e38f); } asm volatile("" : "+v"(m0), "+v"(m1)); } \
        mf<1 * 16 + S>(acc[1], bf); if (FILL) { _Pragma("unroll") for (int e = 0; e < FILL; ++e) { m1 = __builtin_amdgcn_fmed3f(m0, m1, sc[(S + e + 3) & 15]); m0 = __builtin_amdgcn_fmed3f(m0, m1, -3e38f); } asm volatile("" : "+v"(m0), "+v"(m1)); } \
        mf<2 * 16 + S>(acc[2], bf); if (FILL) { _Pragma("unroll") for (int e = 0; e < FILL; ++e) { m1 = __builtin_amdgcn_fmed3f(m0, m1, sc[(S + e + 5) & 15]); m0 = __builtin_amdgcn_fmed3f(m0, m1, -3e38f); } asm volatile("" : "+v"(m0), "+v"(m1)); } \
        mf<3 * 16 + S>(acc[3], bf); }
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    float s = m0 + m1;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[((blockIdx.x * 4 + b) * 16 + e) * 256 + threadIdx.x] = acc[b][e] + (b == 0 && e == 0 ? s * 0.f : 0.f);
}

template <int NOP, int FILL>
__global__ __launch_bounds__(256, 1) void kb(const f16x8* __restrict__ img, const f16x8* __restrict__ zt, float* out, int ntiles, unsigned long long* clk) {
    __shared__ f16x8 lds[16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LoadAll<0, 64>::run(img + wave * 64 * 64 + lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = threadIdx.x; i < 16 * 64; i += 256) lds[i] = zt[i];
    __syncthreads();
    f16x8 bf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) bf[s] = lds[s * 64 + lane];
    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ntiles; ++t) {
    float m0 = 1e30f, m1 = 1e30f, sc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = (float)(lane * 16 + e);
#define MFN(F, A, S) { if (FILL) { _Pragma("unroll") for (int e_ = 0; e_ < FILL; ++e_) { const float p_ = __uint_as_float((__float_as_uint(sc[(S + e_) & 15]) & ~31u) | (unsigned)e_); m1 = __builtin_amdgcn_fmed3f(m0, m1, p_); m0 = __builtin_amdgcn_fmed3f(m0, p_, -3e38f); } asm volatile("" : "+v"(m0), "+v"(m1)); } if (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, a[%c2:%c3], %1, %0" : "+v"(A) : "v"(bf[S]), "i"(4 * (F)), "i"(4 * (F) + 3)); else mf<F>(A, bf[S]); __builtin_amdgcn_sched_barrier(0); }
#define BLK(B, A) MFN(B*16+0,A,0) MFN(B*16+1,A,1) MFN(B*16+2,A,2) MFN(B*16+3,A,3) MFN(B*16+4,A,4) MFN(B*16+5,A,5) MFN(B*16+6,A,6) MFN(B*16+7,A,7) MFN(B*16+8,A,8) MFN(B*16+9,A,9) MFN(B*16+10,A,10) MFN(B*16+11,A,11) MFN(B*16+12,A,12) MFN(B*16+13,A,13) MFN(B*16+14,A,14) MFN(B*16+15,A,15)
        BLK(0, acc[0]) BLK(1, acc[1]) BLK(2, acc[0]) BLK(3, acc[1])
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[((blockIdx.x * 4 + b) * 16 + e) * 256 + threadIdx.x] = acc[b][e];
}
int main() {
    const int NT = 64;
    std::vector<_Float16> himg(4 * 64 * 64 * 8), hz(1024 * 8);
    for (size_t i = 0; i < himg.size(); ++i) himg[i] = (_Float16)(((int)(i * 7 % 13) - 6) * 0.125f);
    for (size_t i = 0; i < hz.size(); ++i) hz[i] = (_Float16)(((int)(i * 5 % 11) - 5) * 0.25f);
    f16x8 *img, *zt; float* out; unsigned long long* clk;
    hipMalloc(&img, himg.size() * 2); hipMalloc(&zt, hz.size() * 2); hipMalloc(&out, 256 * 64 * 256 * 4); hipMalloc(&clk, 256 * 8);
    hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice); hipMemcpy(zt, hz.data(), hz.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk);
        hipDeviceSynchronize();
        unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("FILL=0: %.1f memtime ticks per MFMA\n", (double)c / (NT * 64));
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk);
        hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("FILL=2 (4 VALU/gap): %.1f ticks per MFMA\n", (double)c / (NT * 64));
        hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk);
        hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("FILL=3 (6 VALU/gap): %.1f ticks per MFMA\n", (double)c / (NT * 64));
        hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk);
        hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("FILL=4 (8 VALU/gap): %.1f ticks per MFMA\n", (double)c / (NT * 64));
    }
    {
        unsigned long long c;
        hipLaunchKernelGGL((kb<0, 2>), dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk); hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("dependent chain + 2 scores (6 VALU) per gap: %.1f ticks per MFMA\n", (double)c / (NT * 64));
        hipLaunchKernelGGL((kb<0, 1>), dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk); hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("dependent chain + 1 score (3 VALU) per gap: %.1f ticks per MFMA\n", (double)c / (NT * 64));
        hipLaunchKernelGGL((kb<0, 0>), dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk); hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("block-major chain of 16 on one accumulator: %.1f ticks per MFMA\n", (double)c / (NT * 64));
        hipLaunchKernelGGL((kb<1, 0>), dim3(256), dim3(256), 0, 0, img, zt, out, NT, clk); hipDeviceSynchronize(); hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("the same with s_nop 1 in front of every MFMA: %.1f ticks per MFMA\n", (double)c / (NT * 64));
        unsigned long long r0 = 0, r1 = 0;
    }
    // correctness of block 0, wave 0, acc[0]: D[i][j] = sum_k A[i][k] B[k][j] for the last tile
    std::vector<float> ho(16 * 256);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, img, zt, out, 1, clk); hipDeviceSynchronize();
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int e = 0; e < 16; ++e) for (int lane = 0; lane < 64; ++lane) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), j = lane & 31;
        double ref = 0;
        for (int s = 0; s < 16; ++s) for (int hh = 0; hh < 2; ++hh) for (int q = 0; q < 8; ++q) {
            const float a = (float)himg[((0 * 64 + s) * 64 + (hh * 32 + i)) * 8 + q];     // frag s of wave 0 block 0: lane (hh, i) holds A[i][16s+8hh+q]
            const float b = (float)hz[(s * 64 + (hh * 32 + j)) * 8 + q];
            ref += (double)a * b;
        }
        maxerr = fmax(maxerr, fabs(ref - ho[e * 256 + lane]));
    }
    printf("max |err| of acc[0] vs host: %g\n", maxerr);
    return 0;
}
