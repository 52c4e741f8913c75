#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny, int flush) {
  if (flush) __builtin_amdgcn_s_setreg((2 - 1) << 11 | 6 << 6 | 1, 0);
  _Float16 h = (_Float16)tiny;          // 1e-6 -> fp16 subnormal (or 0 when flushed)
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
  // subnormal bit pattern forced in (not through a conversion): 0x0010 = 16 * 2^-24
  unsigned short bits = 0x0010; _Float16 sub = __builtin_bit_cast(_Float16, bits);
  a[0] = sub; b[0] = (_Float16)1024.f;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = (float)h; out[1] = c[0]; out[2] = (float)sub; }
}
int main() {
  float* d; hipMalloc(&d, 64); float h[3];
  for (int flush = 0; flush < 2; ++flush) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1e-6f, flush);
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("flush=%d: cvt(1e-6)=%g  mfma(sub*1024)=%g (expect %g if kept)  f32(sub)=%g\n", flush, h[0], h[1], 16.0/16777216.0*1024.0, h[2]);
  }
  return 0;
}
