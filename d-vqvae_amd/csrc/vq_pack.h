// Packed codebook image of the streaming VQ kernels (built by dvq_vq_pack in vq_stream.hip, consumed by vq_stream.hip and
// vq_stream16.hip): header, canonical |e_k|^2, fp16 image of -2 sE E in MFMA-fragment order.
#pragma once
#include "dvq_internal.h"

namespace {

constexpr int VQ_K = 512, VQ_D = 256;
struct PackHeader {
    float emax;        // upper bound of max_k |e_k|_2 (inf if the codebook is not finite)
    int sexp;          // codebook scale sE = 2^sexp
    int valid;         // 0: codebook magnitudes outside the filter's range -> every row takes the exact path
    int K, D;
    float demax;       // upper bound of max_k |e_k - image_k / (-2 sE)|_2: the image's MEASURED fp16 rounding error
    int layout;        // 3: image in MFMA-fragment order (below)
};
constexpr size_t PK_OFF_EE = 256;                          // [K] f32 canonical |e_k|^2
constexpr size_t PK_OFF_IMG = PK_OFF_EE + (size_t)VQ_K * 4;   // fp16 image, fragment order
constexpr size_t PK_BYTES = PK_OFF_IMG + (size_t)VQ_K * VQ_D * 2;

__device__ __forceinline__ float pow2f(int e) { return __int_as_float((e + 127) << 23); }   // e in [-126, 127]

// image position (in fp16 elements) of dim j of entry k: fragment f = ((w*2 + jn)*16 + s), lane = 32 h + r, element e
//   k = 64 w + 32 jn + r,  j = 16 s + 8 h + e      (lane l of wave w loads 16 B at f*1024 + 16 l: coalesced)
__host__ __device__ __forceinline__ int img_pos(int k, int j) {
    const int w = k >> 6, jn = (k >> 5) & 1, r = k & 31, s = j >> 4, h = (j >> 3) & 1, e = j & 7;
    return ((((w * 2 + jn) * 16 + s) * 64 + (h * 32 + r)) << 3) + e;
}


}  // namespace
