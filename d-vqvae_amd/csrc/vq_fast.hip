// Fast VQ nearest-codebook-entry for the headline shape (K = 512, D = 256): same indices as the exact
// kernel (vq.hip / oracle/vq_canonical.c), bit for bit, at a fraction of the fp32 contraction cost.
//
//   filter (vq_filter_kernel): fp16 MFMA scores  s_k = ee_k + (sum_j h(sz z_j) h(-2 sE e_kj)) / (sz sE)
//       for all 512 entries of every row (h = round to fp16; sz per row and sE per codebook are powers of two
//       that put the largest magnitude in [2^13, 2^14)).  Every lane keeps the FOUR smallest scores it has seen
//       (entry id packed into the low 8 mantissa bits); a row's candidates are the scores within a PROVEN eps_row
//       of the row minimum, so the exact fp32 argmin is always among them.  One candidate -> decided; else
//   refine (vq_refine_kernel): the canonical fp32 evaluation d_k = (zz + ee_k) - 2*dot_k (k-ordered fmaf
//       chains) of the <= 6 candidates, torch.argmin ordering (first minimum, NaN first).  Rows whose
//       candidate set is empty (NaN/Inf, magnitudes outside 2^+-40) or may be incomplete (a lane's fourth-smallest
//       score is still within eps) are evaluated over all K entries.
//
// Error bound.  With u = 2^-12 (fp16 round-to-nearest; elements below the fp16 normal range add at most
// 2^-20 |z||e| in total, flushed or not), for every k
//   |s_k - (true_k - |z|^2)| <= 2 (2u + u^2 + 2^-20) |z||e_k| + gamma_258 (|e_k|^2 + 2|z||e_k|)   (filter)
//   |d_k - true_k|           <= gamma_260 (|z| + |e_k|)^2                                          (exact side)
//   |packed(s_k) - s_k|      <= 2^-15 |s_k| <= 2^-15 (|z| + |e_k|)^2                               (id in the mantissa)
// hence for the exact winner k*:  packed(s_k*) <= min_k packed(s_k) + eps_row,
//   eps_row = 2^-9 (1 + 2^-8) |z| Emax + (2^-13 + 2^-14) (|z| + Emax)^2     (gamma_n = n 2^-24, with slack).
//
// Structure (the fused PointNet trunk's, pointnet.hip): a wave owns 32 rows; their fp16 fragments (64 VGPRs) are
// the MFMA B operand for the whole kernel; the codebook image (256 KB, L2-resident) streams L2 -> LDS by
// global_load_lds in 32 KB chunks of 64 entries (XOR-swizzled source, conflict-free ds_read_b128), double buffered,
// one barrier per chunk; scores land with the row on the lane and the entry on the registers, so the top-3 tracking
// is lane-local VALU work interleaved with the next chunk's MFMAs.  128 rows per 256-thread workgroup, two
// workgroups per CU.  z is read from HBM exactly once (fragment-shaped loads straight to registers).
// Algorithmic HBM bytes per row: D*4 (z) + 8 (int64 index); the codebook (K*D*4) is read once.
#include "dvq_internal.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 512, D = 256;
constexpr int WG_ROWS = 128;                       // 4 waves x 32 rows
constexpr int KS = D / 16;                         // 16 MFMA k-steps
constexpr int NCHUNK = K / 64;                     // 8 chunks of 64 entries
constexpr int MAXC = 8;                            // candidate slots per row in the hand-off record (<= 6 used)
constexpr int EXP_LIMIT = 40;                      // |log2(max magnitude)| beyond this -> exact fallback

constexpr int CHUNK_B = 64 * D * 2;                // 32 768 B: 64 entries x 256 fp16
constexpr int OFF_EE = 2 * CHUNK_B;                // [K] fp32
constexpr int OFF_CNT = OFF_EE + K * 4;            // [4] ints (per-wave ambiguous counts)
constexpr int LDS_BYTES = OFF_CNT + 64;            // 67 648 B -> two workgroups per CU

struct PackHeader {
    float emax;        // upper bound of max_k |e_k|_2 (inf if the codebook is not finite)
    int sexp;          // codebook scale sE = 2^sexp
    int valid;         // 0: codebook magnitudes outside the filter's range -> every row takes the exact path
    int K, D;
};
constexpr size_t PK_OFF_EE = 256;
constexpr size_t PK_OFF_IMG = PK_OFF_EE + (size_t)K * 4;
constexpr size_t PK_BYTES = PK_OFF_IMG + (size_t)K * D * 2;

__device__ __forceinline__ float pow2f(int e) { return __int_as_float((e + 127) << 23); }   // e in [-126, 127]

// ------------------------------------------------------------------------------------------------ pack
__global__ void vq_pack_norm_kernel(const float* __restrict__ E, float* __restrict__ ee, PackHeader* hdr) {
    __shared__ float red[K];
    __shared__ float redm[K];
    const int k = threadIdx.x;                                   // blockDim = K
    const float* p = E + k * D;
    float acc = 0.f, mx = 0.f;
    bool finite = true;
    for (int j = 0; j < D; ++j) {
        acc = fmaf(p[j], p[j], acc);                             // canonical chain (same as rownorm_kernel)
        finite = finite && (fabsf(p[j]) <= 3.0e38f);
        mx = fmaxf(mx, fabsf(p[j]));
    }
    ee[k] = acc;
    red[k] = (finite && acc <= 3.0e38f) ? acc : INFINITY;
    redm[k] = finite ? mx : INFINITY;
    __syncthreads();
    for (int o = K / 2; o > 0; o >>= 1) {
        if (k < o) {
            red[k] = fmaxf(red[k], red[k + o]);
            redm[k] = fmaxf(redm[k], redm[k + o]);
        }
        __syncthreads();
    }
    if (k == 0) {
        hdr->emax = sqrtf(red[0]) * 1.00001f;
        const float m2 = 2.0f * redm[0];                         // the image holds -2 e
        const int e = (int)((__float_as_uint(m2) >> 23) & 0xff) - 127;
        const bool ok = redm[0] > 0.f && e >= -EXP_LIMIT && e <= EXP_LIMIT && red[0] <= 3.0e38f;
        hdr->sexp = ok ? 13 - e : 0;
        hdr->valid = ok ? 1 : 0;
        hdr->K = K;
        hdr->D = D;
    }
}

// image: [K][D] fp16 of -2 sE e, natural order
__global__ void vq_pack_img_kernel(const float* __restrict__ E, const PackHeader* __restrict__ hdr, _Float16* __restrict__ img) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= K * D) return;
    img[gid] = (_Float16)(-2.0f * pow2f(hdr->sexp) * E[gid]);
}

// ------------------------------------------------------------------------------------------------ filter
// DMA one chunk (64 entries x 512 B) into an LDS stage: 32 pieces of 1 KiB (two rows each), eight per wave; a row has 32
// 16-byte chunks, chunk c of row r lands at chunk c ^ (r & 15)
__device__ __forceinline__ void issue_chunk(const _Float16* __restrict__ img, int chunk, char* stage, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rp = wave * 8 + i;                              // row pair
        const int row = 2 * rp + (lane >> 5);
        const _Float16* src = img + (long)(64 * chunk + row) * D + 8 * ((lane & 31) ^ (row & 15));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + rp * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ f16x8 efrag(const char* stage, int row, int chunk) {
    return *reinterpret_cast<const f16x8*>(stage + row * 512 + 16 * (chunk ^ (row & 15)));
}

// keep the four smallest of (m1 <= m2 <= m3 <= m4) and s
__device__ __forceinline__ void top4(float s, float& m1, float& m2, float& m3, float& m4) {
    m4 = __builtin_amdgcn_fmed3f(m3, m4, s);
    m3 = __builtin_amdgcn_fmed3f(m2, m3, s);
    m2 = __builtin_amdgcn_fmed3f(m1, m2, s);
    m1 = fminf(m1, s);
}

// scores of one chunk half (one 32-entry MFMA tile): s = ee + acc * inv, entry id into the low 8 mantissa bits, top-3 update
__device__ __forceinline__ void absorb(const f32x16& acc, const float* __restrict__ ee_s, int c, int jn, int h, float inv,
                                       float& m1, float& m2, float& m3, float& m4) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {                                 // regs 4g..4g+3 <-> entries 64c + 32jn + 8g + 4h + 0..3
        const f32x4 ev = *reinterpret_cast<const f32x4*>(ee_s + 64 * c + 32 * jn + 8 * g + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * g + i;
            const float s = fmaf(acc[e], inv, ev[i]);
            const float p = __uint_as_float((__float_as_uint(s) & 0xffffff00u) | (unsigned)(c << 5 | jn << 4 | e));
            top4(p, m1, m2, m3, m4);
        }
    }
}

__device__ __forceinline__ int decode_entry(float packed, int half) {
    const unsigned id = __float_as_uint(packed) & 0xffu;
    const int c = id >> 5, jn = (id >> 4) & 1, e = id & 15;
    return 64 * c + 32 * jn + (e & 3) + 8 * (e >> 2) + 4 * half;
}

__global__ __launch_bounds__(256, 2) void vq_filter_kernel(const float* __restrict__ z, long M, const char* __restrict__ packed,
                                                           int64_t* __restrict__ idx, uint16_t* __restrict__ cand_out,
                                                           uint8_t* __restrict__ cnt_out, int* __restrict__ amb_count,
                                                           int* __restrict__ amb_list) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* ee_s = reinterpret_cast<float*>(lds + OFF_EE);
    int* wcnt = reinterpret_cast<int*>(lds + OFF_CNT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const PackHeader* hdr = reinterpret_cast<const PackHeader*>(packed);
    const float emax = hdr->emax;
    const int e_sexp = hdr->sexp;
    const bool e_valid = hdr->valid != 0;
    const float* ee_g = reinterpret_cast<const float*>(packed + PK_OFF_EE);
    const _Float16* img = reinterpret_cast<const _Float16*>(packed + PK_OFF_IMG);

    // Two workgroups share a CU; both would stream their rows from HBM first and multiply afterwards, in lockstep.  The
    // second half of the grid starts ~4.5 us late so that its loads run under the first half's matrix work (speed only).
    if (blockIdx.x >= (gridDim.x + 1) / 2 && gridDim.x > 256) __builtin_amdgcn_s_sleep(127);
    // ---- this lane's row: 128 of its 256 floats (k = 16 s + 8 h + j), straight from HBM in fragment shape
    long grow = (long)blockIdx.x * WG_ROWS + wave * 32 + r;
    const bool live = grow < M;
    if (!live) grow = M - 1;
    const float* zr = z + grow * D + 8 * h;
    f32x4 zf[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        zf[s][0] = *reinterpret_cast<const f32x4*>(zr + 16 * s);
        zf[s][1] = *reinterpret_cast<const f32x4*>(zr + 16 * s + 4);
    }
    ee_s[tid] = ee_g[tid];
    ee_s[tid + 256] = ee_g[tid + 256];
    float mx = 0.f, ss = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mx = fmaxf(mx, fabsf(zf[s][q][i]));
                ss = fmaf(zf[s][q][i], zf[s][q][i], ss);
            }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    ss += __shfl_xor(ss, 32);
    const bool bad = !(ss <= 3.0e38f);                          // NaN / Inf anywhere in the row poisons the sum of squares
    const int ez = (int)((__float_as_uint(mx) >> 23) & 0xff) - 127;
    const bool ok = e_valid && !bad && (mx == 0.f || (ez >= -EXP_LIMIT && ez <= EXP_LIMIT));
    const int zs = (mx == 0.f || !ok) ? 0 : 13 - ez;
    const float sc = pow2f(zs);
    const float inv = ok ? pow2f(-(zs + e_sexp)) : __int_as_float(0x7fc00000);     // NaN marks "exact path"
    const float zn = __builtin_amdgcn_sqrtf(ss) * 1.0001f;
    f16x8 zh[KS];                                                // MFMA B operand: B[k = 8h + j][col = row r]
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 v;
        v[0] = (_Float16)(zf[s][0][0] * sc); v[1] = (_Float16)(zf[s][0][1] * sc);
        v[2] = (_Float16)(zf[s][0][2] * sc); v[3] = (_Float16)(zf[s][0][3] * sc);
        v[4] = (_Float16)(zf[s][1][0] * sc); v[5] = (_Float16)(zf[s][1][1] * sc);
        v[6] = (_Float16)(zf[s][1][2] * sc); v[7] = (_Float16)(zf[s][1][3] * sc);
        zh[s] = v;
    }

    // ---- stream the codebook: chunk c in stage c & 1; scores of chunk c are absorbed while chunk c+1 multiplies
    issue_chunk(img, 0, lds, wave, lane);
    float m1 = INFINITY, m2 = INFINITY, m3 = INFINITY, m4 = INFINITY;
    f32x16 accA[2], accB[2];
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        __syncthreads();                                          // chunk c landed everywhere; the other stage is free
        if (c + 1 < NCHUNK) issue_chunk(img, c + 1, lds + ((c + 1) & 1) * CHUNK_B, wave, lane);
        const char* st = lds + (c & 1) * CHUNK_B;
        f32x16 (&cur)[2] = (c & 1) ? accB : accA;
        f32x16 (&prev)[2] = (c & 1) ? accA : accB;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[jn][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 e0 = efrag(st, r, 2 * s + h);
            const f16x8 e1 = efrag(st, 32 + r, 2 * s + h);
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e0, zh[s], cur[0], 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(e1, zh[s], cur[1], 0, 0, 0);
            if (c > 0 && s == 3) absorb(prev[0], ee_s, c - 1, 0, h, inv, m1, m2, m3, m4);   // vector work in the MFMA shadow
            if (c > 0 && s == 9) absorb(prev[1], ee_s, c - 1, 1, h, inv, m1, m2, m3, m4);
        }
    }
    absorb(accB[0], ee_s, NCHUNK - 1, 0, h, inv, m1, m2, m3, m4);
    absorb(accB[1], ee_s, NCHUNK - 1, 1, h, inv, m1, m2, m3, m4);

    // ---- per row: merge the two lane halves' top-3 lists, threshold, decide or hand off
    const float p1 = __shfl_xor(m1, 32), p2 = __shfl_xor(m2, 32), p3 = __shfl_xor(m3, 32), p4 = __shfl_xor(m4, 32);
    const float best = fminf(m1, p1);
    const float eps = 0.00196076f * zn * emax + 0.00018311f * (zn + emax) * (zn + emax);
    const float thr = best + eps + (inv - inv);                  // NaN scale (exact path) poisons the threshold
    const int own = (m1 <= thr) + (m2 <= thr) + (m3 <= thr);
    const int oth = (p1 <= thr) + (p2 <= thr) + (p3 <= thr);
    const int c = own + oth;
    const bool complete = !(m4 <= thr) && !(p4 <= thr);          // no lane may be hiding a fourth score within eps
    const bool unique = c == 1;
    const bool amb = live && (h == 0) && !unique;
    if (live && h == 0) {
        if (unique) {
            idx[grow] = (m1 <= thr) ? decode_entry(m1, 0) : decode_entry(p1, 1);
        } else {
            idx[grow] = -1;
            const bool usable = complete && c >= 2;               // c == 0: NaN/Inf or out-of-range magnitudes
            cnt_out[grow] = usable ? (uint8_t)c : (uint8_t)255;
            unsigned long long lo = 0, hi = 0;                    // up to six 16-bit entry ids, no runtime-indexed arrays
            int n = 0;
            auto push = [&](bool take, int v) {
                if (take) {
                    if (n < 4) lo |= (unsigned long long)v << (16 * n);
                    else hi |= (unsigned long long)v << (16 * (n - 4));
                    ++n;
                }
            };
            push(m1 <= thr, decode_entry(m1, 0));
            push(m2 <= thr, decode_entry(m2, 0));
            push(m3 <= thr, decode_entry(m3, 0));
            push(p1 <= thr, decode_entry(p1, 1));
            push(p2 <= thr, decode_entry(p2, 1));
            push(p3 <= thr, decode_entry(p3, 1));
            uint4 v;
            v.x = (unsigned)lo; v.y = (unsigned)(lo >> 32); v.z = (unsigned)hi; v.w = (unsigned)(hi >> 32);
            *reinterpret_cast<uint4*>(cand_out + grow * MAXC) = v;
        }
    }
    // ambiguous rows of this workgroup -> its own segment of the work list (no global atomics)
    const unsigned long long mask = __ballot(amb);
    if (lane == 0) wcnt[wave] = __builtin_popcountll(mask);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    if (amb) amb_list[(long)blockIdx.x * WG_ROWS + base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (int)grow;
    if (tid == 0) amb_count[blockIdx.x] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
}

// ------------------------------------------------------------------------------------------------ refine
// Canonical chains threaded through G = 4 lanes: lane q of a group holds floats [64q, 64q+64) of its z row and of its
// candidate's codebook row (all loads issued up front: ONE memory latency), then the k-ordered fmaf chain runs as four
// 64-step rounds, round q continuing from the accumulator lane q-1 produced.  Bit-identical to a single 256-step chain.
__device__ __forceinline__ void chain_pair_x4(const float* __restrict__ zr, const float* __restrict__ er, int q, bool active,
                                              float& zz, float& dot) {
    f32x4 x[16], y[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {                              // idle lanes load nothing (a shared dummy row would hot-spot one L2 channel)
        x[u] = active ? *reinterpret_cast<const f32x4*>(zr + 64 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
        y[u] = active ? *reinterpret_cast<const f32x4*>(er + 64 * q + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float a = 0.f, b = 0.f;
    const int lane = threadIdx.x & 63, base = lane & ~3;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const float a_in = round ? __shfl(a, base + round - 1) : 0.f;
        const float b_in = round ? __shfl(b, base + round - 1) : 0.f;
        float ta = a_in, tb = b_in;
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ta = fmaf(x[u][c], x[u][c], ta);
                tb = fmaf(x[u][c], y[u][c], tb);
            }
        if (q == round) { a = ta; b = tb; }
    }
    zz = __shfl(a, base + 3);
    dot = __shfl(b, base + 3);
}

// full-row single-lane form (used by the all-K fallback)
__device__ __forceinline__ void chain_pair(const float* __restrict__ zr, const float* __restrict__ er, float& zz, float& dot) {
    float a = 0.f, b = 0.f;
    for (int j0 = 0; j0 < D; j0 += 32) {
        f32x4 x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = *reinterpret_cast<const f32x4*>(zr + j0 + 4 * u);
            y[u] = *reinterpret_cast<const f32x4*>(er + j0 + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a = fmaf(x[u][c], x[u][c], a);
                b = fmaf(x[u][c], y[u][c], b);
            }
    }
    zz = a;
    dot = b;
}

__global__ __launch_bounds__(256) void vq_refine_kernel(const float* __restrict__ z, const float* __restrict__ E,
                                                        const float* __restrict__ ee, long M, int64_t* __restrict__ idx,
                                                        const uint16_t* __restrict__ cand_out, const uint8_t* __restrict__ cnt_out,
                                                        const int* __restrict__ amb_count, const int* __restrict__ amb_list_all) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int seg = blockIdx.x >> 1, part = blockIdx.x & 1;     // 2 blocks share one filter workgroup's segment
    const int n_amb = amb_count[seg];
    const int* amb_list = amb_list_all + (long)seg * WG_ROWS;
    const int tid = threadIdx.x;
    const int g = tid >> 5;              // 8 rows per block pass, 32 lanes each: 8 candidate slots x 4 lanes
    const int j = (tid >> 2) & 7;        // candidate slot
    const int q = tid & 3;               // quarter of the row
    for (int i0 = part * 8; i0 < n_amb; i0 += 16) {
        const int i = i0 + g;
        float d = INFINITY;
        int k = 0x7fffffff;
        long grow = -1;
        bool work = false, overflow = false;
        if (i < n_amb) {
            grow = amb_list[i];
            const int c = cnt_out[grow];
            if (c == 255) overflow = true;
            else if (j < c) {
                k = cand_out[grow * MAXC + j];
                work = true;
            }
        }
        // every lane runs the (shuffling) chain code; lanes without work load nothing and are ignored
        float zz, dot;
        chain_pair_x4(z + (work ? grow : 0) * D, E + (long)(work ? k : 0) * D, q, work, zz, dot);
        if (work) {
            const float t = zz + ee[k];
            d = t - 2.0f * dot;
        } else {
            k = 0x7fffffff;
        }
#pragma unroll
        for (int o = 4; o < 32; o <<= 1) {                        // over the 8 candidate slots (lanes 4 apart)
            const float od = __shfl_xor(d, o);
            const int ok = __shfl_xor(k, o);
            if (dvq_argmin_better(od, ok, d, k)) { d = od; k = ok; }
        }
        if (grow >= 0 && !overflow && (tid & 31) == 0) idx[grow] = k;
        if (!__syncthreads_or(overflow)) continue;
        // rows without a usable candidate list (NaN/Inf, out-of-range magnitudes, possibly incomplete lists): all K entries,
        // the whole block per row
        for (int u = 0; u < 8; ++u) {
            const int iu = i0 + u;
            if (iu >= n_amb) break;
            const long gr = amb_list[iu];
            if (cnt_out[gr] != 255) continue;
            const float* zr = z + gr * D;
            float bv = INFINITY;
            int bi = 0x7fffffff;
            for (int kk = tid; kk < K; kk += 256) {
                float zz2, dot2;
                chain_pair(zr, E + (long)kk * D, zz2, dot2);
                const float t = zz2 + ee[kk];
                const float dd = t - 2.0f * dot2;
                if (dvq_argmin_better(dd, kk, bv, bi)) { bv = dd; bi = kk; }
            }
            s_v[tid] = bv;
            s_i[tid] = bi;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o && dvq_argmin_better(s_v[tid + o], s_i[tid + o], s_v[tid], s_i[tid])) {
                    s_v[tid] = s_v[tid + o];
                    s_i[tid] = s_i[tid + o];
                }
                __syncthreads();
            }
            if (tid == 0) idx[gr] = s_i[0];
            __syncthreads();
        }
    }
}

struct FastScratch {
    uint16_t* cand;
    uint8_t* cnt;
    int* amb_count;
    int* amb_list;
    long n_wg;
    size_t bytes;
};

FastScratch plan(int64_t M, void* ws) {
    FastScratch s;
    s.n_wg = (M + WG_ROWS - 1) / WG_ROWS;
    char* p = (char*)ws;
    auto take = [&](size_t n) { char* q = p; p += dvq_round_up(n, 256); return q; };
    s.cand = (uint16_t*)take((size_t)M * MAXC * 2);
    s.cnt = (uint8_t*)take((size_t)M);
    s.amb_count = (int*)take((size_t)s.n_wg * 4);
    s.amb_list = (int*)take((size_t)s.n_wg * WG_ROWS * 4);
    s.bytes = (size_t)(p - (char*)ws);
    return s;
}

}  // namespace

extern "C" int dvq_vq_fast_supported(int Kq, int Dq) { return Kq == K && Dq == D; }

extern "C" size_t dvq_vq_pack_bytes(int Kq, int Dq) { return dvq_vq_fast_supported(Kq, Dq) ? PK_BYTES : 0; }

extern "C" int dvq_vq_pack(const float* E, int Kq, int Dq, void* packed, size_t packed_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_pack: the fast path supports K=%d, D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(E && packed && dvq_aligned16(E) && dvq_aligned16(packed), "vq_pack: null/unaligned pointer");
    DVQ_REQUIRE(packed_bytes >= PK_BYTES, "vq_pack: buffer %zu < %zu bytes", packed_bytes, PK_BYTES);
    hipStream_t st = (hipStream_t)stream;
    char* pk = (char*)packed;
    hipLaunchKernelGGL(vq_pack_norm_kernel, dim3(1), dim3(K), 0, st, E, (float*)(pk + PK_OFF_EE), (PackHeader*)pk);
    DVQ_CHECK_LAUNCH("vq_pack_norm");
    hipLaunchKernelGGL(vq_pack_img_kernel, dim3((K * D + 255) / 256), dim3(256), 0, st, E, (const PackHeader*)pk,
                       (_Float16*)(pk + PK_OFF_IMG));
    DVQ_CHECK_LAUNCH("vq_pack_img");
    return DVQ_OK;
}

extern "C" size_t dvq_vq_fast_workspace_bytes(int64_t M, int Kq, int Dq) {
    if (M <= 0 || !dvq_vq_fast_supported(Kq, Dq)) return 256;
    return plan(M, nullptr).bytes;
}

extern "C" int dvq_vq_argmin_fast(const float* z, const float* E, const void* packed, int64_t M, int Kq, int Dq,
                                  int64_t* idx, void* workspace, size_t workspace_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(dvq_vq_fast_supported(Kq, Dq), "vq_argmin_fast: supports K=%d, D=%d only (got %d, %d)", K, D, Kq, Dq);
    DVQ_REQUIRE(M >= 0 && M < (1L << 31), "vq_argmin_fast: bad M");
    if (M == 0) return DVQ_OK;
    DVQ_REQUIRE(z && E && packed && idx && workspace, "vq_argmin_fast: null pointer");
    DVQ_REQUIRE(dvq_aligned16(z) && dvq_aligned16(E) && dvq_aligned16(packed) && dvq_aligned16(workspace),
                "vq_argmin_fast: pointers must be 16-byte aligned (z must be dense [M,256])");
    const FastScratch s = plan(M, workspace);
    if (workspace_bytes < s.bytes) {
        dvq_set_error("vq_argmin_fast: workspace %zu < %zu bytes", workspace_bytes, s.bytes);
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vq_filter_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) {
            dvq_set_error("vq_argmin_fast: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
        attr_set = true;
    }
    const char* pk = (const char*)packed;
    DVQ_PROF("vq_argmin_total", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 4 + (double)M * 8, st);
    {
        DVQ_PROF("vq_filter", 2.0 * M * K * D, (double)M * D * 4 + (double)K * D * 2 + (double)M * 8, st);
        hipLaunchKernelGGL(vq_filter_kernel, dim3((unsigned)s.n_wg), dim3(256), LDS_BYTES, st, z, (long)M, pk, idx, s.cand,
                           s.cnt, s.amb_count, s.amb_list);
    }
    DVQ_CHECK_LAUNCH("vq_filter");
    {
        DVQ_PROF("vq_refine", 0, 0, st);
        hipLaunchKernelGGL(vq_refine_kernel, dim3((unsigned)(s.n_wg * 2)), dim3(256), 0, st, z, E,
                           (const float*)(pk + PK_OFF_EE), (long)M, idx, s.cand, s.cnt, s.amb_count, s.amb_list);
    }
    DVQ_CHECK_LAUNCH("vq_refine");
    return DVQ_OK;
}
