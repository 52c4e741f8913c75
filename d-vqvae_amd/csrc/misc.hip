// Error plumbing and the small data-movement kernels of the path (gathers, concatenations, reductions).
#include "dvq_internal.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dvq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dvq_last_error(void) { return g_err; }
extern "C" int dvq_abi_version(void) { return DVQ_ABI_VERSION; }

// ---------------------------------------------------------------- environment knobs, read once (dvq_internal.h)
namespace {
std::mutex g_knob_lock;
std::atomic<const DvqKnobs*> g_knobs{nullptr};
DvqKnobs* read_knobs() {
    DvqKnobs* k = new DvqKnobs();                       // a superseded set is leaked on purpose: launches may still read it
    auto is = [](const char* name, char c) { const char* e = getenv(name); return e && e[0] == c; };
    auto num = [](const char* name) { const char* e = getenv(name); return e ? atol(e) : 0L; };
    k->gemm_wide = !is("DVQ_GEMM_WIDE", '0');
    k->gemm_tn = (int)num("DVQ_GEMM_TN");
    k->gemm_dephase = is("DVQ_GEMM_DEPHASE", '0') ? 0 : (is("DVQ_GEMM_DEPHASE", '1') ? 1 : 2);    // f16x2 tiled kernel: 2 = ping-pong (default)
    k->gemm_skinny = is("DVQ_GEMM_SKINNY", '0') ? 0 : (is("DVQ_GEMM_SKINNY", '2') ? 2 : 1);   // 2: the register-staged variant
    {
        const int v = getenv("DVQ_VQ_KERNEL") ? atoi(getenv("DVQ_VQ_KERNEL")) : 0;
        k->vq_kernel = v == 8 ? 8 : (v == 32 ? 32 : (v == 17 ? 17 : 16));       // default: the sixteen-wave kernel; 8: eight waves (generated tile body); 32: rows resident, codebook streamed
    }
    k->gemm_skinny_prefetch = !is("DVQ_GEMM_SKINNY_PREFETCH", '0');
    k->gemm_skinny_cols = (int)num("DVQ_GEMM_SKINNY_COLS");
    k->vq_rows_delay = getenv("DVQ_VQ_ROWS_DELAY") ? (int)num("DVQ_VQ_ROWS_DELAY") : 0;
    k->pn_filter = is("DVQ_PN_FILTER", '0') ? 0 : (is("DVQ_PN_FILTER", '2') ? 2 : 1);
    k->pn_tail = is("DVQ_PN_TAIL", '0') ? 0 : 1;
    k->pn_exhaustive = is("DVQ_PN_EXHAUSTIVE", '1');
    k->pn_caps[0] = k->pn_caps[1] = -1;
    if (const char* e = getenv("DVQ_PN_CAPS")) {
        int a = 0, c = 0;
        if (sscanf(e, "%d,%d", &a, &c) == 2) { k->pn_caps[0] = a < 0 ? 0 : a; k->pn_caps[1] = c < 0 ? 0 : c; }
    }
    k->pn_chunk = num("DVQ_PN_CHUNK");
    k->pn_streams = is("DVQ_PN_STREAMS", '0') ? 0 : 1;
#ifdef DVQ_DIAG
    k->pn_trunk3 = is("DVQ_PN_TRUNK3", '1') ? 1 : 0;            // the kernel exists in the diagnostics build only
#else
    k->pn_trunk3 = 0;
#endif
    k->pn_slots = (int)num("DVQ_PN_SLOTS");
    k->pn_stats = getenv("DVQ_PN_STATS") != nullptr;
    k->pixelcnn_chunk = num("DVQ_PIXELCNN_CHUNK");
    k->pixelcnn_tables = is("DVQ_PIXELCNN_TABLES", '0') ? 0 : 1;
    return k;
}
}  // namespace
const DvqKnobs& dvq_knobs() {
    const DvqKnobs* k = g_knobs.load(std::memory_order_acquire);
    if (k) return *k;
    std::lock_guard<std::mutex> g(g_knob_lock);
    k = g_knobs.load(std::memory_order_relaxed);
    if (!k) { k = read_knobs(); g_knobs.store(k, std::memory_order_release); }
    return *k;
}
extern "C" int dvq_reload_env(void) {
    std::lock_guard<std::mutex> g(g_knob_lock);
    g_knobs.store(read_knobs(), std::memory_order_release);
    return DVQ_OK;
}
extern "C" int dvq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

// ------------------------------------------------------------------ per-launch timing
#include <string>
#include <vector>
#include <map>
bool g_dvq_prof_on = false;
namespace {
struct ProfRec { int kind; hipEvent_t a, b; double flops, bytes; };
std::vector<ProfRec> g_recs;
std::vector<std::string> g_kinds;
std::vector<hipEvent_t> g_pool;
std::mutex g_prof_lock;                 // launches may come from several threads (one stream each)
int kind_id(const char* name) {
    for (size_t i = 0; i < g_kinds.size(); ++i) if (g_kinds[i] == name) return (int)i;
    g_kinds.emplace_back(name);
    return (int)g_kinds.size() - 1;
}
hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

DvqProfScope::DvqProfScope(const char* kind, double flops, double bytes, hipStream_t st) : slot(-1), stream(st) {
    if (!g_dvq_prof_on) return;
    std::lock_guard<std::mutex> g(g_prof_lock);
    ProfRec r{kind_id(kind), get_event(), get_event(), flops, bytes};
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    slot = (int)g_recs.size() - 1;
}
DvqProfScope::~DvqProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> g(g_prof_lock);
    if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].b, stream);
}

extern "C" int dvq_prof_enable(int on) { g_dvq_prof_on = on != 0; return DVQ_OK; }
extern "C" int dvq_prof_reset(void) {
    std::lock_guard<std::mutex> g(g_prof_lock);
    for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    return DVQ_OK;
}
extern "C" int dvq_prof_read(dvq_prof_entry* out, int max_entries) {
    std::lock_guard<std::mutex> g(g_prof_lock);
    std::vector<dvq_prof_entry> acc(g_kinds.size());
    for (size_t i = 0; i < g_kinds.size(); ++i) {
        memset(&acc[i], 0, sizeof(dvq_prof_entry));
        snprintf(acc[i].name, sizeof(acc[i].name), "%s", g_kinds[i].c_str());
    }
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
        acc[r.kind].count += 1;
        acc[r.kind].ms += ms;
        acc[r.kind].flops += r.flops;
        acc[r.kind].bytes += r.bytes;
    }
    int n = 0;
    for (auto& e : acc) {
        if (e.count == 0) continue;
        if (n < max_entries) out[n] = e;
        ++n;
    }
    return n;
}

namespace {

// out[m, 0:D] = table[idx[m*idx_stride], 0:D]; out-of-range index -> zeros + error flag
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx, long idx_stride,
                                   long M, int K, int D, float* __restrict__ out, long ldo, int32_t* err_flag) {
    const int lanes_per_row = D / 4;     // D % 4 == 0
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long m = gid / lanes_per_row;
    const int c4 = (int)(gid % lanes_per_row);
    if (m >= M) return;
    const int64_t k = idx[m * idx_stride];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (k >= 0 && k < K) {
        v = *reinterpret_cast<const f32x4*>(table + k * D + c4 * 4);
    } else if (c4 == 0 && err_flag) {
        atomicOr(err_flag, 1);
    }
    *reinterpret_cast<f32x4*>(out + m * ldo + c4 * 4) = v;
}

__global__ void colmax_reduce_kernel(const float* __restrict__ partial, long groups, int tiles, int N, int relu,
                                     float* __restrict__ out, long ldo) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long g = gid / N;
    const int n = (int)(gid % N);
    if (g >= groups) return;
    float mx = -INFINITY;
    for (int t = 0; t < tiles; ++t) mx = fmaxf(mx, partial[(g * tiles + t) * N + n]);
    if (relu) mx = fmaxf(mx, 0.f);
    out[g * ldo + n] = mx;
}

__global__ void copy_cols_kernel(const float* __restrict__ src, long lds, long M, int W, float* __restrict__ out, long ldo) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int w4 = W / 4;
    const long m = gid / w4;
    const int c4 = (int)(gid % w4);
    if (m >= M) return;
    *reinterpret_cast<f32x4*>(out + m * ldo + c4 * 4) = *reinterpret_cast<const f32x4*>(src + m * lds + c4 * 4);
}

__global__ void assemble61_kernel(const float* __restrict__ recon, const float* __restrict__ pos, long B, float* __restrict__ out) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long b = gid / 61;
    const int c = (int)(gid % 61);
    if (b >= B) return;
    float v;
    if (c < 10) v = recon[b * 55 + c];
    else if (c < 13) v = pos[b * 6 + (c - 10)];
    else if (c < 58) v = recon[b * 55 + (c - 3)];
    else v = pos[b * 6 + 3 + (c - 58)];
    out[b * 61 + c] = v;
}

__global__ void transform_cloud_kernel(const float* __restrict__ pc, long pc_bstride, const float* __restrict__ R,
                                       const float* __restrict__ t, long B, int C, int N, float* __restrict__ out) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long b = gid / N;
    const int n = (int)(gid % N);
    if (b >= B) return;
    const float* src = pc + b * pc_bstride;
    const float x = src[n], y = src[N + n], z = src[2 * N + n];
    const float* r = R + b * 9;
    float* dst = out + b * (long)C * N;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = r[i * 3 + 0] * x;
        v = fmaf(r[i * 3 + 1], y, v);
        v = fmaf(r[i * 3 + 2], z, v);
        dst[i * N + n] = v + (t ? t[i] : 0.f);
    }
    for (int c = 3; c < C; ++c) dst[c * N + n] = src[c * N + n];
}

// Philox4x32-10 (Salmon et al., SC'11), counter = (column quad, row low, row high, stream), key = seed
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// out[r, 4q..4q+3] = -log(u), u = (24 random bits + 0.5) 2^-24 in (0, 1): Exp(1) variates that depend only on
// (seed, stream, GLOBAL row row0 + r, column), not on how the rows are split into calls, batches or ranks
// `perm` (optional): output row r holds the draws of global row row0 + perm[r] (the prior is evaluated in label order)
__global__ void exp1_noise_kernel(unsigned long long seed, unsigned stream_id, long row0, long rows, int cols, float* __restrict__ out,
                                  const int64_t* __restrict__ perm) {
    const int quads = cols / 4;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = gid / quads;
    const int q = (int)(gid % quads);
    if (r >= rows) return;
    const unsigned long long grow = (unsigned long long)(row0 + (perm ? (long)perm[r] : r));
    unsigned c[4] = {(unsigned)q, (unsigned)grow, (unsigned)(grow >> 32), stream_id};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = -logf(((float)(c[i] >> 8) + 0.5f) * 5.9604644775390625e-08f);
    *reinterpret_cast<f32x4*>(out + r * cols + 4 * q) = v;
}

// does the matrix core keep fp16 subnormal inputs?  out[0] = sum over k of a_k b_k with a_0 = a_8 = 16 * 2^-24 (subnormal),
// b = 1024: 2 * 2^-10 when kept, 0 when flushed; out[1] = the same product by scalar fp32 arithmetic
typedef _Float16 probe_f16x8 __attribute__((ext_vector_type(8)));
__global__ void probe_f16_subnormal_kernel(float* out) {
    probe_f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    const unsigned short bits = 0x0010;
    a[0] = __builtin_bit_cast(_Float16, bits);
    b[0] = (_Float16)1024.f;
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        out[1] = 2.0f * (16.0f * 5.9604644775390625e-08f) * 1024.0f;
    }
}

}  // namespace

int dvq_launch_gather_rows(const float* table, const int64_t* idx, long idx_stride, long M, int K, int D, float* out,
                           long ldo, int32_t* err_flag, hipStream_t stream) {
    DVQ_REQUIRE(table && idx && out, "gather: null pointer");
    DVQ_REQUIRE(D % 4 == 0 && ldo % 4 == 0 && dvq_aligned16(table) && dvq_aligned16(out), "gather: rows not 16-byte aligned");
    if (M == 0) return DVQ_OK;
    const long total = M * (D / 4);
    {
        DVQ_PROF("gather_rows", 0, 2.0 * M * D * 4, stream);
        DVQ_LAUNCH(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, table, idx,
                           idx_stride, M, K, D, out, ldo, err_flag);
    }
    DVQ_CHECK_LAUNCH("gather_rows");
    return DVQ_OK;
}

int dvq_launch_colmax_reduce(const float* partial, long groups, int tiles_per_group, int N, int relu, float* out,
                             long ldo, hipStream_t stream) {
    const long total = groups * N;
    {
        DVQ_PROF("colmax_reduce", 0, (double)groups * (tiles_per_group + 1) * N * 4, stream);
        DVQ_LAUNCH(colmax_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, partial, groups,
                           tiles_per_group, N, relu, out, ldo);
    }
    DVQ_CHECK_LAUNCH("colmax_reduce");
    return DVQ_OK;
}

extern "C" int dvq_vq_lookup(const float* E, const int64_t* idx, int64_t idx_stride, int64_t M, int K, int D, float* out,
                             int64_t ldo, int32_t* err_flag, dvq_stream_t stream) {
    DVQ_REQUIRE(K > 0 && D > 0 && M >= 0 && idx_stride >= 1, "vq_lookup: bad shape M=%ld K=%d D=%d", (long)M, K, D);
    return dvq_launch_gather_rows(E, idx, idx_stride, M, K, D, out, ldo, err_flag, (hipStream_t)stream);
}

extern "C" int dvq_copy_cols(const float* src, int64_t lds, int64_t M, int W, float* out, int64_t ldo, dvq_stream_t stream) {
    DVQ_REQUIRE(src && out, "copy_cols: null pointer");
    DVQ_REQUIRE(W % 4 == 0 && lds % 4 == 0 && ldo % 4 == 0 && dvq_aligned16(src) && dvq_aligned16(out),
                "copy_cols: rows not 16-byte aligned");
    if (M == 0) return DVQ_OK;
    const long total = M * (W / 4);
    DVQ_LAUNCH(copy_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                       (long)lds, (long)M, W, out, (long)ldo);
    DVQ_CHECK_LAUNCH("copy_cols");
    return DVQ_OK;
}

extern "C" int dvq_assemble61(const float* recon, const float* recon_pos, int64_t B, float* out, dvq_stream_t stream) {
    DVQ_REQUIRE(recon && recon_pos && out, "assemble61: null pointer");
    if (B == 0) return DVQ_OK;
    const long total = B * 61;
    DVQ_LAUNCH(assemble61_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, recon,
                       recon_pos, (long)B, out);
    DVQ_CHECK_LAUNCH("assemble61");
    return DVQ_OK;
}

extern "C" int dvq_transform_cloud(const float* pc, int64_t pc_batch_stride, const float* R, const float* t, int64_t B,
                                   int C, int N, float* out, dvq_stream_t stream) {
    DVQ_REQUIRE(pc && R && out, "transform_cloud: null pointer");
    DVQ_REQUIRE(C >= 3 && N > 0, "transform_cloud: bad shape C=%d N=%d", C, N);
    if (B == 0) return DVQ_OK;
    const long total = B * N;
    DVQ_LAUNCH(transform_cloud_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       pc, (long)pc_batch_stride, R, t, (long)B, C, N, out);
    DVQ_CHECK_LAUNCH("transform_cloud");
    return DVQ_OK;
}

extern "C" int dvq_exp1_noise(uint64_t seed, uint32_t stream_id, int64_t row0, int64_t rows, int cols, float* out, dvq_stream_t stream) {
    return dvq_exp1_noise_rows(seed, stream_id, row0, nullptr, rows, cols, out, stream);
}

extern "C" int dvq_exp1_noise_rows(uint64_t seed, uint32_t stream_id, int64_t row0, const int64_t* perm, int64_t rows, int cols, float* out,
                                   dvq_stream_t stream) {
    DVQ_REQUIRE(rows >= 0 && cols > 0 && cols % 4 == 0 && row0 >= 0, "exp1_noise: bad shape rows=%ld cols=%d row0=%ld", (long)rows, cols, (long)row0);
    if (rows == 0) return DVQ_OK;
    DVQ_REQUIRE(out && dvq_aligned16(out), "exp1_noise: null/unaligned output");
    const long total = rows * (cols / 4);
    DVQ_REQUIRE(total < (1L << 31) * 256, "exp1_noise: too many elements");
    {
        DVQ_PROF("exp1_noise", 0, (double)rows * cols * 4, (hipStream_t)stream);
        DVQ_LAUNCH(exp1_noise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                   (unsigned long long)seed, (unsigned)stream_id, (long)row0, (long)rows, cols, out, perm);
    }
    DVQ_CHECK_LAUNCH("exp1_noise");
    return DVQ_OK;
}

extern "C" int dvq_probe_f16_subnormal(float* out, dvq_stream_t stream) {
    DVQ_REQUIRE(out, "probe_f16_subnormal: null pointer");
    DVQ_LAUNCH(probe_f16_subnormal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    DVQ_CHECK_LAUNCH("probe_f16_subnormal");
    return DVQ_OK;
}
