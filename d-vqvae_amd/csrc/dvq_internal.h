// Internal declarations shared by the HIP translation units of libdvq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include "../../include/dvq.h"

// Diagnostics build (make EXTRA=-DDVQ_DIAG): timing-only ablation variants (DVQ_VQ_ABL, DVQ_PN_ABL, DVQ_GEMM_ABL: results
// INVALID), phase stamps (DVQ_VQ_DBG, DVQ_GEMM_CLK) and the DVQ_GEMM_NODMA switch exist only there.  The shipped library
// never reads those variables: a stray one in the environment cannot change a result.
#ifdef DVQ_DIAG
#define DVQ_DIAG_ON 1
#ifndef DVQ_GEMM_DIAG
#define DVQ_GEMM_DIAG
#endif
#else
#define DVQ_DIAG_ON 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that publishes LDS-DMA data (global_load_lds): the wave's own pieces must have landed BEFORE it arrives.
// __syncthreads() alone only guarantees lgkmcnt(0): the compiler does not treat the DMA's LDS write as a store to release
// (seen in the ISA of gemm_bf16x3_wide_kernel: s_barrier first, the vmcnt wait after it).
#ifdef __HIPCC__
__device__ __forceinline__ void dvq_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// Workgroup barrier with the wave's own LDS operations COMPLETE before it arrives, spelled out.  __syncthreads() is supposed to
// imply it, and the compiler normally emits "s_waitcnt lgkmcnt(0)" in front of s_barrier -- but not always: in pn_trunk3_kernel
// (round 5) the barrier at the head of the conv3 loop came out WITHOUT the wait although the back edge carries LDS stores (the next
// chunk's W3 rows, the ring pairs).  A wave then passes the barrier with its stores still in flight, its neighbours read the old
// contents -- nothing at one workgroup per CU, where the four waves run in step, a wrong tile record every few thousand tiles as soon
// as workgroups share a CU.  (This is the shape of the round-3 fault of pn_trunk_filter_kernel, whose barrier has the wait in the
// current build.)  EVERY workgroup barrier of pointnet.hip and pointnet_filter.hip goes through here (round 6: the exact stage's,
// the centre kernel's and the unfused trunk's too).
__device__ __forceinline__ void dvq_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
#endif

void dvq_set_error(const char* fmt, ...);

#define DVQ_REQUIRE(cond, ...)                      \
    do {                                            \
        if (!(cond)) {                              \
            dvq_set_error(__VA_ARGS__);             \
            return DVQ_EINVAL;                      \
        }                                           \
    } while (0)

// Function attributes (dynamic LDS limit) are per device: one process may drive several GPUs, from several threads.
// run(fn) calls fn() once per device; concurrent first calls are serialised, and a FAILED attempt is not remembered
// (the next launch on that device tries again and reports again).
struct DvqOncePerDevice {
    std::atomic<unsigned char> done[128];
    std::mutex lock;
    DvqOncePerDevice() {
        for (auto& d : done) d.store(0, std::memory_order_relaxed);
    }
    template <class F>
    hipError_t run(F&& fn) {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 127;
        if (done[d].load(std::memory_order_acquire)) return hipSuccess;
        std::lock_guard<std::mutex> g(lock);
        if (done[d].load(std::memory_order_relaxed)) return hipSuccess;
        const hipError_t e = fn();
        if (e == hipSuccess) done[d].store(1, std::memory_order_release);
        return e;
    }
};

// Launch with the thread's sticky HIP error cleared first: hipGetLastError() after the launch must report THIS launch,
// not an unrelated earlier failure of another library in the same thread (e.g. a device probe before the runtime was up).
#define DVQ_LAUNCH(...)                  \
    do {                                 \
        (void)hipGetLastError();         \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
#define DVQ_CHECK_LAUNCH(what)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            dvq_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e__)); \
            return DVQ_ELAUNCH;                                                       \
        }                                                                             \
    } while (0)

#define DVQ_PROPAGATE(expr)        \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != DVQ_OK) return rc__; \
    } while (0)

static inline bool dvq_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline size_t dvq_round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---------------------------------------------------------------- optional per-launch timing (HIP events)
// Off by default.  When enabled (dvq_prof_enable), every kernel launch of the library is bracketed by two
// events on the launch stream; dvq_prof_read() resolves them into per-kernel totals.
struct DvqProfScope {
    int slot;
    hipStream_t stream;
    DvqProfScope(const char* kind, double flops, double bytes, hipStream_t st);
    ~DvqProfScope();
};
extern bool g_dvq_prof_on;
#define DVQ_PROF(kind, flops, bytes, st) DvqProfScope prof_scope__(kind, flops, bytes, st)

// ---------------------------------------------------------------- fp32 MFMA GEMM with fused epilogues
enum GemmEpilogue {
    EPI_BIAS = 0,    // out = act(acc + bias)
    EPI_RESID = 1,   // out = acc + bias + resid
    EPI_GATE = 2,    // out = tanh(a) * sigmoid(b) over gate-packed channel pairs (+ optional pre-gate store)
    EPI_COLMAX = 3,  // per-tile column max over valid rows -> partial[tile_m][N]
    EPI_ARGMIN = 4,  // A = codebook, W = z rows: per z row, (min, argmin) of (zz+ee)-2*acc over the tile's entries
    EPI_STATE = 5    // fp16-plane kernels: the raw accumulator pair after the launch's k-steps -> out (first sums), pre (second sums)
};

struct GemmSrc {
    const float* A;
    const float* W;
    long lda, ldw;
    int K;
    int wp_kind;          // DVQ_PLANES_BF16X3: Wp = three bf16 planes; DVQ_PLANES_F16X2: two fp16 planes of w * 2^t[row] (+ GemmParams::wscale)
    const uint16_t* Wp;   // optional: W pre-split into planes [planes][N][ldw] (plane stride wp_plane elements)
    long wp_plane;
    // optional (fp16-plane kernels only): output row m reads activation row arow[m] instead of row m -- rows that are a function of
    // a class label only are computed once per class and read through the label (pixelcnn.hip)
    const int64_t* arow;
};

struct GemmParams {
    GemmSrc src[DVQ_MAX_SRC];
    int nsrc;
    int N;
    long M;
    const float* bias;   // [N] or null
    const float* wscale; // f16x2 planes: [N] row scales 2^-t[n] (one array for all sources of the launch)
    float* out;          // EPI_BIAS/RESID: [M,N]; EPI_GATE: [M,N/2]
    long ldo;
    int relu;
    // EPI_RESID
    const float* resid;
    long ldr;
    // EPI_GATE
    const float* cls;        // [n_classes][N] gate-packed, or null
    const int64_t* label;    // [M]
    float* pre;              // optional pre-gate output [M,N] (acc + bias, gate-packed order)
    long ldpre;
    // EPI_COLMAX
    int rows_per_group;      // padded rows per sample (multiple of 128)
    int valid_rows;          // real rows per sample
    float* partial;          // [M/128][N]
    // EPI_ARGMIN (A rows = codebook entries, W rows = z rows)
    const float* row_norm;   // ee[M]
    const float* col_norm;   // zz[N]
    float* part_val;         // [tiles_m][N]
    int* part_idx;           // [tiles_m][N]
    // fp16-plane kernels: the accumulator pair starts from a stored state (EPI_STATE of a launch over the FIRST sources of the full
    // product) instead of zero; output row m continues row acc_row[m] (or m).  Continuing a state gives the bits of the one-launch
    // product: every accumulator sees the same MFMA sequence.
    const float* acc_hi;
    const float* acc_lo;
    long ldacc;
    const int64_t* acc_row;
    int prof_cls;            // 1 = a launch on a class table's rows: "gemm_*_cls" in the per-launch breakdown, small-batch kernels up to 1 024 rows
    int dbg_abl;             // diagnostics only (env DVQ_GEMM_ABL): 2 = no MFMAs
    unsigned long long* dbg_clk;   // diagnostics only (env DVQ_GEMM_CLK=1): block 0 stores {memtime, memrealtime} x {begin, end}
};

// torch.argmin ordering: a NaN beats everything, among equals the lower index wins
__device__ __forceinline__ bool dvq_argmin_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn || bn) return vn && (!bn || i < bi);
    return v < bv || (v == bv && i < bi);
}

// compute units of the current device (cached per device; 256 if the runtime does not say)
inline int dvq_num_cus() {
    static int cus[128] = {0};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 127;
    int v = __atomic_load_n(&cus[d], __ATOMIC_RELAXED);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        __atomic_store_n(&cus[d], v, __ATOMIC_RELAXED);
    }
    return v;
}

int dvq_launch_gemm(const GemmParams& p, GemmEpilogue epi, hipStream_t stream);
int dvq_launch_gemm_bf16x3(const GemmParams& p, GemmEpilogue epi, hipStream_t stream);
int dvq_launch_gemm_f16x2(const GemmParams& p, GemmEpilogue epi, hipStream_t stream);
// 1 .. DVQ_GEMM_GROUP_MAX gated GEMMs that do not depend on each other: one launch where the launch COUNT is what costs (fp16
// planes, same M <= 256 and N), a launch each otherwise.  Results are those of dvq_launch_gemm(ps[i], EPI_GATE) bit for bit.
constexpr int DVQ_GEMM_GROUP_MAX = 3;
int dvq_launch_gemm_gate_group(const GemmParams* ps, int n, hipStream_t stream);
int dvq_launch_gemm_f16x2_gate_group(const GemmParams* ps, int n, hipStream_t stream);   // < 0: not applicable, nothing launched
// 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-bf16 (default); env DVQ_GEMM=fp32|bf16x3
int dvq_gemm_mode();

// Behaviour knobs read from the environment ONCE (first use) -- DVQ_GEMM_WIDE, DVQ_GEMM_DEPHASE, DVQ_GEMM_SKINNY, DVQ_PN_FILTER, DVQ_PN_EXHAUSTIVE,
// DVQ_PN_CAPS, DVQ_PN_CHUNK, DVQ_PN_STATS, DVQ_PIXELCNN_CHUNK -- none of them changes a result (tile shapes, chunk sizes, the
// exhaustive PointNet evaluation the filter is tested against).  dvq_reload_env() re-reads them (tests flip them in-process).
struct DvqKnobs {
    int gemm_wide;        // 0: 128 x 128 kernels only
    int gemm_dephase;
    int gemm_tn;          // 0 (default): the tiled f16x2 kernel picks 128 x 256 or 128 x 128 tiles per launch; 128 / 256 force one (DVQ_GEMM_TN)
    int vq_kernel;        // 16 (default): vq_stream16.hip; 8: vq_stream.hip's eight-wave kernel; 32: vq_rows.hip (DVQ_VQ_KERNEL)
    int vq_rows_delay;    // vq_rows.hip: start delay of the second half of the grid, 10 ns ticks (DVQ_VQ_ROWS_DELAY)
    int gemm_skinny_prefetch;   // 0: no helper workgroups (DVQ_GEMM_SKINNY_PREFETCH=0)
    int gemm_skinny;      // 0: tiled kernels also for M <= 256 (DVQ_GEMM_SKINNY=0; the two must agree bitwise)
    int gemm_skinny_cols; // f16x2 skinny kernel: output columns per wave, 16 / 8 / 4 (DVQ_GEMM_SKINNY_COLS; 0 = by the launch's size; same bits)
    int pn_filter;        // 0 six-product trunk, 1 default, 2 filtered trunk whatever the tile fill
    int pn_tail;          // 1 (default): a cloud's 1 .. 32 points beyond a multiple of 256 as a one-block tail tile (DVQ_PN_TAIL=0: a full tile)
    int pn_exhaustive;    // 1: exact stage evaluates every point (what the filter must reproduce bit for bit)
    int pn_caps[2];       // candidate-list capacities (tests shrink them to reach the overflow paths); <= 0: default
    long pn_chunk;        // samples per PointNet launch (<= 0: at most 4 096, at least four launches per pass; DVQ_PN_CHUNK)
    int pn_trunk3;        // diagnostics build only: DVQ_PN_TRUNK3=1: full tiles on pn_trunk3_kernel (three workgroups per CU; measured 4.5 % slower); default 0: pn_trunk_filter_kernel (two)
    int pn_streams;       // 1 (default): the exact stage / STN FCs of a launch on a second stream beside the next launch's trunk kernel (DVQ_PN_STREAMS=0: one stream)
    int pn_slots;         // scratch sets the launches rotate through (<= 0: 2; DVQ_PN_SLOTS)
    int pn_stats;
    long pixelcnn_chunk;  // <= 0: default
    int pixelcnn_tables;  // 1 (default): what depends on the class label only is evaluated once per class (DVQ_PIXELCNN_TABLES=0: per row)
};
const DvqKnobs& dvq_knobs();
int dvq_launch_vq_stream16(const float* z, const float* E, const void* packed, long M, int64_t* idx, unsigned long long* slow_rows,
                           unsigned long long* dbg, hipStream_t st);
int dvq_launch_vq_pipe(const float* z, const float* E, const void* packed, long M, int64_t* idx, unsigned long long* slow_rows,
                       unsigned long long* dbg, hipStream_t st);
int dvq_launch_vq_rows(const float* z, const float* E, const void* packed, long M, int64_t* idx, unsigned long long* slow_rows,
                       hipStream_t st);
// simple helpers implemented in misc.hip
int dvq_launch_gather_rows(const float* table, const int64_t* idx, long idx_stride, long M, int K, int D,
                           float* out, long ldo, int32_t* err_flag, hipStream_t stream);
int dvq_launch_colmax_reduce(const float* partial, long groups, int tiles_per_group, int N, int relu,
                             float* out, long ldo, hipStream_t stream);
