// PointNet encoder (PointNetEncoder.forward, network/pointnet_encoder.py:140-169; STN3d.forward :27-45).
// Default path: ONE fused trunk kernel per encoder pass (pn_trunk_kernel, below): conv1 (C -> 64, vector ALU, with the 3x3
// input transform xyz @ trans) -> conv2 (64 -> 128) -> conv3 (128 -> 1024) -> max over the points, on the split-bf16
// (bf16x3) matrix-core arithmetic; no per-point activation ever leaves the CU, the kernel writes one row of column maxima
// per 128 points and colmax_reduce_kernel finishes the max.  The STN's FC layers (1024 -> 512 -> 256 -> 9) use the GEMM.
// With DVQ_GEMM=fp32 (exercised by tests/test_gpu_parity.py::test_fp32_gemm_branch_matches_goldens) the trunk runs
// unfused instead: pn_layer1_kernel (vector ALU) + two fp32-MFMA GEMMs, the second with the column-max epilogue.
// BatchNorm (eval) is folded into the weights by the packer (packing.py, fp64 fold, rounded once).
#include "dvq_internal.h"
#include <map>
#include <utility>
#include <vector>

int dvq_launch_pn_trunk(const float* pc, int C, int N, long B, const float* trans, const float* W1, const float* b1,
                        const uint16_t* W2p, const float* b2, const uint16_t* W3p, const float* b3, float* partial,
                        hipStream_t st);
// pointnet_filter.hip
size_t dvq_pn_filter_image_bytes();
int dvq_launch_pn_filter_pack(const float* w2, const float* w3, void* image, hipStream_t st);
int dvq_launch_pn_trunk_filter(const float* pc, int C, int N, int Npad, long B, const float* trans, const float* W1, const float* b1,
                               const float* W2, const uint16_t* W2p, const float* b2, const void* w3f, const float* w3, const float* b3,
                               int relu, float* h2buf, void* part, unsigned* tstat, float* cbuf, float* feat, long ld_feat,
                               unsigned long long* stats, hipStream_t st);
int dvq_launch_pn_filter_front(const float* pc, int C, int N, int Npad, long B, const float* trans, const float* W1, const float* b1,
                               const float* W2, const uint16_t* W2p, const float* b2, const void* w3f, float* h2buf, void* part,
                               unsigned* tstat, float* cbuf, unsigned long long* stats, hipStream_t st);
int dvq_launch_pn_filter_back(int N, int Npad, long B, const void* w3f, const float* w3, const float* b3, int relu, const float* h2buf,
                              const void* part, const unsigned* tstat, const float* cbuf, float* feat, long ld_feat, unsigned long long* stats,
                              hipStream_t st);
int dvq_pn_fault_counters(unsigned long long* out2, int reset);

namespace {

// 16 threads per point, 4 output channels each; rows >= N of a sample's padded block are zero-filled.
__global__ void pn_layer1_kernel(const float* __restrict__ pc, int C, int N, int Npad, long B,
                                 const float* __restrict__ trans /* [B,9] or null */, const float* __restrict__ W /* [64,4] */,
                                 const float* __restrict__ bias, float* __restrict__ out /* [B*Npad,64] */) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long pt = gid >> 4;
    const int q = (int)(gid & 15);
    if (pt >= B * Npad) return;
    const long b = pt / Npad;
    const int n = (int)(pt % Npad);
    f32x4 y = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const float* src = pc + b * (long)C * N;
        float x0 = src[n], x1 = src[N + n], x2 = src[2 * N + n];
        const float x3 = (C > 3) ? src[3 * N + n] : 0.f;
        if (trans) {   // xyz_new = xyz @ trans[b]  (row vector times matrix, pointnet_encoder.py:146)
            const float* t = trans + b * 9;
            const float n0 = fmaf(x2, t[6], fmaf(x1, t[3], x0 * t[0]));
            const float n1 = fmaf(x2, t[7], fmaf(x1, t[4], x0 * t[1]));
            const float n2 = fmaf(x2, t[8], fmaf(x1, t[5], x0 * t[2]));
            x0 = n0; x1 = n1; x2 = n2;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float* w = W + (q * 4 + o) * 4;
            float v = x0 * w[0];
            v = fmaf(x1, w[1], v);
            v = fmaf(x2, w[2], v);
            v = fmaf(x3, w[3], v);
            y[o] = fmaxf(v + bias[q * 4 + o], 0.f);
        }
    }
    *reinterpret_cast<f32x4*>(out + pt * 64 + q * 4) = y;
}

// Scratch of one encode call.  The filtered trunk works in LAUNCHES of ``chunk`` samples; a launch's trunk kernel ("front": centres,
// conv1/conv2/conv3 filter, h2 rows + tile records) and its exact stage ("back": pn_exact_kernel, which reads them) use one of
// ``slots`` scratch sets, so that the back of launch i can run on a second stream beside the front of launch i + 1 (below).
struct PnSlot {
    float *h2, *part, *cbuf;
    unsigned* tstat;
};
struct PnScratch {
    PnSlot slot[4];
    int slots;
    float *h1, *f0, *f1, *f2, *tr;
    unsigned long long* stats;
    long chunk;
    int Npad;
    size_t bytes;
};

PnScratch plan(int64_t B, int N, void* ws) {
    PnScratch s;
    s.Npad = (N + 255) / 256 * 256;                       // tiles of 128 (fused / unfused trunk) and of 256 (filtered trunk)
    const size_t per_sample = (size_t)s.Npad * 128 * 4 + (size_t)s.Npad * 96 + (size_t)(s.Npad / 256) * 16 + 512;
    const size_t budget = (size_t)9 << 29;                // 4.5 GB per scratch set: 7 282 samples of 1 024 points
    long chunk = (long)(budget / per_sample);
    {                                                     // samples per launch (DVQ_PN_CHUNK)
        const long v = dvq_knobs().pn_chunk;
        if (v > 0 && v < chunk) chunk = v;
        // default: launches of at most 4 096 samples and at least four launches per pass once there are 2 048 samples, so that the two
        // streams have something to overlap (the trunk kernel also likes a small, re-used h2 buffer: DESIGN.md 3.3)
        if (v <= 0) {
            if (chunk > 4096) chunk = 4096;
            if (B >= 2048 && chunk > (B + 3) / 4) chunk = (B + 3) / 4;
        }
    }
    if (chunk > B) chunk = B;
    if (chunk < 1) chunk = 1;
    // equal launches: 8 192 samples run as 4 x 2 048, not 3 x 2 731 + 1
    if (B > 0) chunk = (B + ((B + chunk - 1) / chunk) - 1) / ((B + chunk - 1) / chunk);
    s.chunk = chunk;
    const long launches = B > 0 ? (B + chunk - 1) / chunk : 1;
    s.slots = launches >= 2 ? (dvq_knobs().pn_slots > 0 ? dvq_knobs().pn_slots : 2) : 1;   // 2, 3 and 4 sets measured equal (17.9 ms per 16 384-cloud encode)
    if (s.slots > 4) s.slots = 4;
    if (s.slots > 2 * launches) s.slots = (int)(2 * launches);
    char* p = (char*)ws;
    auto take = [&](size_t n) { char* q = p; p += dvq_round_up(n, 256); return (float*)q; };
    for (int i = 0; i < 4; ++i) s.slot[i] = PnSlot{nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < s.slots; ++i) {
        s.slot[i].h2 = take((size_t)chunk * s.Npad * 128 * 4);
        s.slot[i].part = take((size_t)chunk * s.Npad * 96);   // [tiles128][1024] floats, or [tiles256][1024] float4 + [tiles256][1024] float2
        s.slot[i].tstat = (unsigned*)take((size_t)chunk * (s.Npad / 256) * 16);
        s.slot[i].cbuf = take((size_t)chunk * 128 * 4);
    }
    // conv1 rows of the UNFUSED trunk: DVQ_GEMM=fp32, or a weights struct without bf16 planes (include/dvq.h: the planes are optional).
    // That trunk runs on one stream and one scratch set, so with two sets it borrows the second set's conv2 rows (twice its size);
    // with one set (a single launch) it gets rows of its own.
    s.h1 = s.slots >= 2 ? s.slot[1].h2 : take((size_t)chunk * s.Npad * 64 * 4);
    s.stats = (unsigned long long*)take(64);
    s.f0 = take((size_t)chunk * 1024 * 4);
    s.f1 = take((size_t)chunk * 512 * 4);
    s.f2 = take((size_t)chunk * 256 * 4);
    s.tr = take((size_t)(B > 0 ? B : 1) * 16 * 4);        // the whole batch's transforms: pass 1 writes them, pass 2 reads them
    s.bytes = (size_t)(p - (char*)ws);
    return s;
}

int dense(const float* x, long ldx, int K, const float* w, const uint16_t* wp, const float* b, long M, int N, int relu,
          float* y, long ldy, hipStream_t st) {
    GemmParams p = {};
    p.src[0] = GemmSrc{x, w, ldx, (long)K, K, 0, wp, (long)N * K};
    p.nsrc = 1;
    p.M = M;
    p.N = N;
    p.bias = b;
    p.out = y;
    p.ldo = ldy;
    p.relu = relu;
    return dvq_launch_gemm(p, EPI_BIAS, st);
}

// DVQ_PN_FILTER: 0 = six-product trunk everywhere, 2 = filtered trunk whatever the fill of its tiles (tests), default 1
int filter_mode() { return dvq_knobs().pn_filter; }

// does the filtered trunk (pointnet_filter.hip) take this shape?  When its 256-point tiles are at least 3/4 full: padding slots repeat
// real points, and many repeats mean many ties for the exact stage (N = 300 in two tiles: 4.6 ms against 3.6 ms for the six-product trunk)
bool use_filter(const dvq_pointnet_weights* w, int N) {
    const int fm = filter_mode();
    const int tiles256 = (N + 255) / 256, over = N - 256 * (tiles256 - 1);
    // slots the filtered trunk evaluates: whole tiles, or whole tiles + a 32-point tail block (pointnet_filter.hip)
    const long slots = (tiles256 >= 2 && over <= 32 && dvq_knobs().pn_tail) ? 256L * (tiles256 - 1) + 32 : 256L * tiles256;
    return w->w2p && w->w3f && w->s_w2p && w->s_w3f && dvq_gemm_mode() == 1 && fm && N <= 16384 && (fm == 2 || 4L * N >= 3L * slots);
}

int trunk(const float* pc, int C, int N, long Bc, const float* trans, const float* w1, const float* b1, const float* w2,
          const uint16_t* w2p, const float* b2, const float* w3, const uint16_t* w3p, const void* w3f, const float* b3, int relu3,
          bool filtered, const PnScratch& s, float* feat, long ld_feat, hipStream_t st) {
    const PnSlot& sl = s.slot[0];
    if (filtered)
        return dvq_launch_pn_trunk_filter(pc, C, N, s.Npad, Bc, trans, w1, b1, w2, w2p, b2, w3f, w3, b3, relu3, sl.h2, sl.part, sl.tstat,
                                          sl.cbuf, feat, ld_feat, dvq_knobs().pn_stats ? s.stats : nullptr, st);
    if (w2p && w3p && dvq_gemm_mode() == 1) {       // fused trunk; w3p is the k-permuted plane image (see pn_trunk_kernel)
        DVQ_PROPAGATE(dvq_launch_pn_trunk(pc, C, N, Bc, trans, w1, b1, w2p, b2, w3p, b3, sl.part, st));
        return dvq_launch_colmax_reduce(sl.part, Bc, (N + 127) / 128, 1024, relu3, feat, ld_feat, st);   // the kernel's own tiling
    }
    w3p = nullptr;                                  // the unfused GEMM path takes natural-order planes only: split on the fly
    DVQ_REQUIRE(s.h1, "pointnet: no scratch for the unfused trunk");
    const long rows = Bc * s.Npad;
    const long threads = rows * 16;
    {
        DVQ_PROF("pn_layer1", 2.0 * rows * 64 * 4, (double)rows * (16 + 256), st);
        DVQ_LAUNCH(pn_layer1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, pc, C, N, s.Npad, Bc,
                           trans, w1, b1, s.h1);
    }
    DVQ_CHECK_LAUNCH("pn_layer1");
    DVQ_PROPAGATE(dense(s.h1, 64, 64, w2, w2p, b2, rows, 128, 1, sl.h2, 128, st));
    GemmParams p = {};
    p.src[0] = GemmSrc{sl.h2, w3, 128, 128, 128, 0, w3p, 1024L * 128};
    p.nsrc = 1;
    p.M = rows;
    p.N = 1024;
    p.bias = b3;
    p.relu = 0;                  // max(relu(x)) == relu(max(x)): the ReLU is applied by the reduction
    p.rows_per_group = s.Npad;
    p.valid_rows = N;
    p.partial = sl.part;
    DVQ_PROPAGATE(dvq_launch_gemm(p, EPI_COLMAX, st));
    return dvq_launch_colmax_reduce(sl.part, Bc, s.Npad / 128, 1024, relu3, feat, ld_feat, st);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Second stream per (device, caller's stream), with a pool of events: the exact stage and the STN's FC layers of launch i run there,
// beside the trunk kernel of launch i + 1 on the caller's stream.  pn_exact_kernel waits on gathers (71 % of its wave cycles) and
// the trunk kernel issues matrix / vector work: resident together they fill each other's stalls.
struct PnSide {
    hipStream_t s2 = nullptr;
    std::vector<hipEvent_t> ev;
    hipEvent_t event(size_t i) {
        while (ev.size() <= i) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[i];
    }
};
PnSide* side_for(hipStream_t st) {
    static std::mutex lock;
    static std::map<std::pair<int, hipStream_t>, PnSide*> sides;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(lock);
    PnSide*& sd = sides[std::make_pair(dev, st)];
    if (!sd) {
        PnSide* n = new PnSide();
        // diagnostics build, DVQ_PN_S2_PRIO=1: the second stream at the device's highest priority (measured: no effect)
        int lo = 0, hi = 0;
#ifdef DVQ_DIAG
        const bool prio = getenv("DVQ_PN_S2_PRIO") && getenv("DVQ_PN_S2_PRIO")[0] == '1' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
#else
        const bool prio = false;
        (void)lo;
#endif
        if ((prio ? hipStreamCreateWithPriority(&n->s2, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&n->s2, hipStreamNonBlocking)) != hipSuccess) {
            delete n;
            return nullptr;
        }
        sd = n;
    }
    return sd;
}

#define PN_HIP(call, what)                                                                  \
    do {                                                                                    \
        const hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                            \
            dvq_set_error("pointnet_encode: %s failed: %s", what, hipGetErrorString(e__)); \
            return DVQ_ELAUNCH;                                                             \
        }                                                                                   \
    } while (0)

// Both passes (STN, main) of the filtered trunk over the launches of a batch, two streams:
//   S1 (the caller's): front(i) = centres + trunk kernel into scratch set i % slots
//   S2:                back(i) = pn_exact_kernel of that set -> features; pass 1: the STN's FCs of the launch -> its transforms
// front(i) waits for back(i - slots) (its scratch set is free) and, in pass 2, for the transforms of its launch.
int encode_two_streams(const dvq_pointnet_weights* w, const float* pc, int64_t B, int N, float* feat, int64_t ld_feat, float* trans_out,
                       const PnScratch& s, hipStream_t st) {
    PnSide* sd = side_for(st);
    DVQ_REQUIRE(sd, "pointnet_encode: no second stream");
    hipStream_t s2 = sd->s2;
    const long launches = (B + s.chunk - 1) / s.chunk;
    // events: [0] start / join, [1 .. slots] front done, [1 + slots .. 2 slots] set free, then one per launch: transforms ready
    const size_t e_front = 1, e_free = 1 + s.slots, e_tr = 1 + 2 * s.slots;
    DVQ_REQUIRE(sd->event(e_tr + launches), "pointnet_encode: event creation failed");
    unsigned long long* stats = dvq_knobs().pn_stats ? s.stats : nullptr;
    PN_HIP(hipEventRecord(sd->event(0), st), "hipEventRecord");
    PN_HIP(hipStreamWaitEvent(s2, sd->event(0), 0), "hipStreamWaitEvent");        // S2 starts behind everything enqueued on S1 so far
    float* tr_all = trans_out ? trans_out : s.tr;
    int rc = DVQ_OK;
    // inside the loops a failed event / wait call sets rc and leaves them: the join below is ALWAYS attempted, so that nothing S2 still
    // has in flight can touch the caller's workspace, features or transforms after this function has returned
#define PN_TRY(call, what)                                                                      \
    if (rc == DVQ_OK) {                                                                         \
        const hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                                \
            dvq_set_error("pointnet_encode: %s failed: %s", what, hipGetErrorString(e__));     \
            rc = DVQ_ELAUNCH;                                                                   \
        }                                                                                       \
    }
    long idx = 0;
    for (int pass = 0; pass < 2 && rc == DVQ_OK; ++pass)
        for (long c = 0; c < launches && rc == DVQ_OK; ++c, ++idx) {
            const int64_t b0 = c * s.chunk;
            const long Bc = (long)((B - b0 < s.chunk) ? (B - b0) : s.chunk);
            const float* pcb = pc + b0 * (long)w->C * N;
            const PnSlot& sl = s.slot[idx % s.slots];
            float* tr = tr_all + b0 * 9;
            if (idx >= s.slots) PN_TRY(hipStreamWaitEvent(st, sd->event(e_free + idx % s.slots), 0), "hipStreamWaitEvent");
            if (pass == 1) PN_TRY(hipStreamWaitEvent(st, sd->event(e_tr + c), 0), "hipStreamWaitEvent");
            if (rc == DVQ_OK)
                rc = pass == 0 ? dvq_launch_pn_filter_front(pcb, w->C, N, s.Npad, Bc, nullptr, w->s_w1, w->s_b1, w->s_w2, w->s_w2p, w->s_b2, w->s_w3f,
                                                        sl.h2, sl.part, sl.tstat, sl.cbuf, stats, st)
                           : dvq_launch_pn_filter_front(pcb, w->C, N, s.Npad, Bc, tr, w->w1, w->b1, w->w2, w->w2p, w->b2, w->w3f, sl.h2, sl.part,
                                                        sl.tstat, sl.cbuf, stats, st);
            if (rc != DVQ_OK) break;
            PN_TRY(hipEventRecord(sd->event(e_front + idx % s.slots), st), "hipEventRecord");
            PN_TRY(hipStreamWaitEvent(s2, sd->event(e_front + idx % s.slots), 0), "hipStreamWaitEvent");
            if (pass == 0) {
                // STN3d: trunk with ReLU on the last layer, then fc1/fc2 (BN folded, ReLU) and fc3 (+identity)
                if (rc == DVQ_OK) rc = dvq_launch_pn_filter_back(N, s.Npad, Bc, w->s_w3f, w->s_w3, w->s_b3, 1, sl.h2, sl.part, sl.tstat, sl.cbuf, s.f0, 1024, stats, s2);
                PN_TRY(hipEventRecord(sd->event(e_free + idx % s.slots), s2), "hipEventRecord");
                if (rc == DVQ_OK) rc = dense(s.f0, 1024, 1024, w->s_f1, w->s_f1p, w->s_c1, Bc, 512, 1, s.f1, 512, s2);
                if (rc == DVQ_OK) rc = dense(s.f1, 512, 512, w->s_f2, w->s_f2p, w->s_c2, Bc, 256, 1, s.f2, 256, s2);
                if (rc == DVQ_OK) rc = dense(s.f2, 256, 256, w->s_f3, w->s_f3p, w->s_c3, Bc, 9, 0, tr, 9, s2);
                PN_TRY(hipEventRecord(sd->event(e_tr + c), s2), "hipEventRecord");
            } else {
                // main trunk on the transformed cloud; no ReLU after the last BN (pointnet_encoder.py:162)
                if (rc == DVQ_OK) rc = dvq_launch_pn_filter_back(N, s.Npad, Bc, w->w3f, w->w3, w->b3, 0, sl.h2, sl.part, sl.tstat, sl.cbuf, feat + b0 * ld_feat, ld_feat, stats, s2);
                PN_TRY(hipEventRecord(sd->event(e_free + idx % s.slots), s2), "hipEventRecord");
            }
        }
#undef PN_TRY
    // join: whatever happened, the caller's stream continues behind everything enqueued on S2 (best effort: a failure here is reported
    // only when nothing failed before)
    {
        const hipError_t e1 = hipEventRecord(sd->event(0), s2);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(st, sd->event(0), 0) : e1;
        if (e2 != hipSuccess && rc == DVQ_OK) {
            dvq_set_error("pointnet_encode: joining the second stream failed: %s", hipGetErrorString(e2));
            rc = DVQ_ELAUNCH;
        }
    }
    return rc;
}
}  // namespace

extern "C" size_t dvq_pointnet_workspace_bytes(int64_t B, int N) {
    if (B <= 0 || N <= 0) return 256;
    return plan(B, N, nullptr).bytes;
}

extern "C" int dvq_pointnet_encode(const dvq_pointnet_weights* w, const float* pc, int64_t B, int N, float* feat,
                                   int64_t ld_feat, float* trans_out, void* workspace, size_t workspace_bytes,
                                   dvq_stream_t stream) {
    DVQ_REQUIRE(w && pc && feat, "pointnet_encode: null pointer");
    DVQ_REQUIRE(w->C == 3 || w->C == 4, "pointnet_encode: channel count %d not supported (3 or 4)", w->C);
    DVQ_REQUIRE(B >= 0 && N > 0, "pointnet_encode: bad shape B=%ld N=%d", (long)B, N);
    DVQ_REQUIRE(ld_feat >= 1024, "pointnet_encode: ld_feat=%ld < 1024", (long)ld_feat);
    if (B == 0) return DVQ_OK;
    DVQ_REQUIRE(workspace && dvq_aligned16(workspace), "pointnet_encode: null/unaligned workspace");
    const PnScratch s = plan(B, N, workspace);
    if (workspace_bytes < s.bytes) {
        dvq_set_error("pointnet_encode: workspace %zu < %zu bytes", workspace_bytes, s.bytes);
        return DVQ_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool filtered = use_filter(w, N);
    if (filtered && s.slots >= 2 && dvq_knobs().pn_streams && !dvq_knobs().pn_stats) return encode_two_streams(w, pc, B, N, feat, ld_feat, trans_out, s, st);
    for (int64_t b0 = 0; b0 < B; b0 += s.chunk) {
        const long Bc = (long)((B - b0 < s.chunk) ? (B - b0) : s.chunk);
        const float* pcb = pc + b0 * (long)w->C * N;
        // STN3d: trunk with ReLU on the last layer, then fc1/fc2 (BN folded, ReLU) and fc3 (+identity)
        DVQ_PROPAGATE(trunk(pcb, w->C, N, Bc, nullptr, w->s_w1, w->s_b1, w->s_w2, w->s_w2p, w->s_b2, w->s_w3, w->s_w3p, w->s_w3f, w->s_b3,
                            1, filtered, s, s.f0, 1024, st));
        DVQ_PROPAGATE(dense(s.f0, 1024, 1024, w->s_f1, w->s_f1p, w->s_c1, Bc, 512, 1, s.f1, 512, st));
        DVQ_PROPAGATE(dense(s.f1, 512, 512, w->s_f2, w->s_f2p, w->s_c2, Bc, 256, 1, s.f2, 256, st));
        float* tr = (trans_out ? trans_out : s.tr) + b0 * 9;
        DVQ_PROPAGATE(dense(s.f2, 256, 256, w->s_f3, w->s_f3p, w->s_c3, Bc, 9, 0, tr, 9, st));
        // main trunk on the transformed cloud; no ReLU after the last BN (pointnet_encoder.py:162)
        DVQ_PROPAGATE(trunk(pcb, w->C, N, Bc, tr, w->w1, w->b1, w->w2, w->w2p, w->b2, w->w3, w->w3p, w->w3f, w->b3, 0, filtered, s,
                            feat + b0 * ld_feat, ld_feat, st));
    }
    return DVQ_OK;
}

extern "C" size_t dvq_pointnet_filter_bytes(void) { return dvq_pn_filter_image_bytes(); }

extern "C" int dvq_pointnet_fault_counters(uint64_t* out, int reset) {
    DVQ_REQUIRE(out, "pointnet_fault_counters: null pointer");
    unsigned long long v[2] = {0, 0};
    DVQ_PROPAGATE(dvq_pn_fault_counters(v, reset));
    out[0] = v[0];
    out[1] = v[1];
    return DVQ_OK;
}

extern "C" int dvq_pointnet_pack_filter(const float* w2, const float* w3, void* image, dvq_stream_t stream) {
    DVQ_REQUIRE(w2 && w3 && image && dvq_aligned16(image), "pointnet_pack_filter: null/unaligned pointer");
    return dvq_launch_pn_filter_pack(w2, w3, image, (hipStream_t)stream);
}

// =====================================================================================================================
// Fused PointNet trunk: conv1 (C->64, VALU) -> conv2 (64->128) -> conv3 (128->1024) -> max over points, ONE kernel,
// split-bf16 MFMA arithmetic (gemm_bf16x3.hip), nothing but the 16-byte points is read per point and nothing but the
// per-128-point column maxima is written: the [N,64], [N,128] and [N,1024] activations never leave the CU.
//   * a wave owns 32 points (lanes): conv1 builds the h1 fragments in registers; conv2 runs with the WEIGHT fragment as
//     MFMA operand A, so its accumulator (h2 channel on registers, point on lanes) IS the A-operand fragment of conv3 after
//     ReLU + exact 3-way split -- the k order inside each 16-channel block comes out permuted
//     (pos 8h+j <-> channel 8(j>>2)+4h+(j&3)), so the packer stores W3 with the same permutation;
//   * W3 (768 KB of bf16 planes) streams L2 -> LDS by global_load_lds in 24 KB half-chunks (64 output channels x 64 k),
//     double buffered, one barrier per half-chunk; conv3's A operand stays in 96 VGPRs for the whole kernel;
//   * column maxima over the wave's 32 points are lane-local (+ one half swap), waves meet once through LDS at the end.
// Padding points of the last tile duplicate point N-1 (a max ignores duplicates).
namespace {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));

constexpr int T_STAGE = 3 * 64 * 128;                    // one half-chunk: 3 planes x 64 rows x 128 B
constexpr int T_OFF_RED = 2 * T_STAGE;                   // [4 waves][1024] fp32
constexpr int T_OFF_W1 = T_OFF_RED + 4 * 1024 * 4;       // [64][4] fp32
constexpr int T_OFF_B1 = T_OFF_W1 + 64 * 4 * 4;          // [64]
constexpr int T_OFF_B2 = T_OFF_B1 + 64 * 4;              // [128]
constexpr int T_LDS = T_OFF_B2 + 128 * 4;                // 67 328 B -> 2 workgroups per CU

__device__ __forceinline__ void t_split_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
    const unsigned a0 = __float_as_uint(x0) & 0xffff0000u, a1 = __float_as_uint(x1) & 0xffff0000u;
    const float r0 = x0 - __uint_as_float(a0), r1 = x1 - __uint_as_float(a1);
    const unsigned b0 = __float_as_uint(r0) & 0xffff0000u, b1 = __float_as_uint(r1) & 0xffff0000u;
    const float s0 = r0 - __uint_as_float(b0), s1 = r1 - __uint_as_float(b1);
    p1 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    p2 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    p3 = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// eight fp32 values -> three bf16x8 fragments (exact split)
__device__ __forceinline__ void t_split8(const float (&v)[8], pbf16x8 (&out)[3]) {
    unsigned p[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t_split_pair(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl] = __builtin_bit_cast(pbf16x8, uint4{p[pl][0], p[pl][1], p[pl][2], p[pl][3]});
}

// DMA one half-chunk (3 planes x 64 rows x 128 B) of a [rows][ld] bf16 plane set into an LDS stage; 24 x 1 KiB pieces,
// six per wave; rows of 128 B = 8 chunks, chunk c of row r lands at chunk c ^ ((r >> 1) & 7)
__device__ __forceinline__ void t_issue(const uint16_t* __restrict__ planes, long plane_stride, long ld, int row0, int k0,
                                        char* stage, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int id = wave * 6 + i;
        const int pl = id >> 3, rb = id & 7;
        const int row = rb * 8 + (lane >> 3);
        const uint16_t* src = planes + pl * plane_stride + (long)(row0 + row) * ld + k0 + 8 * ((lane & 7) ^ ((row >> 1) & 7));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(stage + (pl * 64 + rb * 8) * 128), 16, 0, 0);
    }
}

__device__ __forceinline__ pbf16x8 t_frag(const char* stage, int pl, int row, int chunk) {
    return *reinterpret_cast<const pbf16x8*>(stage + (pl * 64 + row) * 128 + 16 * (chunk ^ ((row >> 1) & 7)));
}

#define T_MFMA6(ACC, X, Y)                                                            \
    do {                                                                              \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[2], Y[0], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[0], Y[2], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[1], Y[1], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[1], Y[0], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[0], Y[1], ACC, 0, 0, 0);      \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X[0], Y[0], ACC, 0, 0, 0);      \
    } while (0)

template <int C>
__global__ __launch_bounds__(256, 2) void pn_trunk_kernel(const float* __restrict__ pc, const float* __restrict__ trans, int N,
                                                          int tiles, const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const uint16_t* __restrict__ W2p, const float* __restrict__ b2,
                                                          const uint16_t* __restrict__ W3p, const float* __restrict__ b3,
                                                          float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char tl[];
    float* red = reinterpret_cast<float*>(tl + T_OFF_RED);
    float* w1s = reinterpret_cast<float*>(tl + T_OFF_W1);
    float* b1s = reinterpret_cast<float*>(tl + T_OFF_B1);
    float* b2s = reinterpret_cast<float*>(tl + T_OFF_B2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const long b = blockIdx.x / tiles;
    const int tile = blockIdx.x % tiles;

    // W2 planes [3][128][64]: rows of 128 B; rows 0..63 -> stage 0, rows 64..127 -> stage 1
    t_issue(W2p, 128L * 64, 64, 0, 0, tl, wave, lane);
    t_issue(W2p, 128L * 64, 64, 64, 0, tl + T_STAGE, wave, lane);
    w1s[tid] = W1[tid];                                    // 64 x 4
    if (tid < 64) b1s[tid] = b1[tid];
    if (tid < 128) b2s[tid] = b2[tid];

    // ---- conv1 on this lane's point (both lane halves hold the same point, different k)
    int pidx = tile * 128 + wave * 32 + r;
    if (pidx >= N) pidx = N - 1;
    const float* src = pc + b * (long)C * N + pidx;
    float x0 = src[0], x1 = src[N], x2 = src[2L * N];
    const float x3 = (C > 3) ? src[3L * N] : 0.f;
    if (trans) {                                           // xyz @ trans[b]  (pointnet_encoder.py:146)
        const float* t = trans + b * 9;
        const float n0 = fmaf(x2, t[6], fmaf(x1, t[3], x0 * t[0]));
        const float n1 = fmaf(x2, t[7], fmaf(x1, t[4], x0 * t[1]));
        const float n2 = fmaf(x2, t[8], fmaf(x1, t[5], x0 * t[2]));
        x0 = n0; x1 = n1; x2 = n2;
    }
    dvq_dma_barrier();                                     // W1/b1/b2 visible, W2 planes landed
    pbf16x8 h1f[4][3];                                     // B-operand fragments of conv2: k = 16 s + 8 h + j
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s + 8 * h + j;
            const f32x4 w = *reinterpret_cast<const f32x4*>(w1s + 4 * k);
            float a = x0 * w[0];
            a = fmaf(x1, w[1], a);
            a = fmaf(x2, w[2], a);
            a = fmaf(x3, w[3], a);
            v[j] = fmaxf(a + b1s[k], 0.f);
        }
        t_split8(v, h1f[s]);
    }
    // ---- conv2: weight fragment as operand A -> accumulator = (h2 channel on registers, point on lanes)
    pbf16x8 a3[8][3];                                      // conv3 A operand: step = 2 * tile32 + q
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {                       // 32 h2 channels per tile
        const char* st = tl + (t4 >> 1) * T_STAGE;
        const int row = 32 * (t4 & 1) + r;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            pbf16x8 wf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[pl] = t_frag(st, pl, row, 2 * s + h);
            T_MFMA6(acc, wf, h1f[s]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * q + j;
                const int ch = 32 * t4 + (e & 3) + 8 * (e >> 2) + 4 * h;
                v[j] = fmaxf(acc[e] + b2s[ch], 0.f);
            }
            t_split8(v, a3[2 * t4 + q]);
        }
    }
    dvq_lds_barrier();                                       // everybody is done with W2 in the stages
    // ---- conv3: 16 chunks of 64 output channels, each in two K halves of 64
    t_issue(W3p, 1024L * 128, 128, 0, 0, tl, wave, lane);
    int stage = 0;
    for (int c = 0; c < 16; ++c) {
        f32x16 acc2[2];
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[jn][e] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            dvq_dma_barrier();                             // this half-chunk landed everywhere; the other stage is free
            const int nxt = 2 * c + kh + 1;
            if (nxt < 32) t_issue(W3p, 1024L * 128, 128, 64 * (nxt >> 1), 64 * (nxt & 1), tl + (stage ^ 1) * T_STAGE, wave, lane);
            const char* st = tl + stage * T_STAGE;
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    pbf16x8 wf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wf[pl] = t_frag(st, pl, 32 * jn + r, 2 * sq + h);
                    T_MFMA6(acc2[jn], a3[4 * kh + sq], wf);
                }
            }
            stage ^= 1;
        }
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {                   // max over this wave's 32 points (rows = registers + lane half)
            const int n = 64 * c + 32 * jn + r;
            const float bias = b3[n];
            float m = acc2[jn][0];
#pragma unroll
            for (int e = 1; e < 16; ++e) m = fmaxf(m, acc2[jn][e]);
            m = fmaxf(m, __shfl_xor(m, 32)) + bias;         // max(x + b) == max(x) + b
            if (h == 0) red[wave * 1024 + n] = m;
        }
    }
    dvq_lds_barrier();
    for (int col = tid; col < 1024; col += 256)
        partial[(long)blockIdx.x * 1024 + col] = fmaxf(fmaxf(red[col], red[1024 + col]), fmaxf(red[2048 + col], red[3072 + col]));
}

}  // namespace

int dvq_launch_pn_trunk(const float* pc, int C, int N, long B, const float* trans, const float* W1, const float* b1,
                        const uint16_t* W2p, const float* b2, const uint16_t* W3p, const float* b3, float* partial,
                        hipStream_t st) {
    const int tiles = (N + 127) / 128;
    static DvqOncePerDevice attr_once;
    {
        const hipError_t e = attr_once.run([] {
            const hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_trunk_kernel<3>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
            const hipError_t e4 = hipFuncSetAttribute(reinterpret_cast<const void*>(&pn_trunk_kernel<4>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
            return e3 != hipSuccess ? e3 : e4;
        });
        if (e != hipSuccess) {
            dvq_set_error("pointnet: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return DVQ_ELAUNCH;
        }
    }
    const long grid = B * tiles;
    DVQ_REQUIRE(grid < (1L << 31), "pointnet: grid too large");
    const double pts = (double)B * tiles * 128;
    {
        DVQ_PROF("pn_trunk", 2.0 * pts * (4.0 * 64 + 64.0 * 128 + 128.0 * 1024), pts * 16 + (double)grid * 4096, st);
        if (C == 3)
            DVQ_LAUNCH(pn_trunk_kernel<3>, dim3((unsigned)grid), dim3(256), T_LDS, st, pc, trans, N, tiles, W1, b1, W2p, b2, W3p, b3, partial);
        else
            DVQ_LAUNCH(pn_trunk_kernel<4>, dim3((unsigned)grid), dim3(256), T_LDS, st, pc, trans, N, tiles, W1, b1, W2p, b2, W3p, b3, partial);
    }
    DVQ_CHECK_LAUNCH("pn_trunk");
    return DVQ_OK;
}
