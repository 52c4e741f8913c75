// Gated PixelCNN prior on the 3x3 latent grid: cached (incremental) sampler and teacher-forced forward.
// Reference: GatedPixelCNN.generate / .forward, network/pixelcnn/models.py:161-198 with
// GatedMaskedConv2d.forward :65-88.  The network is exactly causal, so every grid position is evaluated
// once: the vertical stack of row r (all layers, 3 columns) as soon as rows < r are sampled, then the
// horizontal stack + head position by position.  Every conv tap is one K=dim source of a multi-source
// fp32-MFMA GEMM whose epilogue applies bias + class-conditional bias + tanh*sigmoid gate (gate-packed
// channels) or the residual add; taps that fall into the zero padding are simply skipped.
//
// Geometry (models.py:41-55): layer 0 has k=5 and mask 'A' (last kernel row of the vertical stack and
// last kernel column of the horizontal stack are zero: applied by skipping those taps), layers >= 1
// have k=3 (mask 'B').  vertical: kernel (k/2+1, k), pad (k/2, k/2), output row r sees rows r-k/2..r;
// horizontal: kernel (1, k/2+1), pad (0, k/2), output col c sees cols c-k/2..c.
#include "dvq_internal.h"

namespace {

constexpr int GRID = 3, NPOS = 9;

__global__ void sanitize_labels_kernel(const int64_t* __restrict__ label, long B, int n_classes, int64_t* __restrict__ out,
                                       int32_t* err_flag) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t l = label[b];
    if (l < 0 || l >= n_classes) {
        if (err_flag) atomicOr(err_flag, 1);
        l = 0;
    }
    out[b] = l;
}

// out = gate(bias + cls[label]) (and pre = bias) for a position with no in-range taps
__global__ void bias_gate_kernel(const float* __restrict__ bias, const float* __restrict__ cls, const int64_t* __restrict__ label,
                                 long M, int dim, float* __restrict__ out, float* __restrict__ pre) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long m = gid / dim;
    const int c = (int)(gid % dim);
    if (m >= M) return;
    // gate-packed index of natural channel c: block c/64, wave (c%64)/32, lane c%32; sigmoid partner +32
    const int na = (c / 64) * 128 + ((c % 64) / 32) * 64 + (c % 32);
    const int nb = na + 32;
    float a = bias[na], g = bias[nb];
    if (pre) {
        pre[m * 2 * dim + na] = a;
        pre[m * 2 * dim + nb] = g;
    }
    const float* crow = cls + label[m] * 2 * dim;
    a += crow[na];
    g += crow[nb];
    out[m * dim + c] = tanhf(a) * (1.0f / (1.0f + expf(-g)));
}

// One wave per sample: p = softmax(logits); code = argmax_k p_k / q_k (exponential race, lowest k on ties);
// then the token embedding of the drawn code is written as level-0 activation of this position.
__global__ void sample_kernel(const float* __restrict__ logits, const float* __restrict__ noise /* [B,9,n_in] at this chunk */,
                              int pos, int n_in, long Bc, const int64_t* __restrict__ forced /* [B,9] or null */,
                              int64_t* __restrict__ codes /* [B,9] */, const float* __restrict__ tok_emb, int dim,
                              float* __restrict__ x0 /* [Bc,dim] */, float* __restrict__ logits_out /* [B,9,n_in] or null */,
                              int32_t* err_flag, const int64_t* __restrict__ lrow /* null, or: sample b's logits are row lrow[b] */) {
    const int lane = threadIdx.x & 63;
    const long b = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= Bc) return;
    const float* lg = logits + (lrow ? lrow[b] : b) * n_in;
    int64_t code;
    if (logits_out)
        for (int k = lane; k < n_in; k += 64) logits_out[(b * NPOS + pos) * n_in + k] = lg[k];
    if (forced) {
        code = forced[b * NPOS + pos];
        if (code < 0 || code >= n_in) {
            if (lane == 0 && err_flag) atomicOr(err_flag, 1);
            code = 0;
        }
    } else {
        float mx = -INFINITY;
        for (int k = lane; k < n_in; k += 64) mx = fmaxf(mx, lg[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int k = lane; k < n_in; k += 64) sum += expf(lg[k] - mx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float* q = noise + (b * NPOS + pos) * n_in;
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int k = lane; k < n_in; k += 64) {
            const float p = expf(lg[k] - mx) / sum;
            const float s = p / q[k];
            if (s > best || (s == best && k < bi)) { best = s; bi = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        const bool nan_draw = bi == 0x7fffffff;
        if (nan_draw) {                    // all scores NaN: torch.argmax would return the first NaN; keep 0 -- and say so (bit 2 of the
            bi = 0;                        // flag: non-finite logits, e.g. an activation beyond the fp16 range of the default GEMM
            if (lane == 0 && err_flag) atomicOr(err_flag, 4);       // arithmetic; the host mirror generates the batch again on the bf16 split)
        }
        code = bi;
        // the caller's copy of a draw from all-NaN logits is -1 (include/dvq.h): the host mirror regenerates exactly those rows
        if (lane == 0) codes[b * NPOS + pos] = nan_draw ? -1 : code;
    }
    const float* e = tok_emb + code * dim;
    for (int c = lane * 4; c < dim; c += 256) *reinterpret_cast<f32x4*>(x0 + b * dim + c) = *reinterpret_cast<const f32x4*>(e + c);
}

__global__ void iota_kernel(int64_t* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i;
}

struct Plan {
    long chunk;
    int L, dim;
    float *xv, *xh, *hv, *g, *hid, *lg;
    int64_t* lab;
    size_t bytes;
    // Class tables.  Row 0 of the grid sees nothing above it and position (0, 0) nothing before it, so the vertical stack of row 0
    // (all layers, three columns) and the whole horizontal stack + head of position (0, 0) are functions of the CLASS LABEL alone:
    // they are evaluated once per class (nc rows instead of one per sample) and read through the label wherever the batch needs
    // them (GemmSrc::arow, the draw kernel's row index) -- the same kernels per row, hence the same bits.
    bool tab;
    bool prebuilt;                                         // the tables are the weights' own (dvq_pixelcnn_build_tables), not the workspace's
    int nc;
    float *xvc, *hvc, *xhc, *gc, *hidc, *lgc;
    int64_t* iota;
    // Accumulator states.  In the vertical stack of row 1 (layers >= 1) the taps on row 0 come FIRST in the K order, in the
    // horizontal stack of positions (0, 1) and (0, 2) the vertical-to-horizontal source (and (0, 1)'s tap on (0, 0)) does: these
    // leading sources are class-table rows.  The accumulator pair after them is stored once per class (EPI_STATE) and the batch's
    // launch continues it with the remaining sources (GemmParams::acc_*, row = label): the same MFMA sequence per output, bit for
    // bit, at roughly half the K.
    float *sv, *sh;                                        // [L][GRID][2][nc][2 dim], [L][2][2][nc][2 dim]
    float* SV(int layer, int c, int which) const { return sv + (((size_t)layer * GRID + c) * 2 + which) * nc * 2 * dim; }
    float* SH(int layer, int c, int which) const { return sh + (((size_t)layer * 2 + (c - 1)) * 2 + which) * nc * 2 * dim; }
    float* XVC(int level, int c) const { return xvc + ((size_t)(level - 1) * GRID + c) * nc * dim; }      // level 1 .. L, row 0, column c
    float* HVC(int layer, int c) const { return hvc + ((size_t)layer * GRID + c) * nc * 2 * dim; }
    float* XHC(int level) const { return xhc + (size_t)(level - 1) * nc * dim; }                          // level 1 .. L, position (0, 0)
    float* XV(int level, int pos) const { return xv + ((size_t)level * NPOS + pos) * chunk * dim; }
    float* XH(int level, int pos) const { return level == 0 ? XV(0, pos) : xh + ((size_t)(level - 1) * NPOS + pos) * chunk * dim; }
    float* HV(int layer, int col) const { return hv + ((size_t)layer * GRID + col) * chunk * 2 * dim; }
};

bool tables_possible(const dvq_pixelcnn_weights* w) {
    return dvq_knobs().pixelcnn_tables && w->planes_kind == DVQ_PLANES_F16X2 && w->w0_p && dvq_gemm_mode() == 1;
}

// the class-table pointers of ``p`` inside ``base`` (null: sizes only); returns the bytes
size_t carve_tables(Plan& p, const dvq_pixelcnn_weights* w, char* base) {
    char* c = base;
    auto take = [&](size_t n) { char* q = c; c += dvq_round_up(n, 256); return q; };
    const size_t nc = (size_t)w->n_classes, L = (size_t)w->n_layers, dim = (size_t)w->dim;
    p.nc = w->n_classes;
    p.xvc = (float*)take(L * GRID * nc * dim * 4);
    p.hvc = (float*)take(L * GRID * nc * 2 * dim * 4);
    p.xhc = (float*)take(L * nc * dim * 4);
    p.gc = (float*)take(nc * dim * 4);
    p.hidc = (float*)take(nc * w->n_hidden * 4);
    p.lgc = (float*)take(nc * w->n_in * 4);
    p.iota = (int64_t*)take(nc * 8);
    p.sv = (float*)take(L * GRID * 2 * nc * 2 * dim * 4);
    p.sh = (float*)take(L * 2 * 2 * nc * 2 * dim * 4);
    return (size_t)(c - base);
}

Plan make_plan(const dvq_pixelcnn_weights* w, int64_t B, void* ws) {
    Plan p;
    p.L = w->n_layers;
    p.dim = w->dim;
    long chunk = 16384;
    if (dvq_knobs().pixelcnn_chunk > 0) chunk = dvq_knobs().pixelcnn_chunk;
    if (chunk > B) chunk = B;
    if (chunk < 1) chunk = 1;
    p.chunk = chunk;
    char* c = (char*)ws;
    auto take = [&](size_t n) { char* q = c; c += dvq_round_up(n, 256); return q; };
    p.xv = (float*)take((size_t)(p.L + 1) * NPOS * chunk * p.dim * 4);
    p.xh = (float*)take((size_t)p.L * NPOS * chunk * p.dim * 4);
    p.hv = (float*)take((size_t)p.L * GRID * chunk * 2 * p.dim * 4);
    p.g = (float*)take((size_t)chunk * p.dim * 4);
    p.hid = (float*)take((size_t)chunk * w->n_hidden * 4);
    p.lg = (float*)take((size_t)chunk * w->n_in * 4);
    p.lab = (int64_t*)take((size_t)chunk * 8);
    // class tables: fp16-plane kernels only (they read activation rows through an index).  The weights' own when they carry them
    // (any batch size); else built per call in the workspace by batches that are at least twice the table
    p.nc = w->n_classes;
    const bool can = tables_possible(w);
    p.prebuilt = can && w->class_tables != nullptr;
    p.tab = can && (p.prebuilt || B >= 2L * w->n_classes);
    p.xvc = p.hvc = p.xhc = p.gc = p.hidc = p.lgc = nullptr;
    p.iota = nullptr;
    p.sv = p.sh = nullptr;
    if (p.prebuilt) (void)carve_tables(p, w, (char*)w->class_tables);
    else if (p.tab) c += carve_tables(p, w, c);
    p.bytes = (size_t)(c - (char*)ws);
    return p;
}

int check_weights(const dvq_pixelcnn_weights* w) {
    DVQ_REQUIRE(w && w->layers_host && w->tok_emb && w->w0 && w->b0 && w->w2 && w->b2, "pixelcnn: null weights");
    DVQ_REQUIRE(w->n_layers >= 1 && w->dim >= 64 && w->dim % 64 == 0, "pixelcnn: dim=%d must be a multiple of 64", w->dim);
    DVQ_REQUIRE(w->n_hidden % 32 == 0 && w->n_in >= 1 && w->n_classes >= 1, "pixelcnn: bad head sizes");
    DVQ_REQUIRE(w->planes_kind == DVQ_PLANES_BF16X3 || w->planes_kind == DVQ_PLANES_F16X2, "pixelcnn: unknown planes_kind %d", w->planes_kind);
    if (w->planes_kind == DVQ_PLANES_F16X2 && w->w0_p) {
        DVQ_REQUIRE(w->s0 && w->s2, "pixelcnn: fp16 planes without row scales (head)");
        for (int l = 0; l < w->n_layers; ++l)
            DVQ_REQUIRE(w->layers_host[l].sv && w->layers_host[l].sh && w->layers_host[l].sr, "pixelcnn: fp16 planes without row scales (layer %d)", l);
    }
    return DVQ_OK;
}

// ``build`` non-null: only the class tables of ``w``, into that buffer (dvq_pixelcnn_build_tables)
int run(const dvq_pixelcnn_weights* w, const int64_t* label, const float* noise, const int64_t* forced, int64_t B,
        int64_t* codes, float* logits_out, int32_t* err_flag, void* ws, size_t ws_bytes, hipStream_t st, void* build = nullptr) {
    DVQ_PROPAGATE(check_weights(w));
    Plan pl;
    if (build) {
        pl = Plan{};
        pl.L = w->n_layers;
        pl.dim = w->dim;
        pl.tab = true;
        (void)carve_tables(pl, w, (char*)build);
    } else {
        DVQ_REQUIRE(B >= 0, "pixelcnn: negative batch");
        if (B == 0) return DVQ_OK;
        DVQ_REQUIRE(label && (forced || (noise && codes)), "pixelcnn: null input");
        DVQ_REQUIRE(ws && dvq_aligned16(ws), "pixelcnn: null/unaligned workspace");
        pl = make_plan(w, B, ws);
        if (ws_bytes < pl.bytes) {
            dvq_set_error("pixelcnn: workspace %zu < %zu bytes", ws_bytes, pl.bytes);
            return DVQ_EWORKSPACE;
        }
    }
    const int dim = w->dim, L = w->n_layers;
    const int kind = w->planes_kind;                      // DVQ_PLANES_*: what every *_p image of this network holds

    // An activation tensor as a GEMM source: rows of the chunk, or rows of a class table read through the chunk's labels.
    struct Act { const float* p; const int64_t* arow; };
    // ---- one layer of the vertical stack at grid row r for the M rows of (label, xv, out, pre): the row's columns do not depend on
    // each other -- one launch where the launch count is what costs (dvq_launch_gemm_gate_group)
    // ``mode`` 0: the whole product.  1: only the leading sources that are class-table rows, on the table's own rows (M = classes),
    // accumulator pair -> state_of(col, 0 / 1).  2: the remaining sources, continuing state_of(col, .) through the labels.
    auto vertical_layer = [&](int r, int l, long M, const int64_t* lab, auto xv_of /* (level, pos) -> Act */,
                              auto out_of /* (col) -> float* (level l + 1) */, auto pre_of /* (col) -> float* */,
                              int mode, auto state_of /* (col, which) -> float* */) -> int {
        const dvq_pixelcnn_layer& ly = w->layers_host[l];
        const int k = (l == 0) ? 5 : 3, pad = k / 2, KR = k / 2 + 1;
        GemmParams grp[GRID];
        int ngrp = 0;
        for (int c = 0; c < GRID; ++c) {
            GemmParams& g = grp[ngrp];
            g = GemmParams{};
            int ns = 0;
            for (int kr = 0; kr < KR; ++kr) {
                if (l == 0 && kr == KR - 1) continue;            // mask 'A': last kernel row
                const int ir = r - pad + kr;
                if (ir < 0) continue;
                for (int kc = 0; kc < k; ++kc) {
                    const int ic = c - pad + kc;
                    if (ic < 0 || ic >= GRID) continue;
                    const size_t tap = (size_t)(kr * k + kc), wsz = (size_t)2 * dim * dim;
                    const Act a = xv_of(l, ir * GRID + ic);
                    g.src[ns++] = GemmSrc{a.p, ly.wv + tap * wsz, (long)dim, (long)dim, dim, kind,
                                          ly.wv_p ? ly.wv_p + tap * wsz : nullptr, (long)(KR * k) * (long)wsz, a.arow};
                }
            }
            int lead = 0;                                    // leading sources read through the labels (class-table rows)
            while (lead < ns && g.src[lead].arow) ++lead;
            if (mode == 1) {                                 // the state of those sources, on the table's rows
                if (lead == 0 || lead == ns) continue;
                for (int i = 0; i < lead; ++i) g.src[i].arow = nullptr;
                g.nsrc = lead;
                g.M = M;
                g.prof_cls = 1;
                g.N = 2 * dim;
                g.wscale = ly.sv;
                g.out = state_of(c, 0);
                g.ldo = 2 * dim;
                g.pre = state_of(c, 1);
                g.ldpre = 2 * dim;
                DVQ_PROPAGATE(dvq_launch_gemm(g, EPI_STATE, st));
                continue;
            }
            if (mode == 2 && lead > 0 && lead < ns) {        // continue the class's state with the sources that are per row
                for (int i = lead; i < ns; ++i) g.src[i - lead] = g.src[i];
                ns -= lead;
                g.acc_hi = state_of(c, 0);
                g.acc_lo = state_of(c, 1);
                g.ldacc = 2 * dim;
                g.acc_row = lab;
            }
            float* out = out_of(c);
            float* pre = pre_of(c);
            if (ns == 0) {
                const long tot = M * dim;
                DVQ_LAUNCH(bias_gate_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, ly.bv, ly.cls, lab, M, dim, out, pre);
                DVQ_CHECK_LAUNCH("bias_gate");
                continue;
            }
            g.nsrc = ns;
            g.M = M;
            g.prof_cls = lab == pl.iota;
            g.N = 2 * dim;
            g.bias = ly.bv;
            g.wscale = ly.sv;
            g.cls = ly.cls;
            g.label = lab;
            g.out = out;
            g.ldo = dim;
            g.pre = pre;
            g.ldpre = 2 * dim;
            ++ngrp;
        }
        if (ngrp) DVQ_PROPAGATE(dvq_launch_gemm_gate_group(grp, ngrp, st));
        return DVQ_OK;
    };
    // ---- the horizontal stack + head of one position for M rows; logits -> lg
    auto horizontal_position = [&](int r, int c, long M, const int64_t* lab, auto hv_of /* (layer, col) -> Act */,
                                   auto xh_of /* (level, pos) -> Act */, auto xh_out /* (level) -> float* at this position */,
                                   float* gbuf, float* hid, float* lg, int mode, auto state_of /* (layer, which) -> float* */) -> int {
        const int pos = r * GRID + c;
        for (int l = 0; l < L; ++l) {
            const dvq_pixelcnn_layer& ly = w->layers_host[l];
            const int k = (l == 0) ? 5 : 3, pad = k / 2, KC = k / 2 + 1;
            GemmParams g = {};
            int ns = 0;
            {
                const Act a = hv_of(l, c);
                g.src[ns++] = GemmSrc{a.p, ly.wv2h, (long)2 * dim, (long)2 * dim, 2 * dim, kind, ly.wv2h_p, (long)4 * dim * dim, a.arow};
            }
            for (int kc = 0; kc < KC; ++kc) {
                if (l == 0 && kc == KC - 1) continue;            // mask 'A': last kernel column
                const int ic = c - pad + kc;
                if (ic < 0) continue;
                const size_t wsz = (size_t)2 * dim * dim;
                const Act a = xh_of(l, r * GRID + ic);
                g.src[ns++] = GemmSrc{a.p, ly.wh + kc * wsz, (long)dim, (long)dim, dim, kind,
                                      ly.wh_p ? ly.wh_p + kc * wsz : nullptr, (long)KC * (long)wsz, a.arow};
            }
            int lead = 0;
            while (lead < ns && g.src[lead].arow) ++lead;
            if (mode == 1) {
                if (lead == 0 || lead == ns) continue;
                for (int i = 0; i < lead; ++i) g.src[i].arow = nullptr;
                g.nsrc = lead;
                g.M = M;
                g.prof_cls = 1;
                g.N = 2 * dim;
                g.wscale = ly.sh;
                g.out = state_of(l, 0);
                g.ldo = 2 * dim;
                g.pre = state_of(l, 1);
                g.ldpre = 2 * dim;
                DVQ_PROPAGATE(dvq_launch_gemm(g, EPI_STATE, st));
                continue;
            }
            if (mode == 2 && lead > 0 && lead < ns) {
                for (int i = lead; i < ns; ++i) g.src[i - lead] = g.src[i];
                ns -= lead;
                g.acc_hi = state_of(l, 0);
                g.acc_lo = state_of(l, 1);
                g.ldacc = 2 * dim;
                g.acc_row = lab;
            }
            g.nsrc = ns;
            g.M = M;
            g.prof_cls = lab == pl.iota;
            g.N = 2 * dim;
            g.bias = ly.bh;
            g.wscale = ly.sh;
            g.cls = ly.cls;
            g.label = lab;
            g.out = gbuf;
            g.ldo = dim;
            DVQ_PROPAGATE(dvq_launch_gemm(g, EPI_GATE, st));
            GemmParams q = {};
            q.src[0] = GemmSrc{gbuf, ly.wr, (long)dim, (long)dim, dim, kind, ly.wr_p, (long)dim * dim, nullptr};
            q.wscale = ly.sr;
            q.nsrc = 1;
            q.M = M;
            q.prof_cls = lab == pl.iota;
            q.N = dim;
            q.bias = ly.br;
            q.out = xh_out(l + 1);
            q.ldo = dim;
            if (l > 0) {                                        // residual only for layers >= 1 (:82-86)
                q.resid = xh_of(l, pos).p;                      // this position's own level-l activations: never a table read through labels
                q.ldr = dim;
                DVQ_PROPAGATE(dvq_launch_gemm(q, EPI_RESID, st));
            } else {
                DVQ_PROPAGATE(dvq_launch_gemm(q, EPI_BIAS, st));
            }
        }
        if (mode == 1) return DVQ_OK;                        // states only
        GemmParams h0 = {};
        h0.src[0] = GemmSrc{xh_out(L), w->w0, (long)dim, (long)dim, dim, kind, w->w0_p, (long)w->n_hidden * dim, nullptr};
        h0.wscale = w->s0;
        h0.nsrc = 1; h0.M = M; h0.prof_cls = lab == pl.iota; h0.N = w->n_hidden; h0.bias = w->b0; h0.out = hid; h0.ldo = w->n_hidden; h0.relu = 1;
        DVQ_PROPAGATE(dvq_launch_gemm(h0, EPI_BIAS, st));
        GemmParams h2 = {};
        h2.src[0] = GemmSrc{hid, w->w2, (long)w->n_hidden, (long)w->n_hidden, w->n_hidden, kind, w->w2_p, (long)w->n_in * w->n_hidden, nullptr};
        h2.wscale = w->s2;
        h2.nsrc = 1; h2.M = M; h2.prof_cls = lab == pl.iota; h2.N = w->n_in; h2.bias = w->b2; h2.out = lg; h2.ldo = w->n_in;
        DVQ_PROPAGATE(dvq_launch_gemm(h2, EPI_BIAS, st));
        return DVQ_OK;
    };

    if (pl.tab && !pl.prebuilt) {
        // ---- once per call (once per model with dvq_pixelcnn_build_tables): row 0's vertical stack and position (0, 0)'s
        // horizontal stack + head for every class (label = row)
        const long nc = pl.nc;
        DVQ_LAUNCH(iota_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, pl.iota, (int)nc);
        DVQ_CHECK_LAUNCH("iota");
        auto no_state = [](int, int) -> float* { return nullptr; };
        auto xv_c = [&](int level, int pos) { return Act{pl.XVC(level, pos), nullptr}; };     // row 0 only reads row 0 (levels >= 1)
        for (int l = 0; l < L; ++l)
            DVQ_PROPAGATE(vertical_layer(0, l, nc, pl.iota, xv_c, [&](int c) { return pl.XVC(l + 1, c); }, [&](int c) { return pl.HVC(l, c); },
                                         0, no_state));
        auto hv_c = [&](int layer, int col) { return Act{pl.HVC(layer, col), nullptr}; };
        auto xh_c = [&](int level, int pos) { (void)pos; return Act{pl.XHC(level), nullptr}; }; // position (0, 0) reads itself only, levels >= 1
        DVQ_PROPAGATE(horizontal_position(0, 0, nc, pl.iota, hv_c, xh_c, [&](int level) { return pl.XHC(level); }, pl.gc, pl.hidc, pl.lgc,
                                          0, no_state));
        // ---- the accumulator states of the leading class-only sources: vertical stack of row 1, positions (0, 1) and (0, 2).  The
        // source lists are built exactly as for the batch (table rows marked by a row index); mode 1 keeps the marked prefix.
        auto xv_m = [&](int level, int pos) {
            if (level >= 1 && pos < GRID) return Act{pl.XVC(level, pos), pl.iota};
            return Act{pl.XV(level, pos), nullptr};                                            // never read in mode 1
        };
        for (int l = 0; l < L; ++l)
            DVQ_PROPAGATE(vertical_layer(1, l, nc, pl.iota, xv_m, [&](int) -> float* { return nullptr; }, [&](int) -> float* { return nullptr; },
                                         1, [&](int c, int which) { return pl.SV(l, c, which); }));
        auto hv_m = [&](int layer, int col) { return Act{pl.HVC(layer, col), pl.iota}; };
        auto xh_m = [&](int level, int pos) {
            if (level >= 1 && pos == 0) return Act{pl.XHC(level), pl.iota};
            return Act{pl.XH(level, pos), nullptr};
        };
        for (int c = 1; c < GRID; ++c)
            DVQ_PROPAGATE(horizontal_position(0, c, nc, pl.iota, hv_m, xh_m, [&](int) -> float* { return nullptr; }, nullptr, nullptr, nullptr,
                                              1, [&](int layer, int which) { return pl.SH(layer, c, which); }));
    }
    if (build) return DVQ_OK;
    for (int64_t b0 = 0; b0 < B; b0 += pl.chunk) {
        const long Bc = (long)((B - b0 < pl.chunk) ? (B - b0) : pl.chunk);
        DVQ_LAUNCH(sanitize_labels_kernel, dim3((unsigned)((Bc + 255) / 256)), dim3(256), 0, st, label + b0, Bc,
                           w->n_classes, pl.lab, err_flag);
        DVQ_CHECK_LAUNCH("sanitize_labels");
        if (forced)   // teacher forcing: level-0 activations of all nine positions are known up front
            for (int pos = 0; pos < NPOS; ++pos)
                DVQ_PROPAGATE(dvq_launch_gather_rows(w->tok_emb, forced + b0 * NPOS + pos, NPOS, Bc, w->n_in, dim,
                                                     pl.XV(0, pos), dim, err_flag, st));
        // where the batch finds an activation tensor: level 0 = the token embeddings (always per row); row 0's vertical levels >= 1
        // and position (0, 0)'s horizontal levels >= 1 in the class tables, through the labels
        auto xv_b = [&](int level, int pos) {
            if (pl.tab && level >= 1 && pos < GRID) return Act{pl.XVC(level, pos), pl.lab};
            return Act{pl.XV(level, pos), nullptr};
        };
        auto hv_b = [&](int layer, int col, int r) {
            if (pl.tab && r == 0) return Act{pl.HVC(layer, col), pl.lab};
            return Act{pl.HV(layer, col), nullptr};
        };
        auto xh_b = [&](int level, int pos) {
            if (pl.tab && level >= 1 && pos == 0) return Act{pl.XHC(level), pl.lab};
            return Act{pl.XH(level, pos), nullptr};
        };
        for (int r = 0; r < GRID; ++r) {
            // ---- vertical stack of row r, all layers (depends on rows < r only)
            if (!(pl.tab && r == 0))
                for (int l = 0; l < L; ++l)
                    DVQ_PROPAGATE(vertical_layer(r, l, Bc, pl.lab, xv_b, [&](int c) { return pl.XV(l + 1, r * GRID + c); },
                                                 [&](int c) { return pl.HV(l, c); }, (pl.tab && r == 1) ? 2 : 0,
                                                 [&](int c, int which) { return pl.SV(l, c, which); }));
            // ---- horizontal stack + head + draw, position by position
            for (int c = 0; c < GRID; ++c) {
                const int pos = r * GRID + c;
                const bool from_table = pl.tab && pos == 0;     // its logits are the class table's rows
                if (!from_table)
                    DVQ_PROPAGATE(horizontal_position(r, c, Bc, pl.lab, [&](int layer, int col) { return hv_b(layer, col, r); }, xh_b,
                                                      [&](int level) { return pl.XH(level, pos); }, pl.g, pl.hid, pl.lg,
                                                      (pl.tab && r == 0) ? 2 : 0,
                                                      [&](int layer, int which) { return pl.SH(layer, c, which); }));
                {
                DVQ_PROF("pixelcnn_draw", 0, (double)Bc * (2.0 * w->n_in + dim) * 4, st);
                DVQ_LAUNCH(sample_kernel, dim3((unsigned)((Bc + 3) / 4)), dim3(256), 0, st, from_table ? pl.lgc : pl.lg,
                                   noise ? noise + b0 * NPOS * w->n_in : nullptr, pos, w->n_in, Bc,
                                   forced ? forced + b0 * NPOS : nullptr, codes ? codes + b0 * NPOS : nullptr, w->tok_emb, dim,
                                   pl.XV(0, pos), logits_out ? logits_out + b0 * NPOS * w->n_in : nullptr, err_flag,
                                   from_table ? pl.lab : nullptr);
                }
                DVQ_CHECK_LAUNCH("pixelcnn_sample_step");
            }
        }
    }
    return DVQ_OK;
}

}  // namespace

extern "C" size_t dvq_pixelcnn_tables_bytes(const dvq_pixelcnn_weights* w) {
    if (!w || w->planes_kind != DVQ_PLANES_F16X2 || !w->w0_p || dvq_gemm_mode() != 1) return 0;
    Plan p = {};
    return carve_tables(p, w, nullptr);
}

extern "C" int dvq_pixelcnn_build_tables(const dvq_pixelcnn_weights* w, void* tables, size_t tables_bytes, dvq_stream_t stream) {
    DVQ_REQUIRE(w && tables && dvq_aligned16(tables), "pixelcnn_build_tables: null / unaligned buffer");
    DVQ_REQUIRE(w->planes_kind == DVQ_PLANES_F16X2 && w->w0_p && dvq_gemm_mode() == 1,
                "pixelcnn_build_tables: class tables exist for fp16 weight images only (DVQ_PLANES_F16X2)");
    DVQ_REQUIRE(tables_bytes >= dvq_pixelcnn_tables_bytes(w), "pixelcnn_build_tables: buffer %zu < %zu bytes", tables_bytes,
                dvq_pixelcnn_tables_bytes(w));
    return run(w, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, tables);
}

extern "C" size_t dvq_pixelcnn_workspace_bytes(const dvq_pixelcnn_weights* w, int64_t B) {
    if (!w || B <= 0) return 256;
    return make_plan(w, B, nullptr).bytes;
}

extern "C" int dvq_pixelcnn_sample(const dvq_pixelcnn_weights* w, const int64_t* label, const float* noise, int64_t B,
                                   int64_t* codes, float* logits_out, int32_t* err_flag, void* workspace,
                                   size_t workspace_bytes, dvq_stream_t stream) {
    return run(w, label, noise, nullptr, B, codes, logits_out, err_flag, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int dvq_pixelcnn_forward(const dvq_pixelcnn_weights* w, const int64_t* x, const int64_t* label, int64_t B,
                                    float* logits, int32_t* err_flag, void* workspace, size_t workspace_bytes,
                                    dvq_stream_t stream) {
    DVQ_REQUIRE(x && logits, "pixelcnn_forward: null pointer");
    return run(w, label, nullptr, x, B, nullptr, logits, err_flag, workspace, workspace_bytes, (hipStream_t)stream);
}
