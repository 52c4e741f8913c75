/*
 * dvq.h -- C ABI of libdvq_hip.so: the MI355X (gfx950) implementation of the D-VQVAE batched
 * grasp-generation hot path (reference: network/gen_net.py:78-125, `GenNet.gen`).
 *
 * The reference is pure Python/PyTorch and has NO FFI/plugin layer: its "operator
 * interface" for this path is the set of nn.Module methods cited on each entry point below.  These
 * entry points are what a ctypes binding added to the reference would call (INTEGRATION.md shows the
 * stub); the host-side mirror in d-vqvae_amd/network/ is exactly such a binding.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP global memory) unless the name ends in _host;
 *   - tensors are dense row-major fp32 / int64 exactly as the reference holds them;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises,
 *     nothing allocates: scratch comes from the caller (`*_workspace_bytes` tells how much);
 *   - return value: 0 = DVQ_OK, otherwise a dvq_status; dvq_last_error() gives the text
 *     (thread-local).  Invalid shapes/alignments are rejected (never silently clamped).
 *   - out-of-range code indices (>= K) are reported through a device-side error flag that the host
 *     mirror turns into the reference's RuntimeError (quantizer.py:72 scatter_ bounds).
 */
#ifndef DVQ_H
#define DVQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dvq_stream_t; /* hipStream_t */

typedef enum {
    DVQ_OK = 0,
    DVQ_EINVAL = 1,   /* bad shape / alignment / null pointer */
    DVQ_EWORKSPACE = 2, /* workspace too small */
    DVQ_ELAUNCH = 3,  /* HIP launch error */
    DVQ_ENODEVICE = 4
} dvq_status;

/* The version of this header.  dvq_abi_version() returns the library's: a binding checks the two for equality at load time
 * (struct layouts change between versions). */
#define DVQ_ABI_VERSION 9
int dvq_abi_version(void);
const char* dvq_last_error(void);
/* number of visible HIP devices, or -1; does not create a context */
int dvq_device_count(void);
/* The library reads its environment knobs (tile/chunk choices and the PointNet test switches DVQ_PN_FILTER / _EXHAUSTIVE /
 * _CAPS: none changes a result) ONCE, at first use; this re-reads them (tests that flip a knob in-process call it). */
int dvq_reload_env(void);

/* ------------------------------------------------------------------ generic dense layer (MFMA fp32)
 * y[M,N] = act( sum_s x_s[M,K_s] @ w_s[N,K_s]^T + bias[N] ) -- nn.Linear / 1x1 conv / conv taps.
 * Replaces: Decoder.forward (network/DVQVAE.py:183-185), STN3d fc1..fc3 (pointnet_encoder.py:35-37),
 * Encoder.forward (DVQVAE.py:161-166).  K_s % 32 == 0, 16-byte aligned rows.
 * Arithmetic: fp32 operands on the 16-bit matrix cores, fp32 accumulation, fp32-GEMM-class accuracy (<= 4e-6 * sum|x w| against
 * fp64, measured ~1e-7).  What runs follows the weight image handed over in `wp`:
 *   DVQ_PLANES_F16X2  (dvq_split_f16x2; what the host mirror packs by default): w * 2^t[n] (one power of two per output row) as
 *       two fp16 planes, the second holding the remainder * 2^11; activations are split the same way while they are staged;
 *       THREE products per fp32 product in two accumulators: out = (x1 w1 + (x1 w2 + x2 w1) * 2^-11) * 2^-t[n].  |x| < 65 520
 *       (fp16 range): a row beyond it comes out NaN, never silently wrong.
 *   DVQ_PLANES_BF16X3 (dvq_split_bf16x3), or no image at all (split on the fly): every operand split EXACTLY into three bf16
 *       pieces, the six partial products of weight >= 2^-24: fp32's range, twice the matrix work.
 * DVQ_GEMM=fp32 in the environment ignores the images and runs v_mfma_f32_32x32x2_f32 (exact fp32 chain). */
#define DVQ_PLANES_BF16X3 0
#define DVQ_PLANES_F16X2 1
typedef struct {
    const float* x; /* [M, K] row stride ldx */
    const float* w; /* [N, K] row stride ldw */
    int64_t ldx, ldw;
    int32_t K;
    int32_t wp_kind;    /* DVQ_PLANES_* of `wp` */
    const uint16_t* wp; /* optional: w pre-split into planes [3 or 2][N][ldw] */
    int64_t wp_plane;   /* elements between planes */
    const float* w_scale; /* DVQ_PLANES_F16X2: [N] row scales 2^-t[n] (dvq_split_f16x2); ONE array for all sources of a call */
} dvq_gemm_src;

#define DVQ_MAX_SRC 8
#define DVQ_ACT_NONE 0
#define DVQ_ACT_RELU 1

int dvq_linear(const dvq_gemm_src* src_host, int nsrc, int64_t M, int N, const float* bias,
               int act, float* y, int64_t ldy, dvq_stream_t stream);
/* Decoder.forward / Encoder.forward (network/DVQVAE.py:161-166, 183-185): Linear + ReLU, Linear + ReLU, Linear as one entry
 * point (SURVEY.md 8b `mlp3`): 2560 -> 1024 -> 256 -> 55, 2048 -> 1024 -> 128 -> 6, 1024 -> 1024 -> 512 -> 256.  Weights [n_out, k_in]
 * dense row-major (nn.Linear's layout), `wp` optional pre-split planes (dvq_split_bf16x3); hidden sizes % 32 == 0. */
typedef struct {
    const float* w;      /* [n_out, k_in] */
    const float* b;      /* [n_out] or NULL */
    const uint16_t* wp;  /* optional planes [3 or 2][n_out][k_in] */
    int32_t n_out, k_in;
    const float* w_scale; /* DVQ_PLANES_F16X2: [n_out] */
    int32_t wp_kind;     /* DVQ_PLANES_* */
    int32_t _pad;
} dvq_mlp_layer;
size_t dvq_mlp3_workspace_bytes(int64_t M, int n0, int n1);
int dvq_mlp3(const float* x, int64_t ldx, int64_t M, const dvq_mlp_layer* layers_host /* [3] */, float* y, int64_t ldy,
             void* workspace, size_t workspace_bytes, dvq_stream_t stream);
/* planes[p][i] (p = 0,1,2; bf16 bit patterns) with w[i] == planes[0][i] + planes[1][i] + planes[2][i] exactly */
int dvq_split_bf16x3(const float* w, int64_t n, uint16_t* planes, dvq_stream_t stream);
/* The fp16 image of a weight tensor w [outer][N][K] (dense; outer = conv taps, 1 for nn.Linear), N = output rows:
 *   dvq_f16x2_row_absmax: row_absmax[n] = max(row_absmax[n], max_{o,k} |w[o][n][k]|)  -- the caller zeroes row_absmax [N] and
 *       calls this once per tensor whose products are summed into the same outputs (all taps of a conv; horiz_stack and
 *       vert_to_horiz of a gated layer, models.py:76-81);
 *   dvq_split_f16x2: t[n] = largest power of two with row_absmax[n] * 2^t < 2^15; planes[0] = fp16(w * 2^t[n]),
 *       planes[1] = fp16((w * 2^t[n] - planes[0]) * 2^11), both [outer][N][K]; row_scale[n] = 2^-t[n]. */
int dvq_f16x2_row_absmax(const float* w, int64_t outer, int N, int K, float* row_absmax, dvq_stream_t stream);
int dvq_split_f16x2(const float* w, int64_t outer, int N, int K, const float* row_absmax, uint16_t* planes /* [2][outer][N][K] */,
                    float* row_scale /* [N] */, dvq_stream_t stream);

/* ------------------------------------------------------------------ VQ codebook nearest neighbour
 * VectorQuantizer.forward(z, istrain=False), network/vqvae/quantizer.py:46-49:
 *   d[m,k] = (sum_j z[m,j]^2 + sum_j E[k,j]^2) - 2 * sum_j z[m,j] E[k,j];  idx[m] = argmin_k d[m,k]
 * evaluated in fp32 in that association, every sum a k-ordered fmaf chain (the "canonical order",
 * oracle/vq_canonical.c); first minimum wins, a NaN distance wins over everything (torch.argmin).
 * dvq_vq_argmin is the exact fp32 kernel for any K and D % 32 == 0 (fp32 MFMA, canonical order).
 * workspace: dvq_vq_argmin_workspace_bytes(M, K). dmin (optional, [M]) receives the winning distance. */
size_t dvq_vq_argmin_workspace_bytes(int64_t M, int K);
int dvq_vq_argmin(const float* z, int64_t ldz, const float* E, int64_t M, int K, int D,
                  int64_t* idx, float* dmin, void* workspace, size_t workspace_bytes,
                  dvq_stream_t stream);

/* Fast path for the headline shape (K = 512, D = 256, dense z) and, since round 6, for codebooks of K = 32, 64 ... 480 entries at
 * D = 256 (the model's six K = 128 codebooks: the image is padded to 512 entries that never win): returns the SAME indices as
 * dvq_vq_argmin, bit for bit.  One persistent kernel (one workgroup per CU, the fp16 codebook image held in registers): z is read from
 * HBM once and streamed under an fp16-MFMA filter with a proven error bound that keeps every row's candidate entries (the
 * exact fp32 argmin is always among them); rows with one candidate are decided, the rest are re-evaluated in the
 * canonical fp32 order inside the same workgroup (DESIGN.md "vq_argmin").  `packed` is the codebook image built once
 * per codebook by dvq_vq_pack (fp16 image of -2 sE E in MFMA-fragment order, canonical |e_k|^2, max |e_k|, measured
 * rounding error).  Rows are not rescaled: fp16 overflow (|z_j| > 65504), NaN/Inf rows and codebooks with magnitudes
 * outside 2^+-40 take the all-entries path.
 * `slow_rows` (device, may be NULL, never reset by the library): the kernel adds the number of rows that cost far more
 * than a filtered row: all-entries scans and rows with 32 or more candidate pairs.  A caller that sees a large fraction
 * there (ill-conditioned input: |z| >> codebook spread) should use dvq_vq_argmin instead. */
int dvq_vq_fast_supported(int K, int D);
size_t dvq_vq_pack_bytes(int K, int D);
int dvq_vq_pack(const float* E, int K, int D, void* packed, size_t packed_bytes, dvq_stream_t stream);
size_t dvq_vq_fast_workspace_bytes(int64_t M, int K, int D);
int dvq_vq_argmin_fast(const float* z, const float* E, const void* packed, int64_t M, int K, int D,
                       int64_t* idx, unsigned long long* slow_rows, void* workspace, size_t workspace_bytes,
                       dvq_stream_t stream);

/* VectorQuantizer.get_emb / one-hot @ E (quantizer.py:50-53,68-75): out[m, :] = E[idx[m], :].
 * *err_flag (device int32, caller zeroes it) is set to 1 if any idx is outside [0,K). */
int dvq_vq_lookup(const float* E, const int64_t* idx, int64_t idx_stride, int64_t M, int K, int D,
                  float* out, int64_t ldo, int32_t* err_flag, dvq_stream_t stream);

/* ------------------------------------------------------------------ PointNet encoder
 * PointNetEncoder.forward (global_feat=True, feature_transform=False), pointnet_encoder.py:140-169
 * with STN3d.forward :27-45.  BatchNorm (eval) is folded into the preceding conv/fc by the host
 * packer in fp64; fc3's bias has the 3x3 identity folded in. */
typedef struct {
    int32_t C; /* 3 or 4 input channels */
    int32_t _pad;
    /* STN trunk */
    const float *s_w1, *s_b1; /* [64,4]  (C padded to 4 with zeros) */
    const float *s_w2, *s_b2; /* [128,64] */
    const float *s_w3, *s_b3; /* [1024,128] */
    const float *s_f1, *s_c1; /* [512,1024] */
    const float *s_f2, *s_c2; /* [256,512] */
    const float *s_f3, *s_c3; /* [9,256], bias + identity */
    /* main trunk */
    const float *w1, *b1;     /* [64,4] */
    const float *w2, *b2;     /* [128,64] */
    const float *w3, *b3;     /* [1024,128] */
    /* optional split-bf16 planes ([3][out][in], dvq_split_bf16x3) of the GEMM weights; all-or-nothing */
    const uint16_t *s_w2p, *s_w3p, *s_f1p, *s_f2p, *s_f3p, *w2p, *w3p;
    /* optional filter images of a trunk's conv2 / conv3 weights (dvq_pointnet_pack_filter); with them (and the planes) the trunk
     * runs conv3 + max as an fp16 matrix-core filter followed by an exact fp32 re-evaluation of the candidate points */
    const void *s_w3f, *w3f;
} dvq_pointnet_weights;

/* Run-time consistency counters of the filtered trunk on the current device (v8): out[0] = tile records the trunk kernel marked
 * suspect (an input of its merge was stale), out[1] = channels whose exact maximum lay outside the interval their tile records
 * promised.  Both kinds are re-evaluated over all the points concerned (the features stay right); a non-zero count means the
 * filter's bookkeeping failed and should be reported.  Synchronises the device.  reset != 0: zero them after reading. */
int dvq_pointnet_fault_counters(uint64_t* out /* [2] */, int reset);
/* Filter image of one trunk (device pointers; dvq_pointnet_filter_bytes() bytes): conv3's weights w3 [1024,128] as fp16 rows scaled by
 * a per-row power of two in the k order the trunk kernel consumes, the inverse scales and the row norms; and (v8) conv2's weights
 * w2 [128,64] (BatchNorm folded) as the two fp16 planes of the three-product split of dvq_split_f16x2 with their row scales:
 * the filtered trunk multiplies conv2 on them (six bf16 products per fp32 product before). */
size_t dvq_pointnet_filter_bytes(void);
int dvq_pointnet_pack_filter(const float* w2, const float* w3, void* image, dvq_stream_t stream);

size_t dvq_pointnet_workspace_bytes(int64_t B, int N);
/* pc [B,C,N] (channel-major per sample, as the datasets emit it) -> feat [B,1024], trans [B,3,3] */
int dvq_pointnet_encode(const dvq_pointnet_weights* w_host, const float* pc, int64_t B, int N,
                        float* feat, int64_t ld_feat, float* trans /* optional */,
                        void* workspace, size_t workspace_bytes, dvq_stream_t stream);

/* ------------------------------------------------------------------ gated PixelCNN prior sampler
 * GatedPixelCNN.generate (network/pixelcnn/models.py:176-198) on the 3x3 latent grid, as an
 * incremental (cached) sampler: the network is exactly causal, so each grid position is evaluated
 * once (1/9 of the reference's FLOPs, identical math).  The draw at each position is
 * argmax_k softmax(logits)_k / q_k with q ~ Exp(1) supplied by the caller -- the exponential race
 * torch.multinomial(1) evaluates (models.py:195).  Packed layout: d-vqvae_amd/packing.py. */
typedef struct {
    const float* wv;   /* vertical taps   [n_vtaps][2*dim (gate-packed)][dim]            */
    const float* bv;   /* [2*dim] gate-packed                                              */
    const float* wh;   /* horizontal taps [n_htaps][2*dim (gate-packed)][dim]              */
    const float* wv2h; /* [2*dim (gate-packed)][2*dim (gate-packed input order)]           */
    const float* bh;   /* horiz_stack.bias + vert_to_horiz.bias, gate-packed               */
    const float* cls;  /* class_cond_embedding [n_classes][2*dim] gate-packed              */
    const float* wr;   /* horiz_resid [dim][dim]                                           */
    const float* br;   /* [dim]                                                            */
    /* optional weight images of the whole tensors (kind: dvq_pixelcnn_weights.planes_kind): [3 or 2][n_taps][2*dim][dim] for wv / wh */
    const uint16_t *wv_p, *wh_p, *wv2h_p, *wr_p;
    /* DVQ_PLANES_F16X2: row scales of the three GEMMs of a layer: vertical [2*dim], horizontal [2*dim] (wh and wv2h are summed
     * into the same outputs: their images share one absmax), residual [dim] */
    const float *sv, *sh, *sr;
} dvq_pixelcnn_layer;

typedef struct {
    int32_t n_layers, dim, n_in /* tokens */, n_classes, n_hidden /* 2048 */;
    int32_t planes_kind;        /* DVQ_PLANES_* of every *_p image below and in the layers */
    const float* tok_emb;       /* embedding.weight [n_in][dim] */
    const dvq_pixelcnn_layer* layers_host; /* host array [n_layers]; layer 0: mask A applied, k=5 */
    const float *w0, *b0;       /* output_conv.0 [n_hidden][dim] */
    const float *w2, *b2;       /* output_conv.2 [n_in][n_hidden] */
    const uint16_t *w0_p, *w2_p; /* optional weight images */
    const float *s0, *s2;        /* DVQ_PLANES_F16X2: their row scales [n_hidden], [n_in] */
    /* optional (DVQ_PLANES_F16X2 images only): the class tables of dvq_pixelcnn_build_tables for EXACTLY these weights.  With them
     * every call -- B = 1 included -- reads what depends on the class label only (grid row 0's vertical stack, position (0, 0),
     * the accumulator states of the sources that follow from them) instead of computing it; without them calls of at least two
     * rows per class build them per call in the workspace.  The results are the same bits either way. */
    const void* class_tables;
} dvq_pixelcnn_weights;

/* Class tables of a packed prior (weights only: build once per model).  dvq_pixelcnn_tables_bytes() = 0 when the weight images
 * are not DVQ_PLANES_F16X2 or the process runs DVQ_GEMM=fp32 (the tables are read by the fp16-plane kernels). */
size_t dvq_pixelcnn_tables_bytes(const dvq_pixelcnn_weights* w_host);
int dvq_pixelcnn_build_tables(const dvq_pixelcnn_weights* w_host, void* tables, size_t tables_bytes, dvq_stream_t stream);

size_t dvq_pixelcnn_workspace_bytes(const dvq_pixelcnn_weights* w_host, int64_t B);
/* label [B] int64 in [0,n_classes), noise q [B,9,n_in] -> codes [B,9] int64 (raster order).
 * logits_out (optional) [B,9,n_in] receives the logits each draw was made from.
 * *err_flag (device int32, caller zeroes it): bit 0 = a label / token out of range; bit 2 = a draw from logits that were all NaN
 * (that position's entry of `codes` is -1 then (v8; 0 before), the sampler itself continues from token 0): with fp16 weight
 * images, an activation beyond +-65 504 made the row NaN -- run the rows that hold a -1 again with bf16x3 images. */
int dvq_pixelcnn_sample(const dvq_pixelcnn_weights* w_host, const int64_t* label, const float* noise,
                        int64_t B, int64_t* codes, float* logits_out, int32_t* err_flag,
                        void* workspace, size_t workspace_bytes, dvq_stream_t stream);
/* GatedPixelCNN.forward (models.py:161-174) for given tokens x [B,9] (raster order): logits [B,9,n_in]
 * (position-major; the host mirror permutes to the reference's [B,n_in,3,3]) */
int dvq_pixelcnn_forward(const dvq_pixelcnn_weights* w_host, const int64_t* x, const int64_t* label,
                         int64_t B, float* logits /* [B,9,n_in] */, int32_t* err_flag,
                         void* workspace, size_t workspace_bytes, dvq_stream_t stream);

/* ------------------------------------------------------------------ MANO layer (third-party `mano`
 * package at network/gen_net.py:116-118 and gen_diverse_grasp_obman.py:252-253; restated, unpinned)
 * use_pca=True, 45 comps, flat_hand_mean folded into pose_mean by the packer. */
typedef struct {
    const float* v_template;  /* [778,3] = the blendshape GEMM's bias [2334] */
    const float* blend_w;     /* [2334][160]: row e = [shapedirs[.,e] (10) | posedirs[.,e] (135) | 0 (15)]: V = X . blend_w^T */
    const uint16_t* blend_w_planes;   /* optional weight image of blend_w, [3 or 2][2334][160] (planes_kind) */
    const float* j_template;  /* [16,3]   J_regressor @ v_template */
    const float* j_shapedirs; /* [10][48] J_regressor @ shapedirs */
    const float* weights;     /* [778,16] */
    const float* comps;       /* [45,45] hands_components */
    const float* pose_mean;   /* [48] */
    int32_t parents[16];
    const float* blend_w_scale;       /* DVQ_PLANES_F16X2: [2334] row scales */
    int32_t planes_kind;              /* DVQ_PLANES_* of blend_w_planes */
    int32_t _pad;
} dvq_mano_model;

/* betas [B,10] (row stride ldb), pose [B,45] (ldp), optional global_orient [B,3] (ldg) and transl
 * [B,3] (ldt) -> verts; layout 0: [B,778,3] (the mano layer's), 1: [B,3,778] (PointNet input).
 * Three launches per 16 384 samples: pose/chain kernel, blendshape GEMM [B,160] x [160,2334], skinning kernel.
 * workspace: dvq_mano_workspace_bytes(B). */
size_t dvq_mano_workspace_bytes(int64_t B);
int dvq_mano_forward(const dvq_mano_model* m_host, const float* betas, int64_t ldb, const float* pose,
                     int64_t ldp, const float* global_orient, int64_t ldg, const float* transl,
                     int64_t ldt, int64_t B, float* verts, int layout, float* joints /* optional [B,16,3] */,
                     void* workspace, size_t workspace_bytes, dvq_stream_t stream);

/* ------------------------------------------------------------------ small data movement
 * out[m, col0:col0+W] = src[m, 0:W]  (concatenations of gen_net.py:109,121) */
int dvq_copy_cols(const float* src, int64_t lds, int64_t M, int W, float* out, int64_t ldo,
                  dvq_stream_t stream);
/* 61-parameter assembly, gen_diverse_grasp_obman.py:243-247:
 * [betas(10) | global_orient(3)=pos[:, :3] | pca_pose(45) | transl(3)=pos[:, 3:6]] */
int dvq_assemble61(const float* recon /* [B,55] */, const float* recon_pos /* [B,6] */, int64_t B,
                   float* out /* [B,61] */, dvq_stream_t stream);
/* per-grasp random object rotation pre-step, gen_diverse_grasp_ho3d.py:213-230:
 * out[b,:3,:] = R[b] @ pc[:3,:] + t ; extra channels copied */
int dvq_transform_cloud(const float* pc /* [C,N] or [B,C,N] */, int64_t pc_batch_stride, const float* R /* [B,3,3] */,
                        const float* t /* [3] */, int64_t B, int C, int N, float* out /* [B,C,N] */,
                        dvq_stream_t stream);

/* ------------------------------------------------------------------ sampling noise of the prior
 * GatedPixelCNN.generate draws with probs.multinomial(1) (network/pixelcnn/models.py:190-197), i.e. argmax_k p_k / q_k with
 * q ~ Exp(1).  out[r, c] = -log(u) from Philox4x32-10 keyed by `seed`, counter (c / 4, row0 + r, stream_id): the noise of a
 * grasp depends on its GLOBAL row only, so a batch sharded over ranks (row0 = first row of the shard) draws exactly what the
 * unsharded batch draws.  `stream_id` separates independent uses (objects, calls).  cols % 4 == 0. */
int dvq_exp1_noise(uint64_t seed, uint32_t stream_id, int64_t row0, int64_t rows, int cols, float* out /* [rows,cols] */,
                   dvq_stream_t stream);
/* The same draws in another row order: out[r, :] = the noise of global row row0 + perm[r] (perm: device int64 [rows], values in
 * [0, rows)).  GenNet.gen evaluates the prior in the order of the object codes; drawing the noise in that order replaces a
 * gather of the whole [B, 9 * 512] tensor. */
int dvq_exp1_noise_rows(uint64_t seed, uint32_t stream_id, int64_t row0, const int64_t* perm, int64_t rows, int cols,
                        float* out /* [rows,cols] */, dvq_stream_t stream);

/* Self-test: out[0] (device) = an fp16 MFMA product with a SUBNORMAL input, out[1] = its exact value.  The fast VQ kernel's
 * error bound assumes the matrix core keeps fp16 subnormals (measured so on gfx950); tests assert out[0] == out[1]. */
int dvq_probe_f16_subnormal(float* out /* device [2] */, dvq_stream_t stream);

/* ------------------------------------------------------------------ contact / penetration proxies (after the path)
 * utils/utils_loss.py:7-24 get_NN (pytorch3d knn_points, K=1): nearest target point of every source point of the same
 * batch element: squared distance d = fma(dz,dz, fma(dy,dy, dx*dx)) and index (first minimum; NaN first).
 * Strides are in floats, so [B,N,3] tensors and [B,C,N] channel-first clouds are both read in place.  N2 <= 4096. */
int dvq_nn_points(const float* src, int64_t src_batch_stride, int64_t src_point_stride, int64_t src_coord_stride,
                  const float* trg, int64_t trg_batch_stride, int64_t trg_point_stride, int64_t trg_coord_stride,
                  int64_t B, int N1, int N2, float* dist /* [B,N1] */, int64_t* idx /* [B,N1] */, dvq_stream_t stream);
/* Area-weighted vertex normals (utils/loss.py:156-157: Meshes(...).verts_normals_packed()) of B meshes sharing one
 * topology: faces [F,3] int32; vf_off [V+1], vf_face [3F]: the faces incident to each vertex, ascending (CSR). */
int dvq_vertex_normals(const float* verts /* [B,V,3] */, int64_t B, int V, const int32_t* faces, const int32_t* vf_off,
                       const int32_t* vf_face, float* normals /* [B,V,3] */, dvq_stream_t stream);
/* utils/utils_loss.py:27-45 get_interior: interior[b,p] = (hand[b,nn[b,p]] - obj[b,p]) . normals[b,nn[b,p]] > 0 */
int dvq_interior(const float* normals /* [B,V,3] */, const float* hand /* [B,V,3] */, int V, const float* obj,
                 int64_t obj_batch_stride, int64_t obj_point_stride, int64_t obj_coord_stride,
                 const int64_t* nn_idx /* [B,N] */, int64_t B, int N, uint8_t* interior /* [B,N] */, dvq_stream_t stream);

/* ------------------------------------------------------------------ all-gather of the generated MANO parameters (multi-GPU)
 * The batch of objects shards contiguously over R ranks (one process per GPU); the only exchange of the path is
 * the all-gather of the [B/R, 61] parameter rows (61-parameter assembly, gen_diverse_grasp_obman.py:243-247) into [B, 61],
 * rank-major: RCCL over xGMI.  The reference has no distributed code (nothing to replace); the op is new.
 *   dvq_comm_unique_id : rank 0 makes the 128-byte id; the caller hands it to every rank (any side channel)
 *   dvq_comm_init      : every rank, on its own device (hipSetDevice first): the communicator
 *   dvq_allgather_params: out[r * rows_per_rank + i, :] = rank r's local[i, :]; enqueued on `stream`; equal shards only
 *                        (ragged batches: pad the shard, the host mirror does)
 * RCCL is resolved at the first call (dlopen); without it these return DVQ_ENODEVICE and the rest of the library is unaffected. */
/* 1 when librccl and the six entry points used here resolve in this process, else 0: a probe that creates nothing (v8; making a
 * unique id starts RCCL's bootstrap thread and socket, which only rank 0 should do) */
int dvq_comm_available(void);
int dvq_comm_unique_id(void* id_out, size_t id_bytes /* >= 128 */);
int dvq_comm_init(const void* id, size_t id_bytes, int world, int rank, void** comm_out);
int dvq_allgather_params(void* comm, const float* local /* [rows_per_rank, cols] */, int64_t rows_per_rank, int cols /* 61 */,
                         float* out /* [world * rows_per_rank, cols] */, dvq_stream_t stream);
/* ranks of the communicator as RCCL reports them (ncclCommCount; v9): lets a caller show that the collective really spans N processes */
int dvq_comm_count(void* comm, int* ranks_out);
int dvq_comm_destroy(void* comm);

/* ------------------------------------------------------------------ optional per-launch timing
 * When enabled, every kernel launch of the library is bracketed by two HIP events on its stream.
 * dvq_prof_read waits for the recorded events and returns per-kernel-kind totals (bench.py's roofline leg). */
typedef struct {
    char name[32];
    int64_t count;
    double ms;     /* summed launch durations */
    double flops;  /* summed algorithmic FLOPs  */
    double bytes;  /* summed algorithmic bytes  */
} dvq_prof_entry;
int dvq_prof_enable(int on);
int dvq_prof_reset(void);
int dvq_prof_read(dvq_prof_entry* out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif /* DVQ_H */
