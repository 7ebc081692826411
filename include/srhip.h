/* libsrhip -- C ABI of the MI355X (gfx950) hot path of situation-recognition.
 *
 * The reference (vFones/situation-recognition) has NO FFI: its hot path is Python
 * calling torch/torchvision/cuDNN ops.  This ABI is therefore the set
 * of native entry points that sit beneath the reference's Python class API
 * (model.resnet / model.GGSNN / model.FCGGNN); each entry cites the reference
 * lines whose implicit native kernels it replaces.  A reference maintainer binds
 * it with ctypes.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory + sizes; no torch types; never allocates,
 *     never synchronises, never throws; work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = default stream);
 *   - returns SR_OK (0) or a negative SR_ERR_* code (arguments are validated on
 *     the host BEFORE anything is launched);
 *   - `dtype`: SR_F32 (0) or SR_BF16 (1) = storage + MFMA input type of the
 *     activations/weights; accumulation is always fp32;
 *   - activations are row-major [rows, channels] = NHWC for images;
 *   - thread-safe per device (no global mutable state besides a read-only zero page).
 */
#ifndef SRHIP_H
#define SRHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_ERR_ARG (-1)      /* bad shape / alignment / null pointer */
#define SR_ERR_DTYPE (-2)    /* unsupported dtype code */
#define SR_ERR_LAUNCH (-3)   /* hipLaunch reported an error */
#define SR_ERR_UNSUPPORTED (-4)

#define SR_F32 0
#define SR_BF16 1

/* epilogue activation codes for sr_gemm / sr_conv2d */
#define SR_ACT_NONE 0
#define SR_ACT_RELU 1
#define SR_ACT_SIGMOID 2
#define SR_ACT_TANH 3
#define SR_ACT_SIGMOID_MUL 4 /* C = s = sigmoid(v); C2 = s * aux1              (GRU reset: r, r*h)   */
#define SR_ACT_TANH_BLEND 5  /* C2 = c = tanh(v); C = (1-aux2)*aux1 + aux2*c  (GRU candidate+blend) */

/* Bumped on every incompatible change of a signature or struct below (2: sr_conv_args grew in_scale / in_shift, sr_bn_finalize*
 * gained the twin buffers; 3: sr_conv_route, sr_comm_* / sr_allreduce_sum; 4: sr_node_init_bwd's scratch; 5: sr_set_cu_share is per
 * calling thread, SR_ROUTE_C3D128 names the channel-slice kernel, SR_ROUTE_C3D256, sr_conv_pair*).  A binding must refuse a library
 * whose version differs. */
#define SR_ABI_VERSION 5
int sr_abi_version(void);

/* Launches that follow FROM THE CALLING THREAD size their persistent grids (and everything derived from the CU count: tile shapes,
 * partial-statistics row counts returned by sr_conv_stats_rows / sr_gemm_stats_tiles) for 1/share of the device's compute units
 * (share 1..8; default 1, or the SR_CU_SHARE environment variable): `share` streams that run the same kind of work side by side -- the two
 * backbones of FCGGNN.forward, reference model.py:159,116 -- then co-reside on disjoint CUs instead of each launching a
 * full-chip grid that queues behind the other's.  Returns the previous share, or SR_ERR_ARG. */
int sr_set_cu_share(int share);

/* One (activation, weight) operand pair of a GEMM: contributes A[M,K] . W[N,K]^T. */
typedef struct sr_kpair {
  const void* A; /* [M, K] row-major, row stride lda (elements)          */
  const void* W; /* [N, K] row-major (nn.Linear layout), row stride ldw  */
  int64_t lda, ldw;
  int32_t K;     /* multiple of 64 (bf16) / 32 (f32)                      */
  int32_t _pad;
} sr_kpair;

/* C[M,N] = epilogue( sum_p A_p . W_p^T + bias_scale*bias + bias2 ) (+ res)
 *
 * Replaces the nn.Linear calls of GGSNN.forward (reference model.py:64,75,80-83),
 * of the classifiers (model.py:152,168) and, in backward, their autograd GEMMs.
 * Up to 3 operand pairs are summed into one accumulator (W_z n + U_z h in one launch).
 * out_f32 != 0: C / C2 / res / aux are fp32 even when dtype is bf16.
 * stats (optional): fp32 [rows][2][N] partial column sums / sums of squares of (acc + bias) for train-mode
 * BatchNorm, rows = sr_gemm_stats_tiles(M, N).  Every row is written in full (zeros where a workgroup had
 * nothing left to add); their sum over rows is the batch sum, which is all sr_bn_finalize uses.  A row holds what one
 * wave group of a workgroup accumulated over up to 32 of its row tiles.
 */
typedef struct sr_gemm_args {
  sr_kpair kp[3];
  int32_t npairs, M, N, act;
  void* C;  int64_t ldc;
  void* C2;                       /* second output (same ld) for the GRU epilogues */
  const float* bias; const float* bias2; float bias_scale; int32_t out_f32;
  const void* res; int64_t ldres; /* added BEFORE the activation (residual); same type as C */
  const void* aux1; const void* aux2; /* same type / ld as C */
  float* stats;
} sr_gemm_args;
int sr_gemm(const sr_gemm_args* a, int dtype, void* stream);
int sr_gemm_stats_tiles(int M, int N);   /* rows of `stats` sr_gemm / sr_conv2d will write for (M, N) */
/* Which tile shape sr_gemm / sr_conv2d will run an (M, N) launch on: 4 = 256x256 (8 waves, half-step ping-pong loop),
 * 2 = 256x128, 1 = 256x64 (4 waves, two workgroups per CU), 0 = the 128-byte-step fallback kernels.  `linear` != 0: epilogue
 * is SR_ACT_NONE / SR_ACT_RELU; the fused GRU epilogues (`linear` == 0) run on 4, or on 2 when the 256x256 tiles of the launch
 * would cover at most half of the CUs.  Introspection for tests and profiles (a parity test states which kernel it covered). */
int sr_gemm_tile_cfg(int M, int N, int linear, int out_16bit);
/* Diagnostic only (synchronises!): copies the in-kernel cycle stamps of the last v3 GEMM launched with
 * SR_GEMM_DEBUG=4 to host memory: [256 blocks][8 waves][8] uint64 (0 vmcnt wait, 1 barrier, 2 DMA issue, 3 MFMA, 4 epilogue, 5 steps). */
int sr_debug_stamps(unsigned long long* host_out, int count);

/* NHWC convolution as implicit GEMM on the same MFMA kernel:
 * y[b,ho,wo,co] = epilogue( sum_{r,q,c} x[b, ho*s-p+r, wo*s-p+q, c] * w[co,r,q,c] + bias[co] ) (+ res)
 * Replaces the cuDNN/MIOpen conv2d calls inside torchvision's ResNet forward
 * (call site: reference model.py:35).  w is [Cout][KH][KW][Cin]; Cin % 64 == 0
 * (bf16) / 32 (f32).  stem != 0 selects the 7x7/2 stem form: x is the padded
 * 4-channel image written by sr_stem_prep and w is [Cout][8][32] (sr_stem_pack_weight layout).
 */
typedef struct sr_conv_args {
  const void* x; const void* w;
  int32_t B, H, W, Cin, Cout, KH, KW, stride, pad, stem;
  void* y;                 /* [B*Ho*Wo, Cout] */
  const float* bias;       /* folded BN shift (eval mode) or NULL */
  const void* res;         /* residual, same shape/type as y, or NULL */
  int32_t act; int32_t no_store; /* no_store != 0: statistics-only launch, y is not written (Cout > 128 only) */
  float* stats;            /* see sr_gemm_args.stats */
  const float* escale;     /* optional per-output-channel multiplier: y = act(acc*escale + bias (+res)) (Cout > 128 only).
                              With no_store it gives train-mode BatchNorm in two conv launches and no elementwise pass:
                              launch 1 (no_store) -> statistics -> sr_bn_finalize -> launch 2 with escale=scale, bias=shift. */
  const float* in_scale;   /* optional INPUT affine [Cin] (both or neither): the convolution runs on relu(x*in_scale + in_shift), */
  const float* in_shift;   /* i.e. the preceding layer's BatchNorm + ReLU is applied to x on its way into the matrix cores and the
                              normalised tensor is never written (torchvision bottleneck: bn2 -> relu -> conv3, call site reference
                              model.py:35).  Only the launches sr_conv_in_affine_supported() accepts; SR_ERR_UNSUPPORTED otherwise. */
} sr_conv_args;
int sr_conv2d(const sr_conv_args* a, int dtype, void* stream);
/* 1 if sr_conv2d serves this launch (geometry, act, res, stats, no_store as they will be passed) WITH in_scale / in_shift, else 0:
 * bf16 1x1 / stride 1 expansion convolutions (Cin 64 / 128 / 256, Cout 256 / 512 / 1024, >= 32768 output rows) with a residual and
 * ReLU, and the 64-channel 3x3 / stride 1 layer on 56-wide images in its train-mode form (statistics, no bias, no activation).
 * `res` and `stats` are only tested against NULL; no pointer is read. */
int sr_conv_in_affine_supported(const sr_conv_args* a, int dtype);
/* Which kernel sr_conv2d WOULD run this exact launch on (a dry run of its dispatch: same argument checks, nothing is launched,
 * no pointer is dereferenced -- x / w / y must still be non-null and 16-byte aligned).  Returns a negative SR_ERR_* code the launch
 * itself would return, or: 0 / 1 / 2 / 4 = the generic implicit-GEMM kernel on the tile shape sr_gemm_tile_cfg reports;
 * SR_ROUTE_WS = conv1x1_ws_kernel (weight-stationary output-heavy 1x1),
 * SR_ROUTE_C3D = conv3x3_c64_kernel (direct 3x3, 64 channels), SR_ROUTE_C3D128 = conv3x3_slices_kernel<KsL2> (direct 3x3, 128 channels, 28 x 28 images),
 * SR_ROUTE_STEM = stem_conv_kernel (direct 7x7/2 stem).
 * Introspection for the parity tests: a test at a reduced batch asserts that it covered the kernel the benchmark batch runs. */
#define SR_ROUTE_WS 16
#define SR_ROUTE_C3D 18
#define SR_ROUTE_STEM 19
#define SR_ROUTE_C3D128 20
#define SR_ROUTE_C3D256 21 /* conv3x3_slices_kernel<KsL3> (direct 3x3, 256 channels, 14 x 14 images) */
int sr_conv_route(const sr_conv_args* a, int dtype);
/* rows of `stats` this exact launch writes (the stem runs on a direct-convolution kernel with its own partial layout; every
 * other launch follows sr_gemm_stats_tiles(M, Cout)).  Fill the geometry fields of `a`; pointers are not read. */
int sr_conv_stats_rows(const sr_conv_args* a, int dtype);

/* ---- A bottleneck's expansion conv FUSED with the NEXT block's reduce conv (train mode, bf16; torchvision chain
 * conv3 -> bn3 -> (+identity) -> relu -> next.conv1, call site reference model.py:35):
 *     z[M, Cexp] = relu( (relu(x*in_scale + in_shift) . w_exp[Cexp, Cmid]^T) * escale + eshift + res )      the block's output
 *     y[M, Cred] = z . w_red[Cred, Cexp]^T                                                                   the next block's conv1, RAW
 *     stats      = rows x [2][Cred] partial column sums / sums of squares of y's fp32 accumulators (sr_bn_finalize consumes them)
 * (eval mode: folded BatchNorms -- y = relu(z . w_red^T + ybias) and no statistics, see `ybias`)
 * in one pass over z: the block output is written once and never read back by the reduce conv (per layer3 block 8.63 -> 6.17 GB of HBM
 * traffic at batch 6144).  x: the RAW output of the bottleneck's 3x3 conv with in_scale / in_shift = its BatchNorm (both NULL: x is
 * already normalised); escale / eshift: bn3's scale / shift (known before the launch: sr_bn_finalize_gram); res: the identity.
 *   sr_conv_pair_supported   1 if sr_conv_pair serves (M, Cmid, Cexp, Cred, dtype): bf16, Cexp = 4 Cmid, and Cred = Cmid in {256, 128, 64} (the
 *                            next block lies in the same layer: ResNet-50/101/152 layers 3 / 2 / 1) or Cred = 2 Cmid, Cmid in {128, 64} (the next
 *                            block opens the next layer: its conv1 is 1x1 / stride 1 on the block output as well)
 *   sr_conv_pair_pack_bytes  size of the packed weight stream
 *   sr_conv_pair_pack        w_exp [Cexp][Cmid], w_red [Cred][Cexp] (row-major, as sr_conv2d takes them) -> the stream the kernel's LDS
 *                            ring consumes (MFMA fragments in phase order; once per weight version)
 *   sr_conv_pair_stats_rows  rows of `stats` the launch writes (one per workgroup)
 * z is bit-identical to sr_conv2d's weight-stationary expansion kernel and y to the generic kernel fed that z (same fp32 FMA and
 * rounding points, same K order); the statistics differ from the unfused launch's in summation order only. */
typedef struct sr_pair_args {
  const void* x; const void* wpack; const void* res;
  void* z; void* y;
  const float* escale; const float* eshift;
  const float* in_scale; const float* in_shift;
  float* stats;            /* train mode; NULL in eval mode */
  const float* ybias;      /* eval mode (folded BatchNorm): y = [relu](z . w_red^T + ybias), no statistics; NULL in train mode.  Eval mode
                              passes the folded weights, escale = ones, eshift = conv3's folded bias, in_scale / in_shift = NULL */
  int64_t M;
  int32_t Cmid, Cexp;
  int32_t Cred, yrelu;     /* Cred: output channels of the reduce conv (0 = Cmid); yrelu: ReLU on y (eval mode) */
} sr_pair_args;
int sr_conv_pair_supported(int64_t M, int Cmid, int Cexp, int Cred, int dtype);
int sr_conv_pair_pack_bytes(int Cmid, int Cexp, int Cred);
int sr_conv_pair_pack(const void* w_exp, const void* w_red, void* out, int Cmid, int Cexp, int Cred, int dtype, void* stream);
int sr_conv_pair_stats_rows(int64_t M, int Cmid, int Cexp, int Cred);
int sr_conv_pair(const sr_pair_args* a, int dtype, void* stream);

/* Stem + BatchNorm + ReLU + 3x3/2 max-pool in ONE launch: y[b,po,qo,c] = max over the 3x3/2 window (pad 1) of
 * relu(conv7x7/2(xp, w)[.,.,c] * scale[c] + shift[c]), xp / w as for sr_conv2d with stem != 0 (bf16, 64 output channels),
 * y bf16 [B, Po, Qo, 64] with Po = (Ho-1)/2+1.  Replaces conv1 -> bn1 -> relu -> maxpool of torchvision's ResNet (call site
 * reference model.py:35) without ever writing the conv1 output: train mode runs sr_conv2d with no_store first (batch statistics
 * -> sr_bn_finalize -> scale / shift), eval mode passes the folded weights with scale = 1, shift = folded bias. */
int sr_stem_bn_relu_maxpool(const void* xp, const void* w, const float* scale, const float* shift, void* y, int B, int H, int W,
                            int dtype, void* stream);

/* fp32 NCHW image [B,3,H,W] -> zero-padded NHWC4 [B, Hp, Wp, 4] (Hp = (H+7)&~1, Wp = (W+7)&~1)
 * in `dtype`; replaces the layout work cuDNN does for the 7x7 stem (model.py:35). */
int sr_stem_prep(const float* img, void* out, int B, int H, int W, int dtype, void* stream);

/* uint8 input pipeline: decoded NHWC images [B,H0,W0,3] (device) -> the stem's padded NHWC4 input of an HxW crop, normalised
 * with mean3/std3 (HOST pointers, 3 floats each), per-image crop origin crop_yx int32 [B][2] (device, NULL = no crop, needs
 * H0==H, W0==W) and per-image horizontal flip flags uint8 [B] (device, NULL = none).  Fuses the reference's
 * RandomCrop/CenterCrop + RandomHorizontalFlip + ToTensor + Normalize (imsitu_encoder.py:21-36) with sr_stem_prep, so the
 * 3.7 GB fp32 NCHW batch never exists. */
int sr_image_prep_u8(const uint8_t* img, void* out, int B, int H0, int W0, int H, int W, const int32_t* crop_yx,
                     const uint8_t* flip, const float* mean3, const float* std3, int dtype, void* stream);

/* Train-mode BatchNorm, phase 2 (reference: model.train() at sr.py:16 puts the frozen
 * backbone's BatchNorm2d in batch-statistics mode).  Reduces the per-tile partials written by
 * sr_conv2d, produces scale = gamma*rsqrt(var+eps), shift = beta - mean*scale, and applies the
 * running-statistics EMA (momentum, unbiased variance).  running_* may be NULL. */
int sr_bn_finalize(const float* stats, int tiles, int C, int64_t count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps,
                   float* scale, float* shift, double* scratch, int scratch_rows,
                   float* running_mean2, float* running_var2, float momentum2, void* stream);
/* running_mean2 / running_var2 / momentum2 (NULL / 0 when unused): a SECOND BatchNorm that sees the same batch statistics --
 * FCGGNN's two backbones start from the same pretrained weights (reference model.py:16,100-101) and never train them, so
 * while they are identical one pass serves both, and the twin's buffers take its own EMA (two passes' worth: model.py:176-178). */
/* scratch: caller-owned fp64 workspace [scratch_rows][2][C] (scratch_rows >= 1; 1024 rows use full parallelism)
 * for the deterministic two-stage reduction of the partials. */

/* Train-mode BatchNorm statistics of a 1x1 / stride-1 convolution y = x W^T from the Gram matrix of its INPUT, so
 * the convolution is not run for them (replaces launch 1 of the two-launch scheme above for the bottleneck expansion
 * convs, where N = 4C):  sum_m y[m,n] = W[n,:].colsum(x),  sum_m y[m,n]^2 = W[n,:] (x^T x) W[n,:]^T.
 *   sr_gram_plan        number of fp32 partials sr_gram writes for (M, C) and the floats per partial (C*C + C)
 *   sr_gram             x [M, ldx] (bf16, C in {64,128,256,512}) -> partials [npartials][C*C + C]:
 *                       per row slice, the C x C Gram block sums followed by the C column sums
 *   sr_bn_finalize_gram reduces the partials in fp64, forms both sums per output channel of w [N, ldw] (bf16, the
 *                       conv's packed weights) in fp64 and finishes exactly like sr_bn_finalize.
 *                       scratch: fp64 workspace of at least 66*(C*C + C) elements. */
int sr_gram_plan(int64_t M, int C, int64_t* npartials, int64_t* partial_floats);
int sr_gram(const void* x, int64_t M, int C, int64_t ldx, int dtype, float* partials, int64_t npartials, void* stream);
/* sr_bn_apply (ReLU, no residual, in place) + sr_gram in one pass: x holds the raw output of the bottleneck's 3x3 conv on
 * entry and its normalised form on return; partials as sr_gram (C in {64,128,256}). */
int sr_bn_apply_gram(void* x, int64_t M, int C, int64_t ldx, int dtype, const float* scale, const float* shift,
                     float* partials, int64_t npartials, void* stream);
/* As sr_bn_apply_gram, but x is NOT modified: the Gram partials are those of relu(x*scale + shift), which only ever exists in
 * LDS -- for a consumer that applies the same affine on load (sr_conv2d with in_scale / in_shift; call site reference model.py:35,
 * bn2 -> relu -> conv3 of a bottleneck). */
int sr_bn_gram(const void* x, int64_t M, int C, int64_t ldx, int dtype, const float* scale, const float* shift,
               float* partials, int64_t npartials, void* stream);
int sr_bn_finalize_gram(const float* partials, int64_t npartials, int C, const void* w, int64_t ldw, int N, int dtype,
                        int64_t count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float momentum, float eps, float* scale, float* shift, double* scratch, int64_t scratch_elems,
                        float* running_mean2, float* running_var2, float momentum2 /* twin BatchNorm, as sr_bn_finalize */,
                        void* stream);

/* ---- FP8 (OCP e4m3) path of the matrix-bound 3x3 convolutions (BASELINE config 5; the reference's reduced-precision route is
 * autocast, model.py:33,58,114,157,171).  e4m3 activations x e4m3 weights on v_mfma_scale_f32_16x16x128_f8f6f4 (2x the bf16
 * MFMA rate), fp32 accumulation; everything around the 3x3 stays bf16.
 *   sr_quantize_fp8: out[i] = e4m3( f(x[i]) * act_scale ), f = [relu](x*scale[c] + shift[c]) when scale/shift are given (the
 *     BatchNorm-apply in front of the conv writes its fp8 input directly), identity otherwise; x bf16 or fp32 [rows, C], C % 8 == 0.
 *   sr_conv3x3_fp8: y[b,ho,wo,co] = dq[co] * sum x[b, ho*s-1+r, wo*s-1+q, c] * w[co,r,q,c]  (pad 1, stride 1 or 2), x and w e4m3,
 *     y bf16, dq = 1 / (act_scale * weight_scale[co]) fp32; Cin in {128,256,512}, Cout % 128 == 0.  stats (optional): partial
 *     column sums / sums of squares of the dequantised fp32 values, rows = sr_conv3x3_fp8_stats_rows(M, Cout), same layout and
 *     meaning as sr_gemm_args.stats (sr_bn_finalize consumes them). */
int sr_quantize_fp8(const void* x, const float* scale, const float* shift, void* out, int64_t rows, int C, int relu, float act_scale,
                    int dtype, void* stream);
int sr_conv3x3_fp8_stats_rows(int M, int N);
int sr_conv3x3_fp8(const void* x, const void* w, const float* dq, void* y, float* stats, int B, int H, int W, int Cin, int Cout,
                   int stride, void* stream);

/* y = [relu]( x*scale[c] + shift[c] (+ res) ), rows x C, in place allowed. */
int sr_bn_apply(const void* x, const float* scale, const float* shift, const void* res, void* y,
                int64_t rows, int C, int relu, int dtype, void* stream);
/* 3x3/2 pad 1 max-pool over NHWC; scale/shift (nullable) are applied (+ReLU) to every input first. */
int sr_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, const float* scale, const float* shift,
                    int dtype, void* stream);
/* global average pool [B, HW, C] -> [B, C] */
int sr_avgpool(const void* x, void* y, int B, int HW, int C, int dtype, void* stream);

/* node[b*R+r,:] = relu(feat[b,:] * role_emb[role_ids[verb[b]][r],:] * verb_emb[verb[b],:])
 * (reference model.py:117-144 incl. the encoder lookup imsitu_encoder.py:172-180 done on device).
 * role_table: int32 [V][R].
 *
 * PACKED ROLE ROWS (`offs` != NULL in this and the two functions below).  A verb with k < R roles has R - k padded role slots.  Their
 * nodes start at exactly 0 (role_emb's padding row is 0, model.py:95-97), their adjacency row is only their own diagonal 1
 * (imsitu_encoder.py:209-229) and no real role reads them, so through all T steps EVERY padded slot of EVERY image holds the same
 * vector -- one trajectory that depends on the weights alone -- and the reference computes it B*(R-k) times.  In the packed form
 * only the real roles' rows exist: offs int32 [B+1] = prefix sums of the images' role counts, image b's role r is row offs[b] + r,
 * and ONE more row, offs[B], stands for all padded slots (the caller zero-fills it in the node tensor; sr_ggnn_aggregate copies it
 * through as "image" B).  The caller expands the packed classifier output back to [B, R, L] (padded slots <- the shared row) and
 * folds the gradients of all padded slots into the shared row, so results and gradients equal the full form's. */
int sr_node_init_fwd(const void* feat, const float* role_emb, const float* verb_emb, const int64_t* verbs,
                     const int32_t* role_table, void* node, int B, int R, int D, int dtype, const int32_t* offs, void* stream);
/* gradients of the above: fp32 d_role_emb [NR+1,D] and d_verb_emb [V,D], both written IN FULL (the padding row NR and
 * the rows of verbs / roles absent from the batch get zeros, as nn.Embedding(padding_idx) does) with a fixed summation
 * order -- no atomics, bit-reproducible.  The caller supplies the batch grouped by verb and the role table's inverted index:
 *   order    int32 [B]     image indices sorted (stably) by verb id
 *   seg      int32 [V+1]   order[seg[v] .. seg[v+1]) are the images of verb v
 *   inv_ptr  int32 [NR+1], inv_slot int32 [nnz]: inv_slot[inv_ptr[q] .. inv_ptr[q+1]) = the slots v*R+r with role_table[v][r] == q
 *   scratch  fp32 [(V + 2*ceil(B/32)) * R * D]  per-(verb, slot) sums, then two partial-sum slabs per 32-image chunk of `order`
 *            (a verb's images are summed chunk by chunk, so that a batch dominated by one verb is not walked by one wave;
 *            ABI 4: the first form took V*R*D floats)
 * feat gets no gradient (frozen backbone, model.py:17-18). */
int sr_node_init_bwd(const void* dnode, const void* feat, const float* role_emb, const float* verb_emb,
                     const int32_t* order, const int32_t* seg, const int32_t* role_table, const int32_t* inv_ptr,
                     const int32_t* inv_slot, float* scratch, float* d_role_emb, float* d_verb_emb,
                     int B, int R, int D, int V, int NR, int dtype, const int32_t* offs /* packed rows of dnode, or NULL */, void* stream);

/* out[b,i,:] = sum_j A[verb[b]][i][j] * h[b,j,:] (+ add[b,i,:])   (transpose != 0: A^T)
 * The role-graph message step of GGSNN.forward (model.py:66-77) in its algebraic form, with the
 * adjacency lookup of imsitu_encoder.py:209-229 done on device from adj_table fp32 [V][R][R].
 * offs != NULL: packed role rows (see sr_node_init_fwd): h / add / out hold offs[B] + 1 rows. */
int sr_ggnn_aggregate(const void* h, const float* adj_table, const int64_t* verbs, const void* add, void* out,
                      int B, int R, int D, int transpose, int dtype, const int32_t* offs, void* stream);

/* GRU backward, stage 1 (model.py:84,82-83,80 differentiated):
 *   dc_pre = dh*z*(1-c^2); dz_pre = dh*(c-h)*z*(1-z); dh_acc = dh*(1-z) */
int sr_gru_bwd1(const void* dh, const void* z, const void* c, const void* h, void* dc_pre, void* dz_pre,
                void* dh_acc, int64_t n, int dtype, void* stream);
/* stage 2 (model.py:81,83): dr_pre = drh*h*r*(1-r); dh_acc += drh*r */
int sr_gru_bwd2(const void* drh, const void* r, const void* h, void* dr_pre, void* dh_acc, int64_t n, int dtype,
                void* stream);

/* out[C, ld_out] = in[R,C]^T (in row stride ld_in) with optional dtype change; columns R..ld_out of
 * every output row are zero-filled (pads the reduction dimension of a following dW GEMM);
 * optional fp32 column sums colsum[C] += scale * sum_r in[r,c] (bias gradients). out may be NULL. */
int sr_transpose(const void* in, int64_t ld_in, void* out, int64_t R, int64_t C, int64_t ld_out, int in_dtype,
                 int out_dtype, float* colsum, float colsum_scale, void* stream);
/* out[N1, N2] (+)= A^T B for A [M, lda] (N1 columns) and B [M, ldb] (N2 columns), bf16, fp32 out with row stride N2;
 * N1, N2 multiples of 256, any M.  The weight-gradient GEMM dW = dY^T X of the GGNN / classifier backward (autograd of the
 * nn.Linear calls at reference model.py:64,75,80-83) without transposed copies of the operands: both MFMA operands are
 * read out of row-major LDS images with transposed reads.  Rows are cut into sr_gemm_tn_slices(M, N1, N2) slices (so that
 * small outputs still fill the chip), each writing an fp32 partial into scratch (>= slices*N1*N2 floats), folded into out
 * (added to it when accumulate != 0). */
int sr_gemm_tn_slices(int64_t M, int N1, int N2);
int sr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int N1, int N2, int dtype, float* out,
               int accumulate, float* scratch, int64_t scratch_floats, void* stream);
/* colsum[C] += scale * sum_r in[r,c] */
int sr_colsum(const void* in, int64_t ld_in, int64_t R, int64_t C, int dtype, float* colsum, float scale, void* stream);
/* out[r, 0..Cpad) = cast(in[r, 0..C)), zero fill of [C, Cpad); row strides ld_in / ld_out */
int sr_cast_pad(const void* in, int64_t ld_in, void* out, int64_t ld_out, int64_t R, int64_t C, int64_t Cpad,
                int in_dtype, int out_dtype, void* stream);
/* elementwise cast between SR_F32 and SR_BF16 */
int sr_cast(const void* in, void* out, int64_t n, int in_dtype, int out_dtype, void* stream);
/* y = x * keep(i) * 2, keep(i) = bit of a counter-based hash of (seed, i): Dropout(0.5) of the
 * classifiers (model.py:105-111).  mask_out (uint8, nullable) receives keep(i).  The same call with
 * x = upstream gradient is the backward. */
int sr_dropout_half(const void* x, void* y, uint8_t* mask_out, int64_t n, uint64_t seed, int dtype, void* stream);

/* ---- Data-parallel gradient exchange: RCCL all-reduce over xGMI (one process per GPU).
 * Replaces nn.DataParallel (reference sr.py:467-470), whose backward (sr.py:79) sums the replicas' gradients onto GPU 0 with
 * reduce_add_coalesced: here every rank holds its trainable gradients in one flat buffer and sums it in place across the ranks.
 *   sr_comm_unique_id  rank 0 fills SR_COMM_ID_BYTES (an ncclUniqueId) and hands them to the other ranks by any host-side channel
 *   sr_comm_init       collective over `world` processes, each with its own current HIP device; *comm_out = the communicator
 *   sr_comm_world      number of ranks of a communicator (>= 1) or a negative error
 *   sr_allreduce_sum   buf[i] <- sum over ranks of buf[i], in place, `count` elements of `dtype` (SR_F32 / SR_BF16), enqueued on
 *                      `stream`; every rank must issue the same sequence of calls
 *   sr_comm_destroy    releases the communicator
 * RCCL itself is loaded on first use (librccl.so.1); SR_ERR_UNSUPPORTED when it cannot be found. */
#define SR_COMM_ID_BYTES 128
int sr_comm_unique_id(void* id_out);
int sr_comm_init(const void* id_in, int rank, int world, void** comm_out);
int sr_comm_world(void* comm);
int sr_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream);
int sr_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif
