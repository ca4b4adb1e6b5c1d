/*
 * ffvc.h — C ABI of libffvc_hip.so, the MI355X (gfx950) kernel library behind
 * feed_forward_vqgan_clip_amd.
 *
 * The reference (mehdidc/feed_forward_vqgan_clip) is pure Python and has NO
 * FFI/plugin interface: every FLOP of its training step is dispatched by stock
 * torch ops from main.py:715-837.  This header is therefore the
 * boundary a maintainer would bind *instead of* those torch calls; each entry
 * point cites the reference lines whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers + sizes, no torch / pybind types.
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates);
 *     the library borrows it for the duration of the enqueue and keeps nothing.
 *   - `stream` is a hipStream_t passed as void*; work is only ENQUEUED on it.
 *   - return value: 0 = ok, otherwise a hipError_t (>0) or FFVC_E_* (<0);
 *     ffvc_last_error() returns a human readable message for the calling thread.
 *   - dtype codes: FFVC_BF16 = 0 (bfloat16 storage, fp32 accumulate),
 *                  FFVC_F32  = 1 (exact fp32 "parity mode": fp32-input MFMA),
 *                  FFVC_F16  = 2 (IEEE half storage, fp32 accumulate; what the reference's CLIP runs in on CUDA,
 *                                 clip.load -> fp16 weights).
 */
#ifndef FFVC_H
#define FFVC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFVC_BF16 0
#define FFVC_F32 1
#define FFVC_F16 2 /* IEEE half storage, fp32 accumulate: 8x finer than bf16 at the same MFMA rate (loss-scaled backward) */

#define FFVC_E_BADARG (-1)
#define FFVC_E_UNSUPPORTED (-2)

/* activation codes for the GEMM epilogue */
#define FFVC_ACT_NONE 0
#define FFVC_ACT_GELU 1      /* exact erf GELU: mlp_mixer_pytorch.py:19, vitgan.py:33 */
#define FFVC_ACT_QUICKGELU 2 /* x*sigmoid(1.702x): cloob.py:179-181 */
#define FFVC_ACT_LRELU 3     /* LeakyReLU(0.01): the MLPs of the Net2Net prior (main.py:1453-1462) [upstream net2net] */
#define FFVC_ACT_TANH 4      /* tanh: scale heads of the prior's coupling blocks */

/* operand storage modes */
#define FFVC_OP_KMAJOR 0  /* operand[row][k], k contiguous                      */
#define FFVC_OP_TRANS 1   /* operand stored [k][row], row contiguous            */
#define FFVC_OP_CONV3X3 2 /* X only: implicit im2col of an NHWC tensor, 3x3/p1  */

/* epilogue flags */
#define FFVC_F_BIAS_ALONG_M 1   /* bias indexed by output row instead of column */
#define FFVC_F_WRITE_PREACT 2   /* aux <- pre-activation (dtype = in_dtype)     */
#define FFVC_F_MUL_ACT_GRAD 4   /* acc *= act'(aux) (backward of fused act)     */
#define FFVC_F_ATOMIC_OUT 8     /* y += acc with fp32 atomics (split-K / wgrad) */
#define FFVC_F_RES_F32 16       /* residual is fp32 (else in_dtype)             */
#define FFVC_F_OUT_F32 32       /* y is fp32 (else in_dtype)                    */
#define FFVC_F_TR_SAFE 64       /* transposed bf16 fragments via scalar LDS gathers (debug/verification) */
#define FFVC_F_UPSAMPLE2X 128   /* conv: input is nearest-2x upsampled on the fly */
#define FFVC_F_GN_SUMS 512      /* also accumulate GroupNorm moments of the stored output (see gn_sums below) */
#define FFVC_F_COLSUM 1024      /* also accumulate the column sums of the stored output into colsum[N] (see below) */
#define FFVC_F_AUX_ACTGRAD 2048 /* aux holds act'(pre-activation) instead of the pre-activation.  With FFVC_F_WRITE_PREACT the
                                 * forward stores the derivative (formed from the same erf / exp as the activation; launches
                                 * that cannot take the specialised kernel store the pre-activation and convert it in a second
                                 * pass), with FFVC_F_MUL_ACT_GRAD the backward epilogue is a plain multiply.  16-bit dtypes,
                                 * batch 1. */
#define FFVC_F_ACCUM_OUT 256    /* y += acc with plain read-modify-write (fp32 y, split_k == 1: one owner per element) */
#define FFVC_F_GNB_SUMS 8192    /* the stored output is the gradient dy of a GroupNorm(+swish) output: also accumulate that node's BACKWARD
                                 * statistics (see gnb_* below) — the statistics pass of ffvc_groupnorm_bwd folded into the dgrad convolution
                                 * that produces dy.  conv3 row-tile kernel only (ffvc_gemm_gnb_probe says whether a launch takes it). */
#define FFVC_F_VQ_ARGMIN 16384  /* nothing is stored: the launch keeps, per row m, the first n that minimises (vq_xn[m] + vq_cn[n]) - 2 * acc[m, n]
                                 * (see vq_* below) */
#define FFVC_F_SPLITK_INKERNEL 4096 /* split_k > 1 handled INSIDE the launch: every K slice parks its fp32 partial tile in library
                                  * scratch, the last slice to arrive on a tile sums them in slice order and runs the ordinary
                                  * epilogue (any epilogue, FFVC_F_ACCUM_OUT included): no slabs, no reduce launch.  16-bit LDS-DMA
                                  * kernels only (K-major x K-major 128x128 / 256x256 tiles, weight gradients on 256x256 tiles);
                                  * ffvc_gemm fails with FFVC_E_BADARG when the shape cannot take that path; split_k <= 8 (16 for
                                  * the 128x128 tile) */

/*
 * ffvc_gemm — y[m,n] (+)= act( alpha * sum_k X[m,k] * W[n,k] + bias ) (+ residual)
 *
 * One MFMA kernel family serves every dense contraction on the hot path:
 *   Linear fwd          (x @ W^T + b)         mlp_mixer_pytorch.py:16-23,31-35,76-78; vitgan.py:29-30,64,67; cloob.py:188-196
 *   Conv1d(k=1) tokens  (W @ x[b])            mlp_mixer_pytorch.py:28,34      (X = weight, W = x[b] stored TRANS, batched)
 *   dgrad               (dy @ W)              autograd of the above           (W^T shadow, KMAJOR)
 *   wgrad               (dy^T @ x)            autograd of the above           (both operands TRANS, split-K + atomics)
 *   3x3 conv fwd/dgrad  (implicit GEMM)       taming Decoder/ResnetBlock/Upsample [upstream, SURVEY App. A.1]
 *   1x1 conv, attention QK^T / PV (batched)   taming AttnBlock; cloob.py:199-200; vitgan.py:90-93
 *   VQ distance         (x @ codebook^T, f32) main.py:134-136
 *
 * Index maps (all offsets in ELEMENTS):
 *   X KMAJOR: x[ xb(z) + xrow(m) + (k/kseg)*xkso + (k%kseg) ]    TRANS: x[ xb(z) + k*ldx + m ]
 *             xrow(m) = x_mi ? (m/x_mi)*x_so + (m%x_mi)*ldx : m*ldx   (e.g. rows 1..49 of every 50-row image)
 *   W KMAJOR: w[ wb(z) + n*ldw + (k/kseg)*wkso + (k%kseg) ]      TRANS: w[ wb(z) + k*ldw + n ]
 *   y       : y[ yb(z) + (m/y_mi)*y_so + (m%y_mi)*y_sm + n ]
 *   residual: r[ rb(z) + (m/r_mi)*r_so + (m%r_mi)*r_sm + n ]
 *   aux     : aux[ ab(z) + m*ldaux + n ]
 *   batch   : ?b(z) = (z / batch_inner) * ?bo + (z % batch_inner) * ?bi
 *   CONV3X3 : X is NHWC [B, Hin, Win, Cin]; m = (b*H + oy)*W + ox over the OUTPUT grid
 *             (H = Hin, or 2*Hin with FFVC_F_UPSAMPLE2X); k = (kh*3+kw)*Cin + ci; K = 9*Cin;
 *             W is [Cout][3][3][Cin] (KMAJOR).  Cin must be a multiple of 64 (bf16) / 32 (f32).
 * Alignment: operand pointers, leading dims and batch / segment strides must keep 4-byte alignment (even element
 * counts for bf16); the vectorised epilogue additionally needs 4-element aligned y / residual / aux rows, else a
 * scalar epilogue is used.
 */
typedef struct ffvc_gemm_desc {
  const void* x;
  const void* w;
  void* y;
  const float* bias; /* fp32, may be NULL */
  const void* residual; /* may be NULL */
  void* aux;            /* pre-activation in/out, may be NULL */
  int32_t M, N, K;
  int32_t x_mode, w_mode;
  int32_t in_dtype; /* FFVC_BF16 / FFVC_F32: dtype of x, w, aux (and y/residual unless flagged) */
  int32_t act;
  int32_t flags;
  int32_t split_k; /* >=1; >1 requires FFVC_F_ATOMIC_OUT */
  float alpha;
  int64_t ldx, ldw, ldaux;
  int32_t kseg; /* 0 => K */
  int64_t xkso, wkso;
  int32_t y_mi; /* 0 => plain rows: off = m*y_sm */
  int64_t y_so, y_sm;
  int32_t r_mi;
  int64_t r_so, r_sm;
  int32_t batch, batch_inner; /* batch>=1, batch_inner>=1 */
  int64_t xbo, xbi, wbo, wbi, ybo, ybi, rbo, rbi, abo, abi;
  /* conv geometry (x_mode == FFVC_OP_CONV3X3) */
  int32_t conv_H, conv_W, conv_Cin;
  /* optional split row map of a KMAJOR X operand (0 = plain m*ldx) */
  int32_t x_mi;
  int64_t x_so;
  /* split-K through partial slabs: K-slice z stores its fp32 partial tile at y + z*slab_stride (plain vector
   * stores); combine with ffvc_slab_reduce.  0 = off. */
  int64_t slab_stride;
  /* FFVC_F_GN_SUMS: the rows of y are pixels of NHWC images with gn_hw pixels each and N channels in groups of
   * gn_cpg consecutive channels; gn_sums[image][group][2] (fp64, zeroed by the caller) receives sum and sum of squares
   * of the fp32 values just before the store (taming Normalize statistics of the NEXT layer, computed where the tensor is produced
   * instead of by a separate read pass).  bf16 LDS-DMA path only; gn_hw % 256 == 0, gn_cpg % 4 == 0. */
  double* gn_sums;
  int32_t gn_hw, gn_cpg;
  /* FFVC_F_COLSUM: colsum[n] += sum_m y[m, n] of the values just stored (fp32 atomics, one per column and wave): the
   * bias gradient of a Linear whose pre-activation gradient this GEMM produces (FFVC_F_MUL_ACT_GRAD epilogue), computed
   * where the tensor is produced instead of by a separate read pass.  16-bit LDS-DMA path with the row-store epilogue
   * only, batch == 1, no split-K. */
  float* colsum;
  /* RESERVED, leave NULL: scratch of the in-kernel split-K (set by the library itself when an under-filled K-major x K-major
   * launch is split along K: every K slice parks its fp32 partial tile, the last workgroup to arrive on a tile sums them
   * in slice order and runs the ordinary epilogue, so any fused epilogue stays available). */
  float* sk_ws;
  uint32_t* sk_cnt;
  /* RESERVED, leave 0: tail mode of the same machinery — work items >= sk_full are the tiles of the last, partly filled round of
   * a 256x256-tile launch, each cut into sk_slices K slices (260 tiles on 256 CUs would otherwise run two full rounds). */
  int32_t sk_full, sk_slices;
  /* fp8 OUTPUT (ffvc_gemm_fp8 only; NULL = off): y receives fp8 bytes (y8_fmt 0 = e4m3 | 1 = e5m2) = saturate(value * y8_state[0])
   * instead of 16-bit values, and y8_state[1] = max(y8_state[1], max |value|) (delayed per-tensor scaling, as ffvc_fp8_quant keeps
   * it): the producer of the next fp8 GEMM's operand writes the operand itself.  Available with the two MLP kinds of the frozen
   * towers — the activation forward that stores act'(pre) (e4m3 hidden activation) and the aux-multiply backward (e5m2 hidden
   * gradient) — on the 256x256 tile; y rows are N bytes. */
  float* y8_state;
  int32_t y8_fmt;
  /* GROUPED launch (0 = off): the weight gradients of grp_n layers of the same kind in ONE launch — batch entry z reads its
   * operands at x + grp_xoff[z] / w + grp_woff[z] (ELEMENT offsets from the descriptor's base pointers; signed: the layers'
   * activations are separate allocations) instead of the constant batch strides xbo / wbo, and writes y + z * ybo (the layers'
   * gradients sit at a constant stride in the flat gradient bucket).  mlp_mixer_pytorch.py:16-23 x depth: a single weight gradient
   * of the channel MLP has 64 tiles of 256x256 — a quarter of the chip — and needed a 4-way split-K through fp32 slabs + a reduce
   * pass; four layers together are 256 full-K tiles.  Needs grp_n == batch <= 8, batch_inner == 1, both operands FFVC_OP_TRANS, a
   * 16-bit dtype, offsets that keep 16-byte alignment; ffvc_gemm fails with FFVC_E_UNSUPPORTED when the shape does not take the
   * 256x256 LDS-DMA weight-gradient kernel. */
  int32_t grp_n;
  int64_t grp_xoff[8], grp_woff[8];
  /* FFVC_F_GNB_SUMS (round 6; reference: taming Decoder ResnetBlock backward): y = dy [M = images * gn_hw pixels,
   * N channels] is the gradient w.r.t. act(GroupNorm(gnb_x)) (act = swish when gnb_swish).  With xh = (gnb_x - mean) * rstd,
   * ds = round_16(dy) * act'(xh * gamma + beta) * gamma the launch adds, per (image, group of gn_cpg channels),
   *     gnb_sums[image][group][0] += sum ds,   gnb_sums[image][group][1] += sum ds * xh          (fp64 atomics, buffer zeroed by the caller)
   * — exactly what the statistics pass of ffvc_groupnorm_bwd computes from dy and x (2 of its 5 reads); ffvc_groupnorm_bwd_sums then
   * only applies.  gnb_x: 16-bit, contiguous [M, N] like y (y_sm == N, no row map, batch 1); gnb_mean / gnb_rstd: fp32 [images, N / gn_cpg];
   * gnb_gamma / gnb_beta: fp32 [N].  gn_hw % 256 == 0, gn_cpg % 4 == 0. */
  const void* gnb_x;
  const float* gnb_mean;
  const float* gnb_rstd;
  const float* gnb_gamma;
  const float* gnb_beta;
  double* gnb_sums;
  int32_t gnb_swish;
  /* FFVC_F_VQ_ARGMIN (round 6; reference main.py:133-139 `vector_quantize`: d = x.pow(2).sum + codebook.pow(2).sum - 2 x @ codebook.T,
   * indices = d.argmin(-1)): x = the rows to quantise [M, K], w = the codebook [N, K] (both K-major, 16-bit — e.g. the split-precision
   * operands of ffvc_split3), vq_xn[M] / vq_cn[N] the fp32 squared norms.  The distance matrix is never stored: every wave folds its
   * 64 columns in registers — same fp32 expression, same order as ffvc_vq_argmin — and sends one 64-bit atomic minimum per row to
   * vq_out[m] = (order-preserving bits of d) << 32 | n, which the caller initialises to all ones; afterwards the low 32 bits are the
   * index (first minimum, as torch.argmin).  256x256 LDS-DMA kernel only: batch 1, split_k 1, alpha 1, no bias / residual / aux /
   * activation; y is ignored (pass vq_out).  ffvc_gemm fails with FFVC_E_UNSUPPORTED when the operands do not take that kernel. */
  const float* vq_xn;
  const float* vq_cn;
  uint64_t* vq_out;
} ffvc_gemm_desc;

int ffvc_gemm(const ffvc_gemm_desc* d, void* stream);
/* 1 if ffvc_gemm would run this descriptor (flags incl. FFVC_F_GNB_SUMS) on a kernel that implements FFVC_F_GNB_SUMS, 0 if not; nothing
 * is launched.  (The fusion exists on the pipelined row-tile convolution only; callers ask once per shape and otherwise keep the
 * separate statistics pass.) */
int ffvc_gemm_gnb_probe(const ffvc_gemm_desc* d, void* stream);
/* The plain Linear product for a handful of rows (M <= 64, N % 32 == 0, K % 256 == 0: ffvc_gemm_skinny_ok), 16-bit K-major operands with
 * row stride K: the K loop is split across the eight waves of a workgroup — the 64-row remainder of the ViT-L/14 tower's 64 x 257 rows
 * (cloob.py:199-205 at BASELINE configs[4]).  y (fp32 or in_dtype, row stride N) = X W^T (+ bias) (+ residual in fp32 or in_dtype). */
int ffvc_gemm_skinny_ok(int M, int N, int K);
int ffvc_gemm_skinny(const void* x, const void* w, int in_dtype, void* y, int y_dtype, const float* bias, const void* residual,
                     int res_dtype, int M, int N, int K, void* stream);
/* aux[m, n] <- act'(aux[m, n]) in place, 16-bit storage (the conversion pass behind FFVC_F_AUX_ACTGRAD on kernels without
 * the specialised epilogue). */
int ffvc_actgrad_inplace(void* aux, int dtype, int act, int M, int N, int64_t ld, void* stream);

/* OCP fp8 GEMM for the frozen towers (BASELINE.json configs[4], "fp8 MFMA path"): x and w hold fp8 bytes (w: e4m3fn;
 * x: e4m3fn for x_fmt 0, e5m2 for x_fmt 1 = gradients), K-major both, fp32 accumulation on
 * v_mfma_f32_32x32x64_f8f6f4 (the only fp8 instruction of gfx950 that runs at twice the bf16 rate).  d->K, ldx, ldw
 * count fp8 ELEMENTS and must be multiples of 16; y / aux / residual use `lo_dtype` (FFVC_BF16 | FFVC_F16) unless flagged
 * fp32; d->in_dtype is ignored.  The accumulator is multiplied by d->alpha * scale0[0] * scale1[0] (device scalars, may be
 * NULL): the inverse per-tensor quantisation scales.  Same fused epilogues as ffvc_gemm; no batch, no split-K.
 * ffvc_fp8_quant: dst = saturate(src * state[0]) as fp8 (fmt 0 e4m3 | 1 e5m2), state[1] = max(state[1], max|src|).
 * ffvc_fp8_amax: state[1] = max(state[1], max|src|).  ffvc_fp8_update: state[0] = fmt_max / (state[1] * margin),
 * state[2] = 1 / state[0], state[1] = 0  (delayed scaling: the amax seen at step t sets the scale of step t+1).
 * state: 4 floats on the device.  n must be a multiple of 8. */
int ffvc_gemm_fp8(const ffvc_gemm_desc* d, int x_fmt, int lo_dtype, const float* scale0, const float* scale1, void* stream);
/* The same product for a handful of rows (M <= 64, N % 32 == 0, K % 512 == 0: ffvc_gemm_fp8_skinny_ok): the K loop is split across the
 * eight waves of a workgroup instead of walked by one tile's workgroup alone — the 64-row remainder of the ViT-L/14 tower's 64 x 257 rows
 * (cloob.py:199-205 linears at BASELINE configs[4]).  y (y_dtype fp32 | f16 | bf16, row stride N) = scale0 * scale1 * X8 W8^T (+ bias)
 * (+ residual in res_dtype = fp32 or y_dtype); x8 / w8 K-major with row stride K. */
int ffvc_gemm_fp8_skinny_ok(int M, int N, int K);
int ffvc_gemm_fp8_skinny(const void* x8, const void* w8, void* y, int y_dtype, const float* bias, const void* residual, int res_dtype,
                         int M, int N, int K, int x_fmt, const float* scale0, const float* scale1, void* stream);
int ffvc_fp8_quant(const void* src, int src_dtype, void* dst, int fmt, float* state, int64_t n, void* stream);
int ffvc_fp8_amax(const void* src, int src_dtype, float* state, int64_t n, void* stream);
int ffvc_fp8_update(float* state, int fmt, float margin, void* stream);
/* ffvc_fp8_update of every state of a contiguous pool [n][4] whose running amax is non-zero (state[3] holds the format code as a float:
 * 0 e4m3 | 1 e5m2): ONE launch per step instead of one per tensor stream (round 4: 325 launches per cfg5 step). */
int ffvc_fp8_update_many(float* pool, int n, float margin, void* stream);

/* ---------------------------------------------------------------------------
 * Normalisation / softmax (HBM-bound; fp32 statistics; dtype codes per tensor)
 * ------------------------------------------------------------------------- */

/* LayerNorm over the last dim, eps as given (1e-5 everywhere on the path):
 * mlp_mixer_pytorch.py:11,14,37; vitgan.py:14,21; cloob.py:170-176,203-204.  mean/rstd [rows] are saved. */
int ffvc_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                       float* mean, float* rstd, int64_t rows, int dim, float eps, void* stream);
/* dx = LN'(dy) (+ dres, same dtype as x/dx).  If part_g/part_b are given they receive
 * ffvc_layernorm_bwd_blocks(rows) partial rows of dgamma/dbeta (reduce with ffvc_colsum).  dx_lo (optional, only with
 * an fp32 dx): a bf16 copy of dx written in the same pass, for the GEMM that consumes this gradient. */
/* Producer-side quantisation (round 4; the fp8 tower / decoder of BASELINE configs[4], reference main.py:140-143 + cloob.py:170-176):
 * the normalised row also leaves as fp8 bytes y8 = saturate(round_y_dtype(y) * f8_state[0]) (f8_fmt 0 e4m3 | 1 e5m2) and f8_state[1]
 * collects max |round_y_dtype(y)| — byte for byte what ffvc_fp8_quant makes of y.  y may be NULL (fp8 output only).  dim % 4 == 0,
 * y_dtype 16-bit. */
int ffvc_layernorm_fwd_f8(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, void* y8,
                          float* f8_state, int f8_fmt, float* mean, float* rstd, int64_t rows, int dim, float eps, void* stream);
/* Backward of a FROZEN LayerNorm on the fp32 residual stream (the CLIP towers, cloob.py:203-204) whose result feeds an fp8 dgrad: dx fp32
 * (+ dres) and dx8 = the fp8 bytes ffvc_fp8_quant would make of round_{dy_dtype}(dx); no parameter gradients. */
int ffvc_layernorm_bwd_f8(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, void* dx8, float* f8_state, int f8_fmt, int64_t rows, int dim, void* stream);
int ffvc_layernorm_bwd_blocks(int64_t rows);
int ffvc_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                       const float* mean, const float* rstd, const void* dres, void* dx, float* part_g,
                       float* part_b, void* dx_lo, int64_t rows, int dim, void* stream);
/* Same pass, but dgamma / dbeta ([dim] fp32, e.g. slices of the flat gradient bucket) are ACCUMULATED in place with one
 * fp32 atomic per column and workgroup: no partial rows, no follow-up reduction launches (mlp_mixer_pytorch.py:24-25,
 * cloob.py:153-174 LayerNorm parameter gradients under torch autograd). */
int ffvc_layernorm_bwd_acc(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                           const float* mean, const float* rstd, const void* dres, void* dx, float* dgamma,
                           float* dbeta, void* dx_lo, int64_t rows, int dim, void* stream);

/* Self-modulated LayerNorm of the VitGAN generator (vitgan.py:8-21): y = gamma_s*w*LN(hl) + beta_s*w with scalar
 * parameters gamma_s/beta_s (device pointers) and the per-token modulation w; hl, w fp32 [rows, dim].
 * Backward emits dhl (+dres), dw, LN dgamma/dbeta partial rows and {sum dy*w*ln, sum dy*w} partial pairs
 * (ffvc_layernorm_bwd_blocks(rows) rows each). */
int ffvc_sln_fwd(const float* hl, const float* w, const float* gamma, const float* beta, const float* gamma_s,
                 const float* beta_s, void* y, int y_dtype, float* mean, float* rstd, int64_t rows, int dim, float eps,
                 void* stream);
int ffvc_sln_bwd(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma, const float* beta,
                 const float* gamma_s, const float* beta_s, const float* mean, const float* rstd, const float* dres,
                 float* dhl, float* dw, float* part_g, float* part_b, float* part_s, int64_t rows, int dim, void* stream);
/* Same pass, dgamma / dbeta ([dim]) and dscalars ({dgamma_s, dbeta_s}) ACCUMULATED in place (fp32 atomics), like
 * ffvc_layernorm_bwd_acc: no partial rows, no reduction launches. */
int ffvc_sln_bwd_acc(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma, const float* beta,
                     const float* gamma_s, const float* beta_s, const float* mean, const float* rstd, const float* dres,
                     float* dhl, float* dw, float* dgamma, float* dbeta, float* dscalars, int64_t rows, int dim,
                     void* stream);
/* Same, for parameters that live apart in a flat gradient bucket: the two scalar gradients go to separate addresses, and with
 * dw_accumulate != 0 the gradient of the modulation input is ADDED to dw (vitgan.py:132,256: every SLN of the generator is
 * modulated by the same w, so its gradient is one running sum instead of 65 tensors + 64 additions). */
int ffvc_sln_bwd_acc2(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma, const float* beta,
                      const float* gamma_s, const float* beta_s, const float* mean, const float* rstd, const float* dres,
                      float* dhl, float* dw, int dw_accumulate, float* dgamma, float* dbeta, float* dgamma_s, float* dbeta_s,
                      int64_t rows, int dim, void* stream);

/* GroupNorm(G groups, affine) on NHWC [B, HW, C] with optional fused swish x*sigmoid(x):
 * taming Normalize()/nonlinearity() [upstream taming-transformers 0.0.6, SURVEY.md App. A.1].
 * ws: workspace of ffvc_groupnorm_ws_bytes(B, HW, G) bytes.  mean/rstd: [B, G] fp32. */
int64_t ffvc_groupnorm_ws_bytes(int B, int HW, int G);
int ffvc_groupnorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                       void* ws, int B, int HW, int C, int G, float eps, int swish, int dtype, void* stream);
/* Same, with the per-(image, group) sums already accumulated by the producing GEMM (FFVC_F_GN_SUMS): only the
 * normalise / affine / swish pass runs. */
int ffvc_groupnorm_fwd_sums(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd,
                            const double* sums, int B, int HW, int C, int G, float eps, int swish, int dtype,
                            void* stream);
/* dx = GN'(swish'(.) * dy) (+ dres); gamma/beta gradients are not produced (decoder is frozen, main.py:88). */
int ffvc_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, const void* dres, void* dx, void* ws, int B, int HW, int C, int G,
                       int swish, int dtype, void* stream);
/* Same, with the backward statistics already accumulated by the dgrad convolution that produced dy (ffvc_gemm with FFVC_F_GNB_SUMS:
 * sums[B][G][2] fp64): only the apply pass runs (reads dy, x, dres; writes dx).  Reference: taming ResnetBlock backward through
 * Normalize + nonlinearity. */
int ffvc_groupnorm_bwd_sums(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, const void* dres, void* dx, const double* sums, int B, int HW, int C, int G,
                            int swish, int dtype, void* stream);

/* The same two passes with the producer-side quantisation of ffvc_layernorm_fwd_f8 (16-bit tensors): the normalised (+ swish) tensor
 * is the operand of the fp8 3x3 convolution that follows (sums != NULL: moments from the producing GEMM, ws unused), the input gradient
 * is the operand of the dgrad of the convolution that produced x.  y / dx may be NULL when only the fp8 bytes are consumed. */
int ffvc_groupnorm_fwd_f8(const void* x, void* y, void* y8, float* f8_state, int f8_fmt, const float* gamma, const float* beta,
                          float* mean, float* rstd, void* ws, const double* sums, int B, int HW, int C, int G, float eps, int swish,
                          int dtype, void* stream);
int ffvc_groupnorm_bwd_f8(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                          const void* dres, void* dx, void* dx8, float* f8_state, int f8_fmt, void* ws, int B, int HW, int C, int G,
                          int swish, int dtype, void* stream);

/* Row softmax of fp32 scores: p[r,:cols] = softmax(scale*s[r,:cols]); causal: key j visible to query
 * (r % q_len) iff j <= query (cloob.py:510-516).  Columns [cols, ldp) are zero-filled.
 * vitgan.py:91-92; cloob.py:199-200 (nn.MultiheadAttention); taming AttnBlock softmax(dim=2). */
int ffvc_softmax_fwd(const float* s, void* p, int p_dtype, int64_t rows, int cols, int lds, int ldp, float scale,
                     int causal, int q_len, void* stream);
int ffvc_softmax_bwd(const void* p, const float* dp, void* ds, int p_dtype, int64_t rows, int cols, int ldp,
                     int lddp, float scale, void* stream);

/* Fused attention for short sequences (T <= 64, head_dim == 64, bf16): one wave per (batch item, head), scores /
 * softmax / both products in registers.  qkv: [B, T, 3*heads*64] (q | k | v sections, head-major inside each, the
 * packed in_proj output of nn.MultiheadAttention, cloob.py:199-200), out / dout: [B, T, heads*64], dqkv like qkv.
 * Non-causal.  The backward recomputes the probabilities from qkv (nothing but qkv is kept from the forward). */
int ffvc_attn_small_fwd(const void* qkv, void* out, int dtype, int B, int T, int heads, int head_dim, float scale,
                        void* stream);   /* dtype: FFVC_BF16 | FFVC_F16 */
int ffvc_attn_small_bwd(const void* qkv, const void* dout, void* dqkv, int dtype, int B, int T, int heads, int head_dim,
                        float scale, void* stream);

/* Attention for a handful of tokens and ANY head width (VitGAN mapper, vitgan.py:44-97: 16 tokens, 6 heads of 170 channels):
 * one workgroup per (sample, head), q / k / v staged in LDS, fp32 VALU arithmetic, one launch per direction instead of seven
 * batched GEMMs + two softmax passes.  qkv[b, t, which, h, d] sits at b*sb + t*st + which*sk + h*sh + d*sd ELEMENTS, so the
 * projection output is read where it lies: the reference's '(d k h)' interleave (vitgan.py:81-82) is sk = heads, sh = 1,
 * sd = 3*heads.  out: [B, T, out_ld] with heads*head_dim valid columns per row (the pad, if any, is zeroed); dqkv has qkv's
 * layout, columns row_len .. st-1 of every token row zeroed.  dtype FFVC_BF16 | FFVC_F16 | FFVC_F32;
 * ffvc_attn_tiny_supported(T, head_dim): the panels fit the CU's LDS. */
int ffvc_attn_tiny_supported(int T, int head_dim);
int ffvc_attn_tiny_fwd(const void* qkv, void* out, int dtype, int B, int T, int heads, int head_dim, int64_t sb, int64_t st,
                       int64_t sk, int64_t sh, int64_t sd, int64_t out_ld, float scale, void* stream);
int ffvc_attn_tiny_bwd(const void* qkv, const void* dout, void* dqkv, int dtype, int B, int T, int heads, int head_dim,
                       int64_t sb, int64_t st, int64_t sk, int64_t sh, int64_t sd, int64_t out_ld, int64_t row_len, float scale,
                       void* stream);

/* Exact-fp32 attention of a short sequence (T <= 128, head_dim 64), forward only, one launch: the frozen CLIP text tower's 77 causal tokens
 * (cloob.py:199-200,510-516; evaluated every step at main.py:733).  qkv fp32 packed as in ffvc_attn_small_fwd, out fp32 [B, T, heads*64];
 * two-pass softmax in fp32 on the vector ALUs (the matrices are too small for the tiled GEMM + softmax + GEMM sequence it replaces). */
int ffvc_attn_text_fwd(const float* qkv, float* out, int B, int T, int heads, int head_dim, float scale, int causal, void* stream);

/* Flash-style attention for any sequence length, head_dim 64, optional causal mask: the x-transformer mapper's
 * self-attention (transformer.py:11-20 -> x-transformers Decoder, 1024 tokens at cfg4) and ViT-L/14's 257 tokens
 * (cfg5).  Same packed qkv layout as ffvc_attn_small_*.  No score matrix in HBM; causal blocks above the diagonal are
 * skipped.  lse: fp32 [B*heads, T] written by the forward (log2-sum-exp of the scaled scores), read by the backward;
 * delta_ws: fp32 [B*heads, T] scratch of the backward (row sums of dO*O).  out is the forward's result. */
int ffvc_attn_flash_fwd(const void* qkv, void* out, void* lse, int dtype, int B, int T, int heads, int head_dim,
                        float scale, int causal, void* stream);
/* The forward whose output also leaves as fp8 bytes out8 [B, T, heads*64] (producer-side quantisation for the fp8 out_proj of the ViT-L/14
 * tower, cloob.py:199-200; byte contract of ffvc_layernorm_fwd_f8).  `out` is still written: the backward reads it. */
int ffvc_attn_flash_fwd_f8(const void* qkv, void* out, void* lse, void* out8, float* f8_state, int f8_fmt, int dtype, int B, int T,
                           int heads, int head_dim, float scale, int causal, void* stream);
int ffvc_attn_flash_bwd(const void* qkv, const void* out, const void* dout, const void* lse, void* delta_ws, void* dqkv,
                        int dtype, int B, int T, int heads, int head_dim, float scale, int causal, void* stream);

/* ---------------------------------------------------------------------------
 * Glue kernels of the train step (main.py:715-837)
 * ------------------------------------------------------------------------- */
int ffvc_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);
/* Split-precision operand of an fp32-grade GEMM on the 16-bit matrix pipes (the frozen CLIP text tower, reference
 * cloob.py:525-538 runs it in fp32): src fp32 [rows, K] (row stride ld_src) -> dst 16-bit [rows, 3K] holding hi = rn16(x) and
 * lo = rn16(x - hi) as [hi | lo | hi] (weight_order 0: activations) or [hi | hi | lo] (weight_order 1: weights), so that one
 * 16-bit GEMM of depth 3K accumulates every product but lo x lo in fp32 (relative error ~2^-21 with f16 segments). */
int ffvc_split3(const float* src, void* dst, int dst_dtype, int64_t rows, int K, int64_t ld_src, int weight_order, void* stream);
/* dst[b][c][r] = src[b][r][c] (+dtype conversion): einops Rearrange mlp_mixer_pytorch.py:31; W^T shadows for dgrad */
/* dst_ld (0 = rows) >= rows pads every output row with zeros (per-head padding of VitGAN's (d k h) qkv layout,
 * vitgan.py:82) */
int ffvc_transpose(const void* src, int src_dtype, void* dst, int dst_dtype, int batch, int rows, int cols,
                   int64_t src_batch_stride, int64_t dst_batch_stride, int dst_ld, void* stream);

/* n_items independent transposes dst[c][r] = src[r][c] (same element type) in one launch; `items` and `tile_prefix`
 * (tile_prefix[i] = number of 64x64 tiles of items 0..i-1; total_tiles = their sum) are DEVICE arrays the caller builds
 * once.  Used for the W^T shadows of all trainable weights after an optimizer step (arena.refresh). */
typedef struct ffvc_tr_item {
  const void* src;
  void* dst;
  int32_t rows, cols;
} ffvc_tr_item;
int ffvc_transpose_multi(const ffvc_tr_item* items, const int* tile_prefix, int n_items, int total_tiles, int dtype,
                         void* stream);
/* dst[r, c] = c < cols ? src[r, c] : 0, c < dst_cols; independent leading dims (pad / unpad per-head blocks) */
int ffvc_copy2d(const void* src, int src_dtype, int64_t src_ld, void* dst, int dst_dtype, int64_t dst_ld, int64_t rows,
                int cols, int dst_cols, void* stream);
/* out[c] (+)= sum_r x[r,c]: bias gradients of nn.Linear / Conv1d */
int ffvc_colsum(const void* x, int dtype, float* out, int64_t rows, int cols, int64_t ld, int accumulate,
                void* stream);
/* ClampWithGrad (main.py:118-132) applied to u = x*mul + add: used at main.py:763 (mul=1, add=0,
 * scalar codebook min/max) and main.py:142 ((x+1)/2 clamped to [0,1]: mul=.5, add=.5). */
int ffvc_clamp_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, float mul, float add, float lo,
                   float hi, void* stream);
int ffvc_clamp_bwd(const void* x, int x_dtype, const void* g, int g_dtype, void* dx, int64_t n, float mul, float add,
                   float lo, float hi, void* stream);
/* backward of taming Upsample's F.interpolate(scale_factor=2, mode="nearest"): [B,2H,2W,C] -> [B,H,W,C] */
int ffvc_sumpool2x2(const void* src, void* dst, int dtype, int B, int H, int W, int C, void* stream);
/* vector_quantize (main.py:134-138): ||.||^2 of rows; argmin of (xn+cn)-2*dot (first index on ties);
 * gather == one_hot(idx) @ codebook.  ffvc_gather_rows also serves token+positional embedding (cloob.py:526-528). */
int ffvc_rownorm_sq(const float* x, float* out, int64_t rows, int dim, void* stream);
int ffvc_vq_argmin(const float* dot, const float* xnorm, const float* cnorm, int64_t* idx, int64_t rows, int ncodes,
                   int64_t ld, void* stream);
int ffvc_gather_rows(const float* table, const int64_t* idx, const float* pos, int period, void* out, int out_dtype,
                     int64_t rows, int dim, void* stream);
/* out[b,:] = x[b, argmax_t tokens[b,t], :]  (EOT pooling, cloob.py:536) */
int ffvc_eot_gather(const void* x, int x_dtype, const int64_t* tokens, float* out, int B, int L, int dim,
                    void* stream);
/* MakeCutouts pool branch + noise + CLIP mean/std normalisation (main.py:212-229,797), written as the
 * K-major im2col rows of the ViT patch-embedding conv (cloob.py:224,237).  xr: NHWC fp32 [B,H,W,3] in [0,1];
 * noise: NCHW fp32 [cutn*B,3,cut,cut] with facs [cutn*B] (both NULL = noise_fac 0);
 * out: [cutn*B, (cut/patch)^2, 3*patch*patch]. */
int ffvc_cutouts_fwd(const float* xr, const float* noise, const float* facs, void* out, int out_dtype, int B, int H,
                     int W, int cut, int cutn, int patch, float mean_r, float mean_g, float mean_b, float std_r,
                     float std_g, float std_b, void* stream);
int ffvc_cutouts_bwd(const float* xr, const void* gout, int g_dtype, float* dxr, int B, int H, int W, int cut,
                     int cutn, int patch, float std_r, float std_g, float std_b, void* stream);
/* The reference's DEFAULT augmentation chain of MakeCutouts (main.py:164-165,172,178,182,190 + :222-225,797) applied to the
 * pooled image `pooled` [B,3,S,S] fp32 (= ffvc_cutouts_fwd with cutn 1, patch S, mean 0, std 1) in one fused resampling
 * pass: RandomAffine -> RandomPerspective -> ColorJitter(hue, saturation) -> RandomErasing -> + noise -> mean/std ->
 * ViT patch rows.  All random draws are explicit per-cutout parameters (N = cutn*B rows, cut-major like repeat()):
 * pinv [N,9] inverse perspective homography, ainv [N,6] inverse affine (pixel units), cmat [N,9] RGB colour matrix,
 * erase [N,4] int32 rectangle x0,y0,x1,y1 (x1 <= x0: none).  kornia 0.5.10 itself is absent: parity unpinned.
 * The backward scatters into dpooled [B,3,S,S] (zeroed inside); chain it with ffvc_cutouts_bwd(cutn 1, patch S).
 * One launch covers at most 65535 cutouts (cutn * B; one cutout per grid row): FFVC_E_BADARG beyond that — split the batch. */
int ffvc_augment_fwd(const float* pooled, const float* pinv, const float* ainv, const float* cmat, const float* coff /* (N,3) colour offset added after cmat, may be NULL */,
                     const float* cj /* (N,8) kornia ColorJitter parameters [on, brightness, contrast, saturation, hue (turns), order code, -, -] applied after cmat, may be NULL */,
                     const int32_t* erase, const float* noise, const float* facs, void* out, int out_dtype, int B, int S, int S_src, int cutn,
                     int patch, float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream);
/* padding of the homography slot pinv: zeros with grid_sample's one-pixel linear fade when the matrix rotates / shears / projects
 * (RandomPerspective, RandomRotation), plain coordinate clamping when it only scales and shifts (resize, crop, identity) */
/* pooled / coff / cj: the forward's inputs, needed only when cj != NULL (the jitter is not linear; its Jacobian is re-evaluated) */
int ffvc_augment_bwd(const void* gout, int g_dtype, const float* pinv, const float* ainv, const float* cmat,
                     const int32_t* erase, const float* pooled, const float* coff, const float* cj, float* dpooled, int B, int S,
                     int S_src, int cutn, int patch, float std_r, float std_g, float std_b, void* stream);
/* The same launch in the SEQUENTIAL form (round 5): kornia's nn.Sequential (main.py:199,219) runs RandomAffine and RandomPerspective
 * as TWO bilinear resamples, the second reading the first one's output.  Here the value at an output pixel is the homography slot's
 * interpolation (pinv; zeros / fade as above) of the intermediate image I, whose integer pixels are themselves the border-padded
 * affine interpolation of the source, I(p) = bilinear(pooled, clamp(A^-1 p)) (ainv) — evaluated lazily (16 source taps per output
 * pixel), so the intermediate image never exists in memory and the result equals the two-launch form (ffvc_augment_fwd with the
 * affine alone, then with the rest) up to fp32 summation order.  S == S_src. */
int ffvc_augment_seq_fwd(const float* pooled, const float* pinv, const float* ainv, const float* cmat, const float* coff, const float* cj,
                         const int32_t* erase, const float* noise, const float* facs, void* out, int out_dtype, int B, int S, int S_src,
                         int cutn, int patch, float mean_r, float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream);
int ffvc_augment_seq_bwd(const void* gout, int g_dtype, const float* pinv, const float* ainv, const float* cmat,
                         const int32_t* erase, const float* pooled, const float* coff, const float* cj, float* dpooled, int B, int S,
                         int S_src, int cutn, int patch, float std_r, float std_g, float std_b, void* stream);
/* S = side of the cutouts written, S_src = side of the source image `pooled` [B,3,S_src,S_src]: they differ when the chain
 * holds a resize / crop ('R','Re','Cr','Cc' on a pool_size != cut_size or pool=False source, main.py:203-221), which is then
 * part of pinv.  MakeCutouts(interpolate=True) (main.py:226-228): adaptive average pooling of the augmented batch
 * x [N,3,S,S] fp32 (ffvc_augment_fwd with patch S, mean 0, std 1) to So x So, then mean/std and the ViT patch rows. */
int ffvc_avgpool_patches_fwd(const float* x, void* out, int out_dtype, int N, int S, int So, int patch, float mean_r,
                             float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream);
int ffvc_avgpool_patches_bwd(const void* gout, int g_dtype, float* dx, int N, int S, int So, int patch, float std_r,
                             float std_g, float std_b, void* stream);
/* MakeCutouts augmentations that are not one homography (main.py:169,179,181: 'Sh' RandomSharpness, 'Et' RandomElasticTransform,
 * 'Ts' RandomThinPlateSpline; kornia 0.5.10 semantics, csrc/augment_ops.hip): image -> image on the cutout batch x [N,3,S,S] fp32.
 * on[n] == 0 passes sample n through.  grid [N,S,S,2]: NORMALISED sampling coordinates (x, y) of
 * grid_sample(align_corners=False, padding_mode='zeros').  ffvc_tps_grid: tps [N,26] = 5 centres (x,y), 5 kernel weight pairs,
 * affine a0 (2), coefficients of p.x (2), of p.y (2).  ffvc_elastic_grid: noise [N,2,S,S] in [-1,1] -> Gaussian(ksize, sigma,
 * reflect) per component (tmp, disp: [N,2,S,S] scratch) -> grid = clamp(identity + alpha * disp, -1, 1). */
int ffvc_sharpness_fwd(const float* x, const float* factor, const float* on, float* y, int N, int S, void* stream);
int ffvc_sharpness_bwd(const float* g, const float* x, const float* factor, const float* on, float* dx, int N, int S, void* stream);
int ffvc_warp_grid_fwd(const float* x, const float* grid, const float* on, float* y, int N, int S, void* stream);
int ffvc_warp_grid_bwd(const float* g, const float* grid, const float* on, float* dx, int N, int S, void* stream);
int ffvc_tps_grid(const float* tps, float* grid, int N, int S, void* stream);
int ffvc_elastic_grid(const float* noise, float* tmp, float* disp, float* grid, int N, int S, int ksize, float sigma,
                      float alpha_x, float alpha_y, void* stream);
/* Spherical distance loss (main.py:801-811), repeat=1: loss = coef*mean_n 2*asin(|H-E|/2)^2 with
 * H = normalize(feats[n % B]), E = normalize(embed[n]); dembed (may be NULL) <- d loss / d embed. */
int ffvc_spherical_loss(const float* embed, const float* feats, float* rowloss, float* loss, float* dembed, int N,
                        int B, int D, float coef, void* stream);
/* torch.optim.Adam defaults (main.py:591) over a flat fp32 bucket, step is 1-based; optionally refreshes the
 * low-precision weight shadow in the same pass; grad_scale folds 1/world_size or clip_grad_norm (main.py:833-834). */
/* ema (may be NULL): torch_ema's ExponentialMovingAverage.update() folded into the same pass (main.py:520-525,843-844):
 * ema -= ema_weight * (ema - p_new), ema_weight = 1 - min(decay, (1 + n_updates) / (10 + n_updates)). */
int ffvc_adam(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, int64_t n, float lr,
              float beta1, float beta2, float eps, int step, float grad_scale, float* ema, float ema_weight,
              const float* dev_scale, uint32_t* nonfinite_count,
              const float* dev_hyper /* may be NULL; else fp32 [5] on the device = {lr, 1 - beta1^step, sqrt(1 - beta2^step), grad_scale,
                                        ema_weight}, read INSTEAD of the by-value arguments: the launch can sit in a captured hipGraph */,
              void* stream);
/* dev_scale (may be NULL): grad_scale *= dev_scale[0].  nonfinite_count (may be NULL): an element whose scaled gradient is inf /
 * NaN (overflow of a loss-scaled f16 backward) keeps p, m, v, ema unchanged, and the counter is incremented once per wavefront
 * that met one — the host polls it to back the loss scale off; nothing non-finite ever enters the optimizer state. */
/* clip_grad_norm_ (main.py:833-834) without a host round trip: out[0] = min(1, max_norm / (sqrt(sumsq[0])*|grad_scale|
 * + 1e-6)) (feed it to ffvc_adam's dev_scale), out[1] = the total norm. */
int ffvc_clip_coef(const float* sumsq, float max_norm, float grad_scale, float* out, void* stream);
/* out: 16 x uint64 = per XCD x (0..7): out[2x] = s_memtime (shader-clock ticks), out[2x+1] = s_memrealtime (100 MHz ticks), sampled
 * by one-thread workgroups on `stream` (the counters are per XCD; zeros = that XCD ran no sampling block).  Two samples around a
 * region give its average effective engine clock per XCD (bench.py reports the mean next to the roofline fraction). */
int ffvc_clock_sample(uint64_t* out, void* stream);
/* nn.Dropout of the mapper MLPs / attention outputs (mlp_mixer_pytorch.py:20-22; vitgan.py:34-41,114,133):
 * y[i] = (residual ? residual[i] : 0) + (keep(seed, i) ? x[i] / (1 - p) : 0).  The mask is a counter-based hash of
 * (seed, i): the backward pass calls the same function on the gradient with the same seed.  x may alias y. */
int ffvc_dropout(const void* x, int x_dtype, const float* residual, void* y, int y_dtype, int64_t n, float p, uint32_t seed,
                 void* stream);
/* Optional regularisers of the step: l2 = mean(z^2) (main.py:758-762) and tv_loss (main.py:423-428,769-773) on the
 * NHWC fp32 image batch; backward kernels take the upstream scalar gradient from device memory (g[0]). */
int ffvc_mean_sq(const float* x, float* out, int64_t n, void* stream);
int ffvc_mean_sq_bwd(const float* x, const float* g, float* dx, int64_t n, void* stream);
int ffvc_tv_loss_fwd(const float* x, float* out, int B, int H, int W, int C, void* stream);
int ffvc_tv_loss_bwd(const float* x, const float* g, float* dx, int B, int H, int W, int C, void* stream);
/* out[r % period] (+)= sum_c x[r, c]: bias gradient of the token-mixing Conv1d (bias indexed by output row) */
int ffvc_rowsum(const void* x, int dtype, float* out, int64_t rows, int cols, int period, int accumulate, void* stream);
/* dst[r*dst_stride + c] = src[r*src_stride + c] (fp32; src_stride 0 broadcasts one row): class-token row of the
 * ViT token buffer (cloob.py:240-244) */
int ffvc_copy_rows(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows, int cols,
                   void* stream);
/* explicit 3x3/pad-1 im2col of a small-channel NHWC tensor: out[(b,y,x), (kh*3+kw)*C + c] (zero padded to Kp columns);
 * feeds the dgrad of the decoder's conv_out (Cout = 3), taming Decoder.conv_out [upstream] */
int ffvc_im2col3x3(const void* x, int x_dtype, void* out, int out_dtype, int B, int H, int W, int C, int Kp, void* stream);
int ffvc_mul_dev_scalar(const float* x, const float* s, float* y, int64_t n, void* stream); /* y = x * s[0] */
/* y[i] (+)= sum_s slabs[s*n + i]: combine of split-K partial slabs */
int ffvc_slab_reduce(const float* slabs, float* y, int64_t n, int nslab, int accumulate, void* stream);
int ffvc_sumsq(const float* x, float* out, int64_t n, void* stream);            /* out[0] += sum x^2 */
int ffvc_axpby(const float* x, float* y, int64_t n, float a, float b, void* stream); /* y = a*x + b*y */

/* Fused token-mixing MLP of the MLP-Mixer (mlp_mixer_pytorch.py:28,34 inside PreNormResidual :7-14), one launch:
 *   y[b] = W2 @ gelu(W1 @ xn[b] + b1) + b2 + residual[b]     xn: [B,T,D] 16-bit, W1: [O,T], W2: [T,O] (K-major 16-bit
 *   shadows), b1: [O], b2: [T] fp32, residual / y: [B,T,D] fp32.  The hidden activation never leaves the CU.
 * ffvc_tokmix_bwd_hidden recomputes it for the backward pass: h = gelu(W1 xn + b1), dh = (W2^T dy) * gelu'(W1 xn + b1),
 *   written once as [B,O,D] 16-bit (w2t = W2^T as [O,T]); dx = W1^T dh and the weight gradients stay ffvc_gemm calls.
 *   db1 (fp32 [O], may be NULL): += sum over (sample, d) of the stored dh = the first Conv1d's bias gradient.
 * Supported: dtype FFVC_BF16 | FFVC_F16, T in {128, 256}, D % 32 == 0, O % 32 == 0, O <= 4096 (ffvc_tokmix_supported). */
int ffvc_tokmix_supported(int dtype, int T, int D, int O);
int ffvc_tokmix_fwd(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const float* residual,
                    float* y, int dtype, int B, int T, int D, int O, void* stream);
/* ffvc_tokmix_fwd that also writes h = gelu(W1 xn + b1) and gact = gelu'(W1 xn + b1) as [B][O][D] tensors of xn's dtype: the backward
 * pass then takes dh = (W2^T dy) * gact from a plain batched ffvc_gemm (FFVC_F_MUL_ACT_GRAD | FFVC_F_AUX_ACTGRAD) instead of
 * ffvc_tokmix_bwd_hidden's recomputation. */
int ffvc_tokmix_fwd_save(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const float* residual,
                         float* y, void* h, void* gact, int dtype, int B, int T, int D, int O, void* stream);
int ffvc_tokmix_bwd_hidden(const void* xn, const void* dy, const void* w1, const float* b1, const void* w2t, void* h,
                           void* dh, float* db1, int dtype, int B, int T, int D, int O, void* stream);

/* Kernel-selection overrides for tests / A-B measurements (defaults come from FFVC_GEMM2_BM / FFVC_CONV_ROW):
 *   "gemm2_tile": 0 = register-staged kernel only, 1 = heuristic, 128 | 256 | 512 = force the LDS-DMA tile
 *                 (128x128 | 256x128 | 256x256);   "conv_row": 0 off, 1 heuristic, 2 force the haloed row-tile conv;
 *   "gemm8": 1 = every eligible 256x256 launch takes the 8-phase kernel (default: only where it measured faster). */
int ffvc_set_option(const char* name, int value);

/* ---------------------------------------------------------------------------
 * Gradient exchange (hvd.DistributedOptimizer, main.py:626-629): one RCCL communicator per process behind an opaque
 * handle.  RCCL is resolved at run time from the image already loaded in the process (PyTorch's librccl.so; a second copy
 * is never loaded implicitly) — ffvc_rccl_load(path) names one explicitly.  Rank 0 draws the 128-byte unique id and the
 * host hands it to the other ranks (torch.distributed broadcast / store); ffvc_rccl_comm_create is collective.
 * ffvc_allreduce_bucket: in-place SUM over the ranks of buf[0 .. count) (FFVC_F32 gradients, or their FFVC_BF16 / FFVC_F16
 * wire copy), enqueued on `stream` — the host gives it a dedicated exchange stream fenced by events so that neither the
 * dgrad chain nor the weight-gradient stream waits on it; 1/N is folded into ffvc_adam's grad_scale.
 * ------------------------------------------------------------------------- */
int ffvc_rccl_available(void);                /* 0 = no RCCL image in the process, else its version code */
int ffvc_rccl_load(const char* path);
int ffvc_rccl_unique_id(void* out128);
int ffvc_rccl_comm_create(const void* id128, int rank, int world, void** handle);
int ffvc_allreduce_bucket(void* handle, void* buf, int64_t count, int dtype, void* stream);
int ffvc_rccl_comm_destroy(void* handle);

/* Library / device info */
const char* ffvc_last_error(void);
int ffvc_version(void);
/* Writes CU count and clock (kHz) of the current device; used by bench.py for the roofline peak. */
int ffvc_device_info(int32_t* n_cu, int32_t* clock_khz, int64_t* hbm_bytes);

/* Debug probe: dumps the lane->element map of ds_read_b64_tr_b16 (64 lanes x 4 shorts into out[256]). */
int ffvc_probe_tr16(int16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FFVC_H */
