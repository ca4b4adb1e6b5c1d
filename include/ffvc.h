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
 *                  FFVC_F32  = 1 (exact fp32 "parity mode": fp32-input MFMA).
 */
#ifndef FFVC_H
#define FFVC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFVC_BF16 0
#define FFVC_F32 1

#define FFVC_E_BADARG (-1)
#define FFVC_E_UNSUPPORTED (-2)

/* activation codes for the GEMM epilogue */
#define FFVC_ACT_NONE 0
#define FFVC_ACT_GELU 1      /* exact erf GELU: mlp_mixer_pytorch.py:19, vitgan.py:33 */
#define FFVC_ACT_QUICKGELU 2 /* x*sigmoid(1.702x): cloob.py:179-181 */

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
 *   X KMAJOR: x[ xb(z) + m*ldx + (k/kseg)*xkso + (k%kseg) ]      TRANS: x[ xb(z) + k*ldx + m ]
 *   W KMAJOR: w[ wb(z) + n*ldw + (k/kseg)*wkso + (k%kseg) ]      TRANS: w[ wb(z) + k*ldw + n ]
 *   y       : y[ yb(z) + (m/y_mi)*y_so + (m%y_mi)*y_sm + n ]
 *   residual: r[ rb(z) + (m/r_mi)*r_so + (m%r_mi)*r_sm + n ]
 *   aux     : aux[ ab(z) + m*ldaux + n ]
 *   batch   : ?b(z) = (z / batch_inner) * ?bo + (z % batch_inner) * ?bi
 *   CONV3X3 : X is NHWC [B, Hin, Win, Cin]; m = (b*H + oy)*W + ox over the OUTPUT grid
 *             (H = Hin, or 2*Hin with FFVC_F_UPSAMPLE2X); k = (kh*3+kw)*Cin + ci; K = 9*Cin;
 *             W is [Cout][3][3][Cin] (KMAJOR).  Cin must be a multiple of 64 (bf16) / 32 (f32).
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
} ffvc_gemm_desc;

int ffvc_gemm(const ffvc_gemm_desc* d, void* stream);

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
