// gemm_fp8_skinny.hip — fp8 GEMM for a HANDFUL of rows (round 4): y[M <= 64, N] = sx * sw * X8[M, K] W8[N, K]^T (+ bias) (+ residual).
//
// Why: the ViT-L/14 tower at 64 cutouts has 64 x 257 = 16448 rows = 64 whole 256-row tiles + 64 rows.  Inside one launch those 64 rows cost
// a nearly empty extra round whatever the tile (ffvc_gemm_fp8: 260 tiles of 256 x 256 on 256 CUs); as their own launch on the tiled kernel
// they cost 13-26 us of K-loop latency (one workgroup per output tile walks K = 1024 ... 4096 alone; tools/fp8_rows_bench.py).  Here the
// K loop is split across the eight waves of a workgroup instead: a workgroup owns 32 output columns, every wave multiplies its eighth of K
// (v_mfma_f32_32x32x64_f8f6f4, operands straight from global memory: both are K-major, a lane reads 32 contiguous bytes of its row per
// MFMA), the eight partial 64 x 32 tiles meet in LDS and the sum gets scales, bias and residual.  The k order inside an MFMA is whatever
// the hardware makes of (lane half, byte position): A and B are read through the same map, so the products pair up (as in gemm_fp8.hip).
// The reference has no such operator: this is plumbing under cloob.py:199-205's linears for BASELINE configs[4].
#include "common.h"

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));

template <int XFMT>
__device__ __forceinline__ void mma_f8s(f32x16_t& acc, const v8i_t& a, const v8i_t& b) {
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, XFMT, 0, 0, 0, 0);
}

__device__ __forceinline__ v8i_t ld32(const uint8_t* p) {
  const u32x4_t lo = *(const u32x4_t*)p, hi = *(const u32x4_t*)(p + 16);
  return v8i_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

// grid: N / 32 workgroups of 512 threads.  K % 512 == 0 (eight waves x whole 64-byte MFMA steps), N % 32 == 0, M <= 64.
template <typename YT, int XFMT>
__global__ __launch_bounds__(512) void gemm_f8_skinny_kernel(const uint8_t* __restrict__ x, const uint8_t* __restrict__ w, YT* __restrict__ y,
                                                             const float* __restrict__ bias, const void* __restrict__ residual, int res_f32,
                                                             int M, int N, int K, const float* __restrict__ s0,
                                                             const float* __restrict__ s1) {
  extern __shared__ __attribute__((aligned(16))) float red[];           // [8 waves][32 registers][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int kper = K >> 3;
  const int64_t koff = (int64_t)wave * kper + 32 * h;
  const uint8_t* wp = w + (int64_t)(n0 + r) * K + koff;
  const bool ok0 = r < M, ok1 = 32 + r < M;
  const uint8_t* xp0 = x + (int64_t)(ok0 ? r : 0) * K + koff;
  const uint8_t* xp1 = x + (int64_t)(ok1 ? 32 + r : 0) * K + koff;
  f32x16_t acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
  const v8i_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
  for (int k = 0; k < kper; k += 64) {
    const v8i_t a = ld32(wp + k);
    const v8i_t b0 = ok0 ? ld32(xp0 + k) : zero;
    const v8i_t b1 = ok1 ? ld32(xp1 + k) : zero;
    mma_f8s<XFMT>(acc0, a, b0);
    mma_f8s<XFMT>(acc1, a, b1);
  }
  float* mine = red + (size_t)wave * 32 * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    mine[i * 64 + lane] = acc0[i];
    mine[(16 + i) * 64 + lane] = acc1[i];
  }
  __syncthreads();
  const float s = s0[0] * (s1 ? s1[0] : 1.0f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pos = tid + 512 * j;                   // (register, lane) of the 64 x 32 tile
    float v = 0.0f;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) v += red[wv * 2048 + pos];
    const int reg = pos >> 6, ln = pos & 63;
    const int i = reg & 15;
    const int m = 32 * (reg >> 4) + (ln & 31);                       // accumulator column = activation row
    const int n = n0 + 8 * (i >> 2) + 4 * (ln >> 5) + (i & 3);       // accumulator row = weight row = output column
    if (m >= M) continue;
    v *= s;
    if (bias) v += bias[n];
    const int64_t off = (int64_t)m * N + n;
    if (residual) v += res_f32 ? ((const float*)residual)[off] : ElemTraits<YT>::load((const YT*)residual + off);
    ElemTraits<YT>::store(y + off, v);
  }
}

template <typename YT>
int launch_skinny(const void* x8, const void* w8, void* y, const float* bias, const void* residual, int res_f32, int M, int N, int K,
                  int x_fmt, const float* s0, const float* s1, hipStream_t st) {
  constexpr int lds = 8 * 32 * 64 * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {     // 64 KiB of dynamic LDS: ask for it explicitly (once per instantiation)
    (void)hipFuncSetAttribute((const void*)gemm_f8_skinny_kernel<YT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)gemm_f8_skinny_kernel<YT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  if (x_fmt == 0)
    hipLaunchKernelGGL((gemm_f8_skinny_kernel<YT, 0>), dim3(N / 32), dim3(512), lds, st, (const uint8_t*)x8, (const uint8_t*)w8, (YT*)y, bias,
                       residual, res_f32, M, N, K, s0, s1);
  else
    hipLaunchKernelGGL((gemm_f8_skinny_kernel<YT, 1>), dim3(N / 32), dim3(512), lds, st, (const uint8_t*)x8, (const uint8_t*)w8, (YT*)y, bias,
                       residual, res_f32, M, N, K, s0, s1);
  FFVC_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int ffvc_gemm_fp8_skinny_ok(int M, int N, int K) { return M >= 1 && M <= 64 && N >= 32 && (N % 32) == 0 && K >= 512 && (K % 512) == 0; }

// y[M, N] (y_dtype: fp32 | f16 | bf16, row stride N) = scale0 * scale1 * X8[M, K] W8[N, K]^T (+ bias[N]) (+ residual[M, N], res_dtype fp32
// or y's 16-bit type).  x8 / w8: fp8 bytes, K-major, row stride K (x in x_fmt: 0 e4m3 | 1 e5m2; w e4m3).  Shapes: ffvc_gemm_fp8_skinny_ok.
extern "C" int ffvc_gemm_fp8_skinny(const void* x8, const void* w8, void* y, int y_dtype, const float* bias, const void* residual,
                                    int res_dtype, int M, int N, int K, int x_fmt, const float* scale0, const float* scale1, void* stream) {
  FFVC_CHECK_ARG(x8 && w8 && y && scale0, "ffvc_gemm_fp8_skinny: null pointer");
  FFVC_CHECK_ARG(ffvc_gemm_fp8_skinny_ok(M, N, K), "ffvc_gemm_fp8_skinny: M=%d N=%d K=%d unsupported (M <= 64, N %% 32 == 0, K %% 512 == 0)", M, N, K);
  FFVC_CHECK_ARG(x_fmt == 0 || x_fmt == 1, "ffvc_gemm_fp8_skinny: x_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(((uintptr_t)x8 % 16) == 0 && ((uintptr_t)w8 % 16) == 0, "ffvc_gemm_fp8_skinny: misaligned operands");
  FFVC_CHECK_ARG(!residual || res_dtype == FFVC_F32 || res_dtype == y_dtype, "ffvc_gemm_fp8_skinny: residual must be fp32 or y's dtype");
  hipStream_t st = (hipStream_t)stream;
  const int rf = residual && res_dtype == FFVC_F32 ? 1 : 0;
  if (y_dtype == FFVC_F32) return launch_skinny<float>(x8, w8, y, bias, residual, 1, M, N, K, x_fmt, scale0, scale1, st);
  if (y_dtype == FFVC_F16) return launch_skinny<f16_t>(x8, w8, y, bias, residual, rf, M, N, K, x_fmt, scale0, scale1, st);
  if (y_dtype == FFVC_BF16) return launch_skinny<uint16_t>(x8, w8, y, bias, residual, rf, M, N, K, x_fmt, scale0, scale1, st);
  ffvc_set_error("ffvc_gemm_fp8_skinny: y_dtype %d unsupported", y_dtype);
  return FFVC_E_BADARG;
}
