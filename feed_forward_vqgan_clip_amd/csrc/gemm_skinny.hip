// gemm_skinny.hip — 16-bit GEMM for a HANDFUL of rows (round 4): y[M <= 64, N] = X[M, K] W[N, K]^T (+ bias) (+ residual), f16 / bf16 operands.
//
// The 16-bit sibling of gemm_fp8_skinny.hip, for the same reason: the ViT-L/14 tower at 64 cutouts has 64 x 257 = 16448 rows = 64 whole
// 256-row tiles + 64 rows, and the 64 rows cost the tiled kernels a nearly empty extra round (tools/f16_rows_bench.py: 16448 vs 16384 rows
// 131 vs 108 us at N=3072 K=1024, 62 vs 36 at N=1024 K=1024, 152 vs 108 at N=1024 K=4096) or, as their own tiled launch, 16-30 us of K-loop
// latency.  Here a workgroup owns 32 output columns and its eight waves split K (v_mfma_f32_32x32x16 on operands read straight from
// global memory: both are K-major, a lane reads 32 contiguous bytes of its row = its share of two MFMAs), the partial 64 x 32 tiles meet in
// LDS, the sum gets bias and residual.  A and B are read through the same (lane half, position) -> k map, so the products pair up.
#include "common.h"

namespace {

__device__ __forceinline__ void ld32h(const uint16_t* p, u32x4_t& lo, u32x4_t& hi) {
  lo = *(const u32x4_t*)p;
  hi = *(const u32x4_t*)(p + 8);
}

// grid: N / 32 workgroups of 512 threads.  K % 256 == 0 (eight waves x whole 32-deep double-MFMA steps), N % 32 == 0, M <= 64.
template <typename L, typename YT>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, YT* __restrict__ y,
                                                          const float* __restrict__ bias, const void* __restrict__ residual, int res_f32,
                                                          int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float red[];           // [8 waves][32 registers][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int kper = K >> 3;
  const int64_t koff = (int64_t)wave * kper + 16 * h;
  const uint16_t* wp = w + (int64_t)(n0 + r) * K + koff;
  const bool ok0 = r < M, ok1 = 32 + r < M;
  const uint16_t* xp0 = x + (int64_t)(ok0 ? r : 0) * K + koff;
  const uint16_t* xp1 = x + (int64_t)(ok1 ? 32 + r : 0) * K + koff;
  f32x16_t acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.0f;
  const u32x4_t zero = {0u, 0u, 0u, 0u};
#pragma unroll 2
  for (int k = 0; k < kper; k += 32) {
    u32x4_t a0, a1, b00 = zero, b01 = zero, b10 = zero, b11 = zero;
    ld32h(wp + k, a0, a1);
    if (ok0) ld32h(xp0 + k, b00, b01);
    if (ok1) ld32h(xp1 + k, b10, b11);
    mma_lo<L>(acc0, a0, b00);
    mma_lo<L>(acc0, a1, b01);
    mma_lo<L>(acc1, a0, b10);
    mma_lo<L>(acc1, a1, b11);
  }
  float* mine = red + (size_t)wave * 32 * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    mine[i * 64 + lane] = acc0[i];
    mine[(16 + i) * 64 + lane] = acc1[i];
  }
  __syncthreads();
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
    if (bias) v += bias[n];
    const int64_t off = (int64_t)m * N + n;
    if (residual) v += res_f32 ? ((const float*)residual)[off] : ElemTraits<L>::load((const L*)residual + off);
    ElemTraits<YT>::store(y + off, v);
  }
}

template <typename L, typename YT>
int launch_skinny16(const void* x, const void* w, void* y, const float* bias, const void* residual, int res_f32, int M, int N, int K,
                    hipStream_t st) {
  constexpr int lds = 8 * 32 * 64 * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {     // 64 KiB of dynamic LDS: ask for it explicitly (once per instantiation)
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<L, YT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_skinny_kernel<L, YT>), dim3(N / 32), dim3(512), lds, st, (const uint16_t*)x, (const uint16_t*)w, (YT*)y, bias,
                     residual, res_f32, M, N, K);
  FFVC_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int ffvc_gemm_skinny_ok(int M, int N, int K) { return M >= 1 && M <= 64 && N >= 32 && (N % 32) == 0 && K >= 256 && (K % 256) == 0; }

// y[M, N] (y_dtype: fp32 or in_dtype, row stride N) = X[M, K] W[N, K]^T (+ bias[N]) (+ residual[M, N], res_dtype fp32 or in_dtype).
// x / w: f16 or bf16 (in_dtype), K-major, row stride K.  Shapes: ffvc_gemm_skinny_ok.
extern "C" int ffvc_gemm_skinny(const void* x, const void* w, int in_dtype, void* y, int y_dtype, const float* bias, const void* residual,
                                int res_dtype, int M, int N, int K, void* stream) {
  FFVC_CHECK_ARG(x && w && y, "ffvc_gemm_skinny: null pointer");
  FFVC_CHECK_ARG(in_dtype == FFVC_F16 || in_dtype == FFVC_BF16, "ffvc_gemm_skinny: 16-bit operands only");
  FFVC_CHECK_ARG(ffvc_gemm_skinny_ok(M, N, K), "ffvc_gemm_skinny: M=%d N=%d K=%d unsupported (M <= 64, N %% 32 == 0, K %% 256 == 0)", M, N, K);
  FFVC_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, "ffvc_gemm_skinny: misaligned operands");
  FFVC_CHECK_ARG(y_dtype == FFVC_F32 || y_dtype == in_dtype, "ffvc_gemm_skinny: y must be fp32 or the operands' dtype");
  FFVC_CHECK_ARG(!residual || res_dtype == FFVC_F32 || res_dtype == in_dtype, "ffvc_gemm_skinny: residual must be fp32 or the operands' dtype");
  hipStream_t st = (hipStream_t)stream;
  const int rf = residual && res_dtype == FFVC_F32 ? 1 : 0;
  if (in_dtype == FFVC_F16) {
    if (y_dtype == FFVC_F32) return launch_skinny16<f16_t, float>(x, w, y, bias, residual, rf, M, N, K, st);
    return launch_skinny16<f16_t, f16_t>(x, w, y, bias, residual, rf, M, N, K, st);
  }
  if (y_dtype == FFVC_F32) return launch_skinny16<uint16_t, float>(x, w, y, bias, residual, rf, M, N, K, st);
  return launch_skinny16<uint16_t, uint16_t>(x, w, y, bias, residual, rf, M, N, K, st);
}
