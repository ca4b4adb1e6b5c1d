// gemm_fp8.hip — OCP fp8 GEMM for the FROZEN towers (BASELINE.json configs[4]: "OpenCLIP ViT-L/14 ... fp8 MFMA path"):
//
//     y[m,n] = act( s0*s1 * sum_k X8[m,k] * W8[n,k] + bias ) (+ residual)        X8: e4m3 | e5m2,  W8: e4m3,  fp32 accumulate
//
// gfx950 reaches its fp8 rate (2x bf16) only through v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, block
// scales left at 1); the K=16 fp8 MFMAs of CDNA3 run at the bf16 rate.  The kernel is the LDS-DMA ring kernel of
// gemm2_kernels.h with the operands viewed as 16-bit words: a K step of 128 fp8 values is the same 128-byte row as 64
// halves, so the DMA, the XOR-swizzled LDS image and the fragment reads are reused unchanged; two consecutive 16-byte
// fragment reads (k bytes [32s+16h, +16) for s = 2t, 2t+1) form the 32-byte A / B operand of ONE K=64 MFMA.  The
// hardware pairs A and B bytes by (lane half, byte position), both operands are read through the same map, so every k
// meets its partner exactly once.  Half the operand bytes per FLOP also halves the LDS-DMA issue stream that limits the
// 16-bit kernel.  Epilogues (bias / GELU / QuickGELU / residual / pre-activation write / activation-gradient multiply,
// row-store through the LDS pad) are the shared ones; per-tensor scales arrive as device scalars (delayed scaling).
//
// Quantisation helpers: ffvc_fp8_quant (x*scale -> saturating e4m3 / e5m2, accumulates max|x| for the NEXT step's scale),
// ffvc_fp8_amax, ffvc_fp8_update (scale <- fmt_max / (amax * margin)).
#include "gemm2_kernels.h"

extern "C" int ffvc_actgrad_inplace(void* aux, int dtype, int act, int M, int N, int64_t ld, void* stream);

namespace {

typedef int v8i_t __attribute__((ext_vector_type(8)));

// A = weight rows (always e4m3: cbsz 0), B = activation rows (e4m3: blgp 0, e5m2: blgp 1)
template <int XFMT>
__device__ __forceinline__ void mma_f8(f32x16_t& acc, const u32x4_t& a0, const u32x4_t& a1, const u32x4_t& b0,
                                       const u32x4_t& b1) {
  const v8i_t a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
  const v8i_t b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, XFMT, 0, 0, 0, 0);
}

template <typename L, int BM, int BN, int XFMT, int EPI = ffvc_gemm_detail::EPI_ALL>
__global__ __launch_bounds__(64 * 2 * (BN / 64), (BM == 256 ? 2 : 1)) void gemm_f8_kernel(const ffvc_gemm_desc p, int tiles_n,
                                                                                          int n_tiles, int vec_ok,
                                                                                          const float* __restrict__ s0,
                                                                                          const float* __restrict__ s1) {
  constexpr int MT = BM / 64;
  constexpr int NW = 2 * (BN / 64);
  constexpr int XTILE = BM * 128, WTILE = BN * 128;
  constexpr int STAGE = XTILE + WTILE;
  constexpr bool RING = !(BM == 256 && BN == 128);
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int l31 = lane & 31;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int K16 = p.K >> 1;                                   // the operands as 16-bit words
  KMajorDmaB<BM, NW> sx;
  KMajorDmaB<BN, NW> sw;
  sx.init((const uint16_t*)p.x, p.ldx >> 1, m0, p.M, 0, 0, tid, 0, 0);
  sw.init((const uint16_t*)p.w, p.ldw >> 1, n0, p.N, 0, 0, tid, 0, 0);

  f32x16_t acc[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  const int nk = (K16 + BK - 1) / BK;
  auto compute = [&](const unsigned char* sX, const unsigned char* sW, auto&& between) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      u32x4_t fa0[2], fa1[2], fb0[MT], fb1[MT];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int rw = wn * 64 + t * 32 + l31;
        fa0[t] = frag_kmajor(sW, rw, 2 * t2, lane);
        fa1[t] = frag_kmajor(sW, rw, 2 * t2 + 1, lane);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int rx = wm * (32 * MT) + t * 32 + l31;
        fb0[t] = frag_kmajor(sX, rx, 2 * t2, lane);
        fb1[t] = frag_kmajor(sX, rx, 2 * t2 + 1, lane);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) mma_f8<XFMT>(acc[a][b], fa0[a], fa1[a], fb0[b], fb1[b]);
      between(t2);
    }
  };
  if constexpr (!RING) {
    for (int kt = 0; kt < nk; ++kt) {
      sx.issue(smem, kt * BK, K16);
      sw.issue(smem + XTILE, kt * BK, K16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(smem, smem + XTILE, [](int) {});
      __syncthreads();
    }
  } else {
    if (nk > 0) {
      sx.issue(smem, 0, K16);
      sw.issue(smem + XTILE, 0, K16);
    }
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      unsigned char* cur = smem + (kt & 1) * STAGE;
      unsigned char* nxt = smem + ((kt + 1) & 1) * STAGE;
      const bool more = kt + 1 < nk;
      const int kn = (kt + 1) * BK;
      compute(cur, cur + XTILE, [&](int t2) {
        if (more && t2 == 0) sx.issue(nxt, kn, K16);
        if (more && t2 == 1) sw.issue(nxt + XTILE, kn, K16);
      });
    }
  }
  if (s0) {      // product of the two per-tensor inverse scales (device scalars: delayed scaling never syncs the host)
    const float s = s0[0] * (s1 ? s1[0] : 1.0f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] *= s;
  }
  if (vec_ok == 2)
    ffvc_gemm_detail::gemm_epilogue_rows<L, MT, false, EPI>(p, acc, m0, n0, wm, wn, lane, 0, 0,
                                                                   smem + (RING ? 2 : 1) * STAGE + wid * 4096, 0);
  else
    ffvc_gemm_detail::gemm_epilogue<L, MT, true>(p, acc, m0, n0, wm, wn, lane, 0, 0, 1, 0);
}

// ---- fp8 3x3 convolution of the frozen decoder (round 4; BASELINE configs[4], reference main.py:140-143) ------------------------
// conv_row_kernel (gemm2.hip) with fp8 operands: the haloed X row tile and the filter tile hold 128-byte rows = 128 fp8 channels
// per K step (the DMA address generators see 16-bit words: Cin / 2 words per pixel), the three kw taps are served from the same
// X tile, the MFMAs are v_mfma_f32_32x32x64_f8f6f4 on two 16-byte fragment reads per operand exactly as in gemm_f8_kernel.  Same
// two-workgroups-per-CU single-stage schedule, same epilogues (bias, residual, GroupNorm moments).  Needs Cin % 128 == 0.
template <typename L, int XFMT, int EPI>
__global__ __launch_bounds__(256, 2) void conv_row_f8_kernel(const ffvc_gemm_desc p, int tiles_n, int n_tiles, int vec_ok,
                                                             const float* __restrict__ s0, const float* __restrict__ s1) {
  constexpr int MT = 4, BM = 256, BN = 128;
  constexpr int XTILE = 264 * 128, WTILE = BN * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid & 1, wn = wid >> 1;
  const int l31 = lane & 31;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.conv_W, Cin16 = p.conv_Cin >> 1, K16 = p.K >> 1, ldw16 = (int)(p.ldw >> 1);
  ConvRowDmaB sx;
  KMajorDmaB<BN, 4> sw;
  sx.init((const uint16_t*)p.x, m0, p.conv_H, W, Cin16, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
  sw.init((const uint16_t*)p.w + (int64_t)n0 * ldw16, ldw16, p.N - n0, tid);
  f32x16_t acc[2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
  int xrow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int px = wm * 128 + t * 32 + l31;
    const int Wt = W < 256 ? W : 256;
    xrow[t] = (px / Wt) * (Wt + 2) + (px % Wt);
  }
  unsigned char* sX = smem;
  unsigned char* sW = smem + XTILE;
  const int nblk = Cin16 / 64;
  for (int kh = 0; kh < 3; ++kh) {
    for (int cb = 0; cb < nblk; ++cb) {
#pragma unroll 1
      for (int kw = 0; kw < 3; ++kw) {
        if (kw == 0) sx.issue(sX, kh, cb * 64, nullptr, tid);
        sw.issue(sW, (kh * 3 + kw) * Cin16 + cb * 64, K16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
          u32x4_t fa0[2], fa1[2], fb0[MT], fb1[MT];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int rw = wn * 64 + t * 32 + l31;
            fa0[t] = frag_kmajor(sW, rw, 2 * t2, lane);
            fa1[t] = frag_kmajor(sW, rw, 2 * t2 + 1, lane);
          }
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            fb0[t] = frag_kmajor(sX, xrow[t] + kw, 2 * t2, lane);
            fb1[t] = frag_kmajor(sX, xrow[t] + kw, 2 * t2 + 1, lane);
          }
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) mma_f8<XFMT>(acc[a][b], fa0[a], fa1[a], fb0[b], fb1[b]);
        }
        __syncthreads();
      }
    }
  }
  if (s0) {
    const float s = s0[0] * (s1 ? s1[0] : 1.0f);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] *= s;
  }
  if (vec_ok == 2)
    ffvc_gemm_detail::gemm_epilogue_rows<L, MT, false, EPI>(p, acc, m0, n0, wm, wn, lane, 0, 0, smem + XTILE + WTILE + wid * 4096, 0);
  else
    ffvc_gemm_detail::gemm_epilogue<L, MT, true>(p, acc, m0, n0, wm, wn, lane, 0, 0, 1, 0);
}

template <typename L>
int launch_conv_f8(const ffvc_gemm_desc& d, int x_fmt, int vec_ok, const float* s0, const float* s1, hipStream_t st) {
  using namespace ffvc_gemm_detail;
  const int tiles_n = d.N / 128, n_tiles = (d.M / 256) * tiles_n;
  constexpr int lds = 264 * 128 + 128 * 128 + 4 * 4096;
  const bool gnv = (d.flags & FFVC_F_GN_SUMS) != 0;
  auto go = [&](auto xf, auto epi_tag) -> int {
    constexpr int XF = decltype(xf)::value, EPI = decltype(epi_tag)::value;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)conv_row_f8_kernel<L, XF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = true;
    }
    hipLaunchKernelGGL((conv_row_f8_kernel<L, XF, EPI>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, s0, s1);
    FFVC_LAUNCH_CHECK();
    return 0;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if (x_fmt) return gnv ? go(I1{}, std::integral_constant<int, EPI_GN>{}) : go(I1{}, std::integral_constant<int, EPI_LEAN>{});
  return gnv ? go(I0{}, std::integral_constant<int, EPI_GN>{}) : go(I0{}, std::integral_constant<int, EPI_LEAN>{});
}

// returns 0 (launched), 2 (launched, aux already holds act'(pre)) or an error code
template <typename L, int BM, int BN>
int launch_f8(const ffvc_gemm_desc& d, int x_fmt, int vec_ok, const float* s0, const float* s1, hipStream_t st) {
  using namespace ffvc_gemm_detail;
  const int tiles_m = ceil_div(d.M, BM), tiles_n = ceil_div(d.N, BN);
  const int n_tiles = tiles_m * tiles_n;
  constexpr int nthreads = 64 * 2 * (BN / 64);
  constexpr int lds = ((BM == 256 && BN == 128) ? 1 : 2) * (BM * 128 + BN * 128) + 2 * (BN / 64) * 4096;
  auto go = [&](auto xf, auto epi_tag) -> int {
    constexpr int XF = decltype(xf)::value, EPI = decltype(epi_tag)::value;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)gemm_f8_kernel<L, BM, BN, XF, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = true;
    }
    hipLaunchKernelGGL((gemm_f8_kernel<L, BM, BN, XF, EPI>), dim3(n_tiles), dim3(nthreads), lds, st, d, tiles_n, n_tiles, vec_ok, s0, s1);
    FFVC_LAUNCH_CHECK();
    return 0;
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if constexpr (BM == 256 && BN == 256) {
    // epilogue classes as in gemm2_kernels.h::launch2 (profiles/r02_epilogue_code_size.txt): the tower's launches are
    // forward (e4m3 activations): qkv (16-bit plain), out_proj / c_proj (fp32 + fp32 residual), c_fc (activation + act'(pre));
    // backward (e5m2 gradients): plain 16-bit dgrads and the aux-multiply
    const bool wants_act = d.act != FFVC_ACT_NONE || (d.flags & (FFVC_F_MUL_ACT_GRAD | FFVC_F_WRITE_PREACT | FFVC_F_COLSUM));
    const bool plain_store = !(d.flags & (FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_BIAS_ALONG_M)) && vec_ok == 2;
    if (d.y8_state) {
      // fp8 OUTPUT: the hidden activation (forward, e4m3) / hidden gradient (backward, e5m2) of a frozen MLP leaves the epilogue as
      // the next fp8 GEMM's operand — no 16-bit tensor, no separate quantisation pass
      const bool bwd = d.flags & FFVC_F_MUL_ACT_GRAD;
      const bool plain_out8 = plain_store && !d.residual && !(d.flags & FFVC_F_OUT_F32);
      const bool ok = plain_out8 && (d.flags & FFVC_F_AUX_ACTGRAD) && (d.act == FFVC_ACT_GELU || d.act == FFVC_ACT_QUICKGELU);
      if (ok && x_fmt && bwd && !d.bias && !(d.flags & FFVC_F_WRITE_PREACT) && d.y8_fmt == 1)
        return go(I1{}, std::integral_constant<int, EPI_K_MULAUX | EPI_O_F8E5>{});
      if (ok && !x_fmt && !bwd && d.bias && (d.flags & FFVC_F_WRITE_PREACT) && !(d.flags & FFVC_F_COLSUM) && d.y8_fmt == 0) {
        const int r = d.act == FFVC_ACT_GELU ? go(I0{}, std::integral_constant<int, EPI_K_GELU_FWDG | EPI_O_F8E4>{})
                                             : go(I0{}, std::integral_constant<int, EPI_K_QGELU_FWDG | EPI_O_F8E4>{});
        return r == 0 ? 2 : r;
      }
      ffvc_set_error("ffvc_gemm_fp8: fp8 output (y8_state) is available with the activation forward that stores act'(pre) (e4m3) and "
                     "the aux-multiply backward (e5m2) on the 256x256 tile with 8-aligned rows only");
      return FFVC_E_BADARG;
    }
    if (plain_store && !wants_act) {
      if (!d.residual && !(d.flags & FFVC_F_OUT_F32))
        return x_fmt ? go(I1{}, std::integral_constant<int, EPI_LEAN | EPI_O_T>{}) : go(I0{}, std::integral_constant<int, EPI_LEAN | EPI_O_T>{});
      if (!x_fmt && d.residual && (d.flags & FFVC_F_RES_F32) && (d.flags & FFVC_F_OUT_F32))
        return go(I0{}, std::integral_constant<int, EPI_LEAN | EPI_O_F32R>{});
    }
    const bool plain_out = plain_store && !d.residual && !(d.flags & FFVC_F_OUT_F32);
    if (plain_out && (d.flags & FFVC_F_AUX_ACTGRAD) && (d.act == FFVC_ACT_GELU || d.act == FFVC_ACT_QUICKGELU)) {
      const bool bwd = d.flags & FFVC_F_MUL_ACT_GRAD;
      if (x_fmt && bwd && !d.bias && !(d.flags & FFVC_F_WRITE_PREACT)) return go(I1{}, std::integral_constant<int, EPI_K_MULAUX>{});
      if (!x_fmt && !bwd && d.bias && (d.flags & FFVC_F_WRITE_PREACT) && !(d.flags & FFVC_F_COLSUM)) {
        const int r = d.act == FFVC_ACT_GELU ? go(I0{}, std::integral_constant<int, EPI_K_GELU_FWDG>{})
                                             : go(I0{}, std::integral_constant<int, EPI_K_QGELU_FWDG>{});
        return r == 0 ? 2 : r;
      }
    }
  }
  if (d.y8_state) {
    ffvc_set_error("ffvc_gemm_fp8: fp8 output (y8_state) needs the 256x256 tile");
    return FFVC_E_BADARG;
  }
  return x_fmt ? go(I1{}, std::integral_constant<int, EPI_ALL>{}) : go(I0{}, std::integral_constant<int, EPI_ALL>{});
}

template <typename L>
int launch_f8_cfg(const ffvc_gemm_desc& d, int x_fmt, int vec_ok, const float* s0, const float* s1, hipStream_t st, int cfg) {
  if (cfg == 512) return launch_f8<L, 256, 256>(d, x_fmt, vec_ok, s0, s1, st);
  if (cfg == 256) return launch_f8<L, 256, 128>(d, x_fmt, vec_ok, s0, s1, st);
  return launch_f8<L, 128, 128>(d, x_fmt, vec_ok, s0, s1, st);
}

// ---- quantisation ----------------------------------------------------------------------------------------------------
// state[0] = scale applied before the conversion, state[1] = running max|x| (for the next update), state[2] = 1 / scale
template <int FMT>
__device__ __forceinline__ uint32_t cvt4_f8(float a, float b, float c, float d) {
  constexpr float LIM = FMT == 0 ? 448.0f : 57344.0f;       // e4m3fn / e5m2 finite maxima: saturate instead of inf
  // fminf / fmaxf return the non-NaN operand: a NaN activation or gradient would be quantised to -LIM and the divergence
  // hidden.  NaN goes through unclamped and converts to the format's NaN encoding.
  a = a != a ? a : fminf(fmaxf(a, -LIM), LIM);
  b = b != b ? b : fminf(fmaxf(b, -LIM), LIM);
  c = c != c ? c : fminf(fmaxf(c, -LIM), LIM);
  d = d != d ? d : fminf(fmaxf(d, -LIM), LIM);
  int r = 0;
  if constexpr (FMT == 0) {
    r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  } else {
    r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, r, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
  }
  return (uint32_t)r;
}

// ONE atomic per workgroup (every wave hitting the same address serialises in L2: 32k atomics cost 0.3 ms)
__device__ __forceinline__ void block_amax(float m, float* __restrict__ state) {
  __shared__ float part[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
    if (m > 0.0f) atomicMax((unsigned int*)(state + 1), __float_as_uint(m));   // m >= 0: uint order == float order
  }
}

template <typename ST, int FMT>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const ST* __restrict__ src, uint8_t* __restrict__ dst,
                                                        float* __restrict__ state, int64_t n) {
  const float scale = state[0];
  float m = 0.0f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
    const f32x8 v = load8(src + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v.v[j]));
    u32x2_t o;
    o[0] = cvt4_f8<FMT>(v.v[0] * scale, v.v[1] * scale, v.v[2] * scale, v.v[3] * scale);
    o[1] = cvt4_f8<FMT>(v.v[4] * scale, v.v[5] * scale, v.v[6] * scale, v.v[7] * scale);
    *(u32x2_t*)(dst + i) = o;
  }
  block_amax(m, state);
}

template <typename ST>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const ST* __restrict__ src, float* __restrict__ state, int64_t n) {
  float m = 0.0f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 256 * 8) {
    const f32x8 v = load8(src + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v.v[j]));
  }
  block_amax(m, state);
}

__global__ void fp8_update_kernel(float* __restrict__ state, float fmt_max, float margin) {
  const float a = state[1];
  if (a > 0.0f && isfinite(a)) {
    const float s = fmt_max / (a * margin);
    state[0] = s;
    state[2] = 1.0f / s;
  } else if (!(state[0] > 0.0f)) {
    state[0] = 1.0f;
    state[2] = 1.0f;
  }
  state[1] = 0.0f;
}

// every state of a pool [n][4] whose running amax is set: the update of fp8_update_kernel with the format kept in state[3]
// (0 = e4m3, 1 = e5m2); states that saw no tensor since their last update (amax 0) keep their scale
__global__ __launch_bounds__(256) void fp8_update_many_kernel(float* __restrict__ pool, int n, float margin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float* state = pool + 4 * (int64_t)i;
  const float a = state[1];
  if (a == 0.0f) return;
  const float fmt_max = state[3] == 0.0f ? 448.0f : 57344.0f;
  if (a > 0.0f && isfinite(a)) {
    const float s = fmt_max / (a * margin);
    state[0] = s;
    state[2] = 1.0f / s;
  }
  state[1] = 0.0f;
}

inline int f8_grid(int64_t n) {
  const int64_t g = (n / 8 + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int ffvc_gemm_fp8(const ffvc_gemm_desc* dp, int x_fmt, int lo_dtype, const float* scale0, const float* scale1,
                             void* stream) {
  FFVC_CHECK_ARG(dp != nullptr, "ffvc_gemm_fp8: null descriptor");
  const ffvc_gemm_desc& d = *dp;
  FFVC_CHECK_ARG(d.x && d.w && d.y && d.M > 0 && d.N > 0 && d.K > 0, "ffvc_gemm_fp8: bad problem");
  FFVC_CHECK_ARG(x_fmt == 0 || x_fmt == 1, "ffvc_gemm_fp8: x_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(lo_dtype == FFVC_BF16 || lo_dtype == FFVC_F16, "ffvc_gemm_fp8: lo_dtype must be a 16-bit storage type");
  if (d.x_mode == FFVC_OP_CONV3X3) {
    // fp8 3x3 convolution (frozen decoder): the haloed-row kernel's geometry, 128-channel K steps
    const int W = d.conv_W, Cin = d.conv_Cin;
    FFVC_CHECK_ARG(d.w_mode == FFVC_OP_KMAJOR && d.batch <= 1 && d.split_k <= 1 && d.slab_stride == 0 && d.act == FFVC_ACT_NONE &&
                       !(d.flags & (FFVC_F_MUL_ACT_GRAD | FFVC_F_WRITE_PREACT | FFVC_F_COLSUM | FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_OUT_F32)),
                   "ffvc_gemm_fp8 (conv): plain 16-bit output, optional bias / residual / GroupNorm moments only");
    FFVC_CHECK_ARG((W == 64 || W == 128 || (W >= 256 && W % 256 == 0)) && ((int64_t)d.conv_H * W) % 256 == 0 && (d.N % 128) == 0 &&
                       (d.M % 256) == 0 && Cin > 0 && (Cin % 128) == 0 && d.K == 9 * Cin && d.ldw == d.K,
                   "ffvc_gemm_fp8 (conv): needs W in {64, 128, 256 k}, H*W %% 256 == 0, Cout %% 128 == 0, Cin %% 128 == 0 (W=%d Cin=%d N=%d)", W, Cin, d.N);
    FFVC_CHECK_ARG(((uintptr_t)d.x % 16) == 0 && ((uintptr_t)d.w % 16) == 0 && ((uintptr_t)d.y % 16) == 0 && d.y_mi == 0 && d.y_sm == d.N &&
                       (!d.residual || (d.r_mi == 0 && d.r_sm == d.N && !(d.flags & FFVC_F_RES_F32) && ((uintptr_t)d.residual % 16) == 0)) &&
                       (!d.bias || ((uintptr_t)d.bias % 16) == 0),
                   "ffvc_gemm_fp8 (conv): 16-byte aligned NHWC tensors expected");
    const int ups = (d.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0;
    FFVC_CHECK_ARG(((int64_t)(d.M >> (2 * ups)) * Cin) < 0x7FFFFF00ll && (int64_t)d.N * d.K < 0x7FFFFF00ll, "ffvc_gemm_fp8 (conv): operand beyond 32-bit DMA offsets");
    FFVC_CHECK_ARG(!(d.flags & FFVC_F_GN_SUMS) || (d.gn_sums && d.gn_hw > 0 && (d.gn_hw % 256) == 0 && d.gn_cpg >= 4 && (d.gn_cpg % 4) == 0),
                   "ffvc_gemm_fp8 (conv): bad GroupNorm-moment request");
    hipStream_t stc = (hipStream_t)stream;
    return lo_dtype == FFVC_F16 ? launch_conv_f8<f16_t>(d, x_fmt, 2, scale0, scale1, stc) : launch_conv_f8<uint16_t>(d, x_fmt, 2, scale0, scale1, stc);
  }
  FFVC_CHECK_ARG(d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_KMAJOR && d.batch <= 1 && d.split_k <= 1 && d.kseg == 0 &&
                     d.x_mi == 0 && d.slab_stride == 0,
                 "ffvc_gemm_fp8: plain K-major x K-major problems only");
  FFVC_CHECK_ARG((d.K % 16) == 0 && (d.ldx % 16) == 0 && (d.ldw % 16) == 0 && ((uintptr_t)d.x % 16) == 0 &&
                     ((uintptr_t)d.w % 16) == 0,
                 "ffvc_gemm_fp8: K / ldx / ldw must be multiples of 16 and the operands 16-byte aligned (K=%d)", d.K);
  FFVC_CHECK_ARG(!(d.flags & (FFVC_F_GN_SUMS | FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_TR_SAFE)),
                 "ffvc_gemm_fp8: unsupported flags 0x%x", d.flags);
  FFVC_CHECK_ARG((256 * (d.ldx / 2) + d.K / 2) * 2 < 0x7FFFFF00ll && (256 * (d.ldw / 2) + d.K / 2) * 2 < 0x7FFFFF00ll,
                 "ffvc_gemm_fp8: operand rows too long for 32-bit DMA offsets");
  FFVC_CHECK_ARG(!(d.flags & FFVC_F_COLSUM) || d.colsum, "ffvc_gemm_fp8: FFVC_F_COLSUM without a buffer");
  // vectorised epilogue requirements (as ffvc_gemm); the row-store epilogue when everything is 8-aligned
  auto mult = [](int64_t v, int64_t m) { return (v % m) == 0; };
  const bool out32 = d.flags & FFVC_F_OUT_F32;
  int vec_ok = mult(d.N, 4) && mult(d.y_sm, 4) && mult((int64_t)(uintptr_t)d.y, 16);
  if (d.residual) vec_ok = vec_ok && mult(d.r_sm, 4) && mult((int64_t)(uintptr_t)d.residual, 16);
  if (d.aux) vec_ok = vec_ok && mult(d.ldaux, 4) && mult((int64_t)(uintptr_t)d.aux, 16);
  if (d.bias) vec_ok = vec_ok && mult((int64_t)(uintptr_t)d.bias, 16);
  FFVC_CHECK_ARG(vec_ok && d.y_mi == 0 && d.r_mi == 0, "ffvc_gemm_fp8: needs N %% 4 == 0 and 16-byte aligned plain output rows");
  {
    bool ok = mult(d.N, 8) && (out32 ? mult(d.y_sm, 4) : mult(d.y_sm, 8));
    if (d.residual) ok = ok && ((d.flags & FFVC_F_RES_F32) ? mult(d.r_sm, 4) : mult(d.r_sm, 8));
    if (d.aux) ok = ok && mult(d.ldaux, 8);
    if (ok) vec_ok = 2;
  }
  FFVC_CHECK_ARG(!(d.flags & FFVC_F_COLSUM) || vec_ok == 2, "ffvc_gemm_fp8: FFVC_F_COLSUM needs the row-store epilogue");
  // tile choice as in ffvc_gemm2_try (every variant keeps 8 waves on a CU; pick the largest tile that fills whole rounds)
  static int env_bm = -1;
  if (env_bm < 0) {
    const char* e = getenv("FFVC_FP8_BM");
    env_bm = e ? atoi(e) : 0;
  }
  int cfg = (env_bm == 128 || env_bm == 256 || env_bm == 512) ? env_bm : 0;
  if (!cfg) {
    const int64_t t512 = (int64_t)ceil_div(d.M, 256) * ceil_div(d.N, 256);
    const int64_t t256 = (int64_t)ceil_div(d.M, 256) * ceil_div(d.N, 128);
    const int64_t t128 = (int64_t)ceil_div(d.M, 128) * ceil_div(d.N, 128);
    auto eff = [](int64_t t, int slots) { return (double)t / (double)(((t + slots - 1) / slots) * slots); };
    const double e512 = (d.N >= 256 && t512 >= 192) ? eff(t512, 256) * 1.25 : 0.0;
    const double e256 = t256 >= 256 ? eff(t256, 512) * 1.07 : 0.0;
    const double e128 = eff(t128, 512);
    cfg = (e512 >= e256 && e512 >= e128) ? 512 : (e256 >= e128 ? 256 : 128);
  }
  if (d.y8_state) {
    FFVC_CHECK_ARG(vec_ok == 2 && (d.y8_fmt == 0 || d.y8_fmt == 1), "ffvc_gemm_fp8: fp8 output needs 8-aligned rows and y8_fmt 0 | 1");
    cfg = 512;                               // the specialised MLP kinds exist on the 256x256 tile
  }
  hipStream_t st = (hipStream_t)stream;
  if (d.flags & FFVC_F_AUX_ACTGRAD)
    FFVC_CHECK_ARG(d.aux && (d.act == FFVC_ACT_GELU || d.act == FFVC_ACT_QUICKGELU),
                   "ffvc_gemm_fp8: FFVC_F_AUX_ACTGRAD needs aux and GELU / QuickGELU");
  const int rc = lo_dtype == FFVC_F16 ? launch_f8_cfg<f16_t>(d, x_fmt, vec_ok, scale0, scale1, st, cfg)
                                      : launch_f8_cfg<uint16_t>(d, x_fmt, vec_ok, scale0, scale1, st, cfg);
  if (rc == 2) return 0;
  if (rc) return rc;
  // FFVC_F_AUX_ACTGRAD forward on a kernel without the specialised epilogue: aux holds the pre-activation -> convert it
  if ((d.flags & FFVC_F_AUX_ACTGRAD) && (d.flags & FFVC_F_WRITE_PREACT) && !(d.flags & FFVC_F_MUL_ACT_GRAD))
    return ffvc_actgrad_inplace(d.aux, lo_dtype, d.act, d.M, d.N, d.ldaux, stream);
  return 0;
}

extern "C" int ffvc_fp8_quant(const void* src, int src_dtype, void* dst, int fmt, float* state, int64_t n, void* stream) {
  FFVC_CHECK_ARG(src && dst && state && n > 0 && (n % 8) == 0, "ffvc_fp8_quant: bad args (n must be a multiple of 8)");
  FFVC_CHECK_ARG(fmt == 0 || fmt == 1, "ffvc_fp8_quant: fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "ffvc_fp8_quant: misaligned pointers");
  hipStream_t st = (hipStream_t)stream;
  if (fmt == 0) {
    DISPATCH_DT(src_dtype, ST, hipLaunchKernelGGL((fp8_quant_kernel<ST, 0>), dim3(f8_grid(n)), dim3(256), 0, st, (const ST*)src,
                                                  (uint8_t*)dst, state, n));
  } else {
    DISPATCH_DT(src_dtype, ST, hipLaunchKernelGGL((fp8_quant_kernel<ST, 1>), dim3(f8_grid(n)), dim3(256), 0, st, (const ST*)src,
                                                  (uint8_t*)dst, state, n));
  }
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_fp8_amax(const void* src, int src_dtype, float* state, int64_t n, void* stream) {
  FFVC_CHECK_ARG(src && state && n > 0 && (n % 8) == 0, "ffvc_fp8_amax: bad args (n must be a multiple of 8)");
  FFVC_CHECK_ARG(((uintptr_t)src % 16) == 0, "ffvc_fp8_amax: misaligned pointer");
  DISPATCH_DT(src_dtype, ST, hipLaunchKernelGGL((fp8_amax_kernel<ST>), dim3(f8_grid(n)), dim3(256), 0, (hipStream_t)stream,
                                                (const ST*)src, state, n));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_fp8_update(float* state, int fmt, float margin, void* stream) {
  FFVC_CHECK_ARG(state && (fmt == 0 || fmt == 1) && margin >= 1.0f, "ffvc_fp8_update: bad args");
  hipLaunchKernelGGL(fp8_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, fmt == 0 ? 448.0f : 57344.0f, margin);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_fp8_update_many(float* pool, int n, float margin, void* stream) {
  FFVC_CHECK_ARG(pool && n > 0 && margin >= 1.0f, "ffvc_fp8_update_many: bad args");
  hipLaunchKernelGGL(fp8_update_many_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pool, n, margin);
  FFVC_LAUNCH_CHECK();
  return 0;
}
