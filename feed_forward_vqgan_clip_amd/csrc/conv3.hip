// conv3.hip — 3x3 convolution on the haloed row tile, software-pipelined, TWO workgroups per CU (round 6).
//
// conv_row_kernel (gemm2.hip) keeps ONE stage in LDS: every K step issues its DMA, waits for the data, passes two barriers and
// only then multiplies; the second workgroup on the CU is supposed to cover that and does so only partly (36-44 % MFMA-busy on
// the 128-channel levels, 834-900 TFLOP/s where a plain GEMM of the same FLOPs does 1200+).  conv_row2_kernel (round 3) pipelined
// the K steps but needed the whole CU (114 KiB of LDS, 512 registers) and lost more to its exposed prologue / epilogue than the
// pipeline gained.  This kernel is the pipeline at HALF the stage depth, so that two workgroups fit a CU and each other's
// prologue / epilogue / barrier waits are covered:
//   * K step = one tap kw of one (kernel row kh, 32-channel block): 32 v_mfma_f32_16x16x32 per wave (wave tile 128 pixels x 64
//     channels, workgroup tile 256 pixels x 128 channels as before);
//   * X tile of a (kh, 32-channel block) = the 256 / Wt image rows with one halo pixel either side, <= 264 rows x 64 B = 17 KiB,
//     DOUBLE-buffered: it serves the three kw taps through fragment reads shifted by kw rows and is replaced once per three steps —
//     the next group's 17 pieces stream in under the current group's kw = 0 / kw = 1 steps;
//   * filter tile of a step = 128 rows x 64 B = 8 KiB in a ring of THREE (slot = kw): stage s + 2 is issued during step s;
//   * ONE barrier per step with a counted vmcnt (only the pieces issued in the step's own first half may still be in flight),
//     fragments of step s + 1 fetched under the second half of step s (two register sets);
//   * register-exchange epilogue (gemm_epilogue_perm16): no LDS pads.  LDS: 2 x 17 + 3 x 8 = 58 KiB per workgroup.
// LDS image: 16-byte chunk c of tile row r at r * 64 + ((c ^ swz4(r >> 2)) << 4) (conflict-free 16-row ds_read_b128, as gemm3.hip).
#include "gemm2_kernels.h"

namespace {

__device__ __forceinline__ int c3_swz4(int g) { return (0x78 >> (2 * (g & 3))) & 3; }   // 0,2,3,1

constexpr int C3_XT = 17 * 1024, C3_WT = 128 * 64;          // bytes: X tile (17 pieces of 16 rows), filter stage
constexpr int C3_WOFF = 2 * C3_XT;
constexpr int C3_LDS = 2 * C3_XT + 3 * C3_WT;               // 59392

template <typename L, int EPI, int EXP = 0>
__global__ __launch_bounds__(256, 2) void conv_row3_kernel(const ffvc_gemm_desc p, int tiles_n, int n_tiles) {
  bool in_loop = false;       // (the experiment switches act inside the K loop only)
  // EXP (FFVC_C3_EXP, timing experiments only, WRONG results; f16 lean instantiations): 1 = no X-tile DMA inside the loop, 2 = no filter DMA inside the
  // loop, 4 = no barrier inside the loop, 8 = no fragment reads inside the loop
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;
  const int l15 = lane & 15, g4 = lane >> 4;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 128;
  const int W = p.conv_W, H = p.conv_H, Cin = p.conv_Cin;
  const int ups = (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0;
  const int Win = W >> ups, Hin = H >> ups;
  const int wincin = Win * Cin;
  const int img = m0 / (H * W);
  const int rem = m0 - img * (H * W);
  const int oy0 = rem / W;
  const int Wt = W < 256 ? W : 256;                          // tile row width (a segment of the image row when W > 256)
  const int x0 = W > 256 ? rem - oy0 * W : 0;
  const int tr = (256 / Wt) * (Wt + 2);                      // tile rows incl. the halo columns

  // ---- X DMA: 17 pieces of 16 tile rows x 64 B; wave w stages pieces w, w + 4, w + 8, w + 12 (and wave 0 piece 16).
  //      lane -> tile row 16 pc + (lane >> 2), source 8-channel chunk (lane & 3) ^ swz4(row >> 2)
  const rsrc_t rsx = make_rsrc(p.x);
  int colpart[5];               // ((img*Hin)*Win + (ix >> ups)) * Cin + chunk * 8 (a multiple of 8) | tile segment in the low two bits,
                                // or -1: beyond the tile / column halo outside the image
  {
    const int rowp = lane >> 2;
    const int c = (lane & 3) ^ c3_swz4(lane >> 4);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int r = (wid + 4 * j) * 16 + rowp;
      const int sg = r / (Wt + 2);
      const int ix = x0 + (r - sg * (Wt + 2)) - 1;
      colpart[j] = (r < tr && (unsigned)ix < (unsigned)W) ? (((img * Hin * Win + (ix >> ups)) * Cin + c * 8) | sg) : -1;
    }
  }
  auto issue_x = [&](unsigned char* xbuf, int j, int kh, int ci0) {      // piece j of this wave of the tile (kh, channels ci0..ci0+31)
    const int pc = wid + 4 * j;
    if (pc >= 17) return;                                                // wave-uniform
    if ((EXP & 1) && in_loop) return;
    int cp = colpart[j];
    asm volatile("" : "+v"(cp));          // opaque: keeps the compiler from hoisting the two unpacked halves out of the K loop (10 registers)
    const int iy = oy0 + (cp & 3) + kh - 1;
    const bool ok = cp >= 0 && (unsigned)iy < (unsigned)H;
    const uint32_t off = (uint32_t)((cp & ~3) + (iy >> ups) * wincin) * 2u;
    dma16bs(rsx, ok ? off : DMA_OOB, (uint32_t)ci0 * 2u, xbuf + pc * 1024);
  };
  // ---- filter DMA: 8 pieces of 16 rows x 64 B per stage, two per wave
  const rsrc_t rsw = make_rsrc((const uint16_t*)p.w + (int64_t)n0 * p.ldw);
  uint32_t voffw[2];
  {
    const int rowp = lane >> 2;
    const int c = (lane & 3) ^ c3_swz4(lane >> 4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (2 * wid + j) * 16 + rowp;
      voffw[j] = (n0 + row < p.N) ? (uint32_t)(((int64_t)row * p.ldw + c * 8) * 2) : DMA_OOB;
    }
  }
  auto issue_w = [&](int slot, int j, int koff) {                        // koff < 0: no such stage (zeros into a slot nobody reads)
    const bool live = koff >= 0;
    if ((EXP & 2) && in_loop) return;
    dma16bs(rsw, live ? voffw[j] : DMA_OOB, live ? (uint32_t)koff * 2u : 0u, smem + C3_WOFF + slot * C3_WT + (2 * wid + j) * 1024);
  };

  // ---- fragment addresses.  X: pixel px of the wave's half h (64 pixels) -> tile row seg * (Wt + 2) + ix (+ kw for the tap);
  //      blocks inside a half are 16 rows apart (same swizzle phase), the tap shift changes the phase -> one address per (kw, half)
  uint32_t xa[3][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int px = wm * 128 + h * 64 + l15;
    const int base = (px / Wt) * (Wt + 2) + (px % Wt);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int r = base + kw;
      xa[kw][h] = r * 64 + ((g4 ^ c3_swz4(r >> 2)) << 4);
    }
  }
  const int rw = wn * 64 + l15;
  const uint32_t wa = C3_WOFF + rw * 64 + ((g4 ^ c3_swz4(rw >> 2)) << 4);

  f32x4_t acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // Fragment registers: the step multiplies pixel blocks 0..3 (the wave's first 64 pixels) in its first half and 4..7 in its second,
  // each against all four filter blocks — so the X fragments need ONE set (the half that is not being multiplied is being loaded)
  // and only the filter fragments two: 64 registers instead of 96 (the two-set form spilled at 256 registers per lane).
  u32x4_t fw[2][4], fxlo[4], fxhi[4];
  auto read_x = [&](u32x4_t (&x4)[4], int i, const unsigned char* xbuf, int kw, int half) {
    if ((EXP & 8) && in_loop) return;
    x4[i] = *(const u32x4_t*)(xbuf + xa[kw][half] + i * 1024);
  };
  auto read_w = [&](u32x4_t (&w4)[4], int i, int wslot) { if ((EXP & 8) && in_loop) return; w4[i] = *(const u32x4_t*)(smem + wa + wslot * C3_WT + i * 1024); };

  const int nblk = Cin / 32, ngroups = 3 * nblk;
  auto koff_of = [&](int g, int kw) -> int {                            // K offset of the filter stage of step (g, kw); -1 beyond the end
    if (g >= ngroups) return -1;
    const int kh = g / nblk, cb = g - kh * nblk;
    return (3 * kh + kw) * Cin + cb * 32;
  };
  // prologue: X tile of group 0, filter stages of steps (0, 0) and (0, 1)
#pragma unroll
  for (int j = 0; j < 5; ++j) issue_x(smem, j, 0, 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) issue_w(0, j, 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) issue_w(1, j, Cin);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) read_x(fxlo, i, smem, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) read_w(fw[0], i, 0);
  if (EXP & 8) {           // (experiment: every fragment register defined once)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fxhi[i] = fxlo[i];
      fw[1][i] = fw[0][i];
    }
  }
  in_loop = true;

  // One K step: KW = tap (= filter ring slot), P = fragment set, XB = X buffer of the step's group.
  //   g = group index; (khn, cin) = kernel row / first channel of group g + 1 (khn < 0: none)
  auto step = [&](auto kw_tag, auto par_tag, auto xb_tag, int g, int khn, int cin) {
    constexpr int KW = decltype(kw_tag)::value, P = decltype(par_tag)::value, XB = decltype(xb_tag)::value;
    unsigned char* xnxt = smem + (XB ^ 1) * C3_XT;
    const int k2 = KW == 0 ? koff_of(g, 2) : koff_of(g + 1, KW - 1);     // filter stage of step s + 2 -> slot (KW + 2) % 3
    constexpr int S2 = (KW + 2) % 3, S1 = (KW + 1) % 3;
    const unsigned char* xcur = smem + XB * C3_XT;
    // first half: all filter blocks x pixel blocks 0..3 (16 MFMAs) | X fragments 4..7 of THIS step | filter piece 0 of stage s + 2 |
    // X pieces of the next group
    // (pinning the reads in front of the MFMAs with sched_barrier was tried: the allocator then spills 12-32 registers INSIDE the loop, and
    // scratch reloads share vmcnt with the LDS-DMA ring; the compiler's own placement — reads close to their consumers — spills nothing)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        mma16_lo<L>(acc[a][b], fw[P][a], fxlo[b]);
        const int i = a * 4 + b;
        if (i < 4) read_x(fxhi, i, xcur, KW, 1);
        if (i == 5) issue_w(S2, 0, k2);
        if constexpr (KW == 0) {
          if (i == 8 && khn >= 0) issue_x(xnxt, 0, khn, cin);
          if (i == 11 && khn >= 0) issue_x(xnxt, 1, khn, cin);
        } else if constexpr (KW == 1) {
          if (i == 8 && khn >= 0) issue_x(xnxt, 3, khn, cin);
        }
      }
    // everything older than this half's own pieces has landed: the filter stage of step s + 1 and (KW == 2) the next group's X tile
    if constexpr (KW == 0) {
      if (khn >= 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else if constexpr (KW == 1) {
      if (khn >= 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    }
    if (!(EXP & 4)) __builtin_amdgcn_s_barrier();
    // second half: all filter blocks x pixel blocks 4..7 (16 MFMAs) | X fragments 0..3 and the filter fragments of step s + 1 |
    // filter piece 1 | more X pieces
    const unsigned char* xrd = smem + (KW == 2 ? (XB ^ 1) : XB) * C3_XT;
    constexpr int KWN = KW == 2 ? 0 : KW + 1;
    // fxlo is dead since the first half, fw[P ^ 1] since the previous step: the eight reads of step s + 1 go out right behind the barrier
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        mma16_lo<L>(acc[a][4 + b], fw[P][a], fxhi[b]);
        const int i = a * 4 + b;
        if (i < 4) read_x(fxlo, i, xrd, KWN, 0);
        else if (i < 8) read_w(fw[P ^ 1], i - 4, S1);
        if (i == 9) issue_w(S2, 1, k2);
        if constexpr (KW == 0) {
          if (i == 12 && khn >= 0) issue_x(xnxt, 2, khn, cin);
        } else if constexpr (KW == 1) {
          if (i == 12 && khn >= 0) issue_x(xnxt, 4, khn, cin);
        }
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  // two groups (six steps) per iteration: fragment set and X buffer are compile-time (ngroups = 3 Cin / 32 is even: Cin % 64 == 0)
#pragma unroll 1
  for (int g = 0; g < ngroups; g += 2) {
    {
      const int gn = g + 1;
      const int khn = gn < ngroups ? gn / nblk : -1;
      const int cin = gn < ngroups ? (gn - khn * nblk) * 32 : 0;
      step(I0{}, I0{}, I0{}, g, khn, cin);
      step(I1{}, I1{}, I0{}, g, khn, cin);
      step(I2{}, I0{}, I0{}, g, khn, cin);
    }
    {
      const int gn = g + 2;
      const int khn = gn < ngroups ? gn / nblk : -1;
      const int cin = gn < ngroups ? (gn - khn * nblk) * 32 : 0;
      step(I0{}, I1{}, I1{}, g + 1, khn, cin);
      step(I1{}, I0{}, I1{}, g + 1, khn, cin);
      step(I2{}, I1{}, I1{}, g + 1, khn, cin);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // zero-fill pieces of the stages beyond the reduction must not outlive the workgroup
  // the epilogue's per-lane addressing starts from an OPAQUE copy of the lane id: otherwise the compiler computes it (row / column
  // offsets, GroupNorm constants) in front of the K loop, keeps it alive through the loop and spills loop state instead
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  ffvc_gemm_detail::gemm_epilogue_perm16<L, 4, EPI>(p, acc, m0, n0, wm, wn, lane_e, 0, 0, 0);
}

template <typename L, int EPI>
int c3_launch(const ffvc_gemm_desc& d, hipStream_t st) {
  const int tiles_n = d.N / 128, n_tiles = (d.M / 256) * tiles_n;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_row3_kernel<L, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS);
    attr = true;
  }
  static int exp_mode = -1;
  if (exp_mode < 0) {
    const char* e = getenv("FFVC_C3_EXP");
    exp_mode = e ? atoi(e) : 0;
  }
  if constexpr (std::is_same<L, f16_t>::value && EPI == ffvc_gemm_detail::EPI_LEAN) {
    if (exp_mode) {             // timing experiments (wrong results): which part of the K step costs what
      auto go = [&](auto tag) {
        constexpr int E = decltype(tag)::value;
        (void)hipFuncSetAttribute((const void*)conv_row3_kernel<L, EPI, E>, hipFuncAttributeMaxDynamicSharedMemorySize, C3_LDS);
        hipLaunchKernelGGL((conv_row3_kernel<L, EPI, E>), dim3(n_tiles), dim3(256), C3_LDS, st, d, tiles_n, n_tiles);
      };
      if (exp_mode == 1) go(std::integral_constant<int, 1>{});
      else if (exp_mode == 2) go(std::integral_constant<int, 2>{});
      else if (exp_mode == 3) go(std::integral_constant<int, 3>{});
      else if (exp_mode == 4) go(std::integral_constant<int, 4>{});
      else if (exp_mode == 7) go(std::integral_constant<int, 7>{});
      else if (exp_mode == 8) go(std::integral_constant<int, 8>{});
      else go(std::integral_constant<int, 15>{});
      return hipGetLastError() == hipSuccess ? 1 : -1000;
    }
  }
  hipLaunchKernelGGL((conv_row3_kernel<L, EPI>), dim3(n_tiles), dim3(256), C3_LDS, st, d, tiles_n, n_tiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ffvc_set_error("conv_row3 launch failed: %s", hipGetErrorString(e));
    return -(int)e - 1000;
  }
  return 1;
}

}  // namespace

// The caller (gemm2.hip) has checked the row-tile geometry (W in {64, 128, 256 k}, H W % 256 == 0, N % 128 == 0, M % 256 == 0,
// batch 1, no split-K, no activation).  0 = not taken.
extern int g_gnb_probe;      // gemm.hip: ffvc_gemm_gnb_probe is asking whether the launch would be taken (nothing is launched)

int ffvc_conv3_try(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok) {
  using namespace ffvc_gemm_detail;
  if (vec_ok != 2 || d.in_dtype == FFVC_F32 || (d.conv_Cin % 64) != 0 || d.alpha != 1.0f) return 0;
  if (d.flags & (FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_BIAS_ALONG_M | FFVC_F_OUT_F32)) return 0;
  if (!g8_offsets_ok<FFVC_OP_CONV3X3>(d)) return 0;
  const bool gnv = (d.flags & FFVC_F_GN_SUMS) != 0;
  if (d.flags & FFVC_F_GNB_SUMS) {
    // backward statistics of the GroupNorm node whose output gradient this dgrad stores: x is read through y's row offsets
    const bool ok = !gnv && d.gnb_x && d.gnb_mean && d.gnb_rstd && d.gnb_gamma && d.gnb_beta && d.gnb_sums && d.gn_hw > 0 &&
                    (d.gn_hw % 256) == 0 && d.gn_cpg >= 4 && (d.gn_cpg % 4) == 0 && (d.N % d.gn_cpg) == 0 && (d.M % d.gn_hw) == 0 &&
                    d.y_mi == 0 && d.y_sm == d.N && d.batch == 1 && ((uintptr_t)d.gnb_x % 16) == 0;
    if (!ok) return 0;
    if (g_gnb_probe) return 3;
    if (!d.residual) {         // a dgrad: plain 16-bit store, none of the generic epilogue's flag tests
      if (d.in_dtype == FFVC_F16) return c3_launch<f16_t, EPI_LEAN | EPI_O_T | EPI_GNB>(d, st);
      return c3_launch<uint16_t, EPI_LEAN | EPI_O_T | EPI_GNB>(d, st);
    }
    if (d.in_dtype == FFVC_F16) return c3_launch<f16_t, EPI_LEAN | EPI_GNB>(d, st);
    return c3_launch<uint16_t, EPI_LEAN | EPI_GNB>(d, st);
  }
  if (!gnv && !d.residual) {   // dgrad convolutions without a GroupNorm in front (upsample levels, conv_in): the straight store
    if (d.in_dtype == FFVC_F16) return c3_launch<f16_t, EPI_LEAN | EPI_O_T>(d, st);
    return c3_launch<uint16_t, EPI_LEAN | EPI_O_T>(d, st);
  }
  if (d.in_dtype == FFVC_F16) return gnv ? c3_launch<f16_t, EPI_GN>(d, st) : c3_launch<f16_t, EPI_LEAN>(d, st);
  return gnv ? c3_launch<uint16_t, EPI_GN>(d, st) : c3_launch<uint16_t, EPI_LEAN>(d, st);
}
