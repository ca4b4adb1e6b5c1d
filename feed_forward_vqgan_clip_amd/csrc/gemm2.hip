// gemm2.hip — LDS-DMA staged bf16 fast path of ffvc_gemm (second kernel generation).
//
// Why: profiling the register-staged kernel (gemm.hip) on MI355X shows it is bound by the LDS *write* path:
// one K step stages 32 KB through ds_write_b128 (~79 B/clk/CU => ~415 cycles) next to 512 cycles of MFMA.
// Here operand tiles travel HBM/L2 -> LDS directly with `global_load_lds_dwordx4` (1 KiB per wave-instruction,
// no VGPR round trip, no ds_write), into a 2-deep LDS ring so the DMA of tile t+1 overlaps the MFMAs of tile t,
// with ONE barrier per K step:
//
//     loop t:  s_waitcnt vmcnt(0)   (my pieces of tile t landed)
//              barrier              (everyone's pieces landed; everyone finished reading tile t-1)
//              issue DMA of tile t+1 into ring[(t+1)&1]
//              64 MFMAs on ring[t&1]
//
// LDS-DMA writes lane-linearly (wave-uniform base + lane*16 B), so the bank-conflict-free image is obtained by
// permuting the per-lane SOURCE address and applying the same involution on the fragment read:
//   K-major / conv tile  [128 rows][128 B]: two rows form a 256-B line of 16 slots; slot' = slot ^ (line & 15)
//                         -> the 16 lanes of every ds_read_b128 lane group hit 16 distinct slots;
//   reduction-major tile [64 k][256 B]:     slot' = slot ^ ((k & 3) << 2) -> the 4 k-rows x 2 column blocks of a
//                         ds_read_b64_tr_b16 lane group land on 8 distinct 32-B ranges.
// Out-of-range rows / conv halo taps / K tails read from a zeroed page instead of being masked (the DMA cannot
// write zeros for inactive lanes).
//
// Same tile geometry and epilogue as v1 (128x128, 4 waves x 64x64, MFMA 32x32x16, A := W rows, B := X rows).
// Shapes this path does not cover (fp32, rows not 16-byte aligned, ...) fall back to gemm.hip.
#include "gemm2_kernels.h"

// the other operand-mode pairs are instantiated in gemm2_modes.hip / gemm2_conv.hip (parallel compilation)
int ffvc_gemm2_launch_kk(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg);
int ffvc_gemm2_launch_conv(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg);
int ffvc_gemm2_launch_nn(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg);
int ffvc_gemm2_launch_tn(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg);
int ffvc_gemm2_launch_tt(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, const uint16_t* zero, int cfg);
// gemm3.hip: the 256x128 ring kernel with two workgroups per CU (epilogue-heavy K-major x K-major launches)
int ffvc_gemm3_try(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, int mode);
// conv3.hip: the software-pipelined row-tile convolution, two workgroups per CU
int ffvc_conv3_try(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok);
int g_opt_conv_row3 = -100; // ffvc_set_option("conv_row3", v): 0 off | 1 on; unset -> FFVC_CONV_ROW3 (default 1)
int g_opt_gemm3 = -100;     // ffvc_set_option("gemm3", v): -1 never | 0 heuristic | 1 every eligible launch; unset -> FFVC_GEMM3 (default 0)

namespace {


// 3x3 conv with the haloed row tile: block tile 256 pixels x 128 output channels, 4 waves x (128 x 64), ONE stage
// (X 33 KiB + W 16 KiB) so that two workgroups share a CU and alternate DMA / MFMA phases.
// K order: kh (3) x 64-channel block (Cin/64) x kw (3); the X tile is (re)loaded only when (kh, block) changes.
#ifdef FFVC_CR_TIMING
// debug build (tools/cr_timing.py): s_memtime stamps of every wave of one workgroup over K steps 3..8 (two kw sweeps)
__device__ unsigned long long cr_stamps[4][6][8];     // [wave][step - 3][event]
#define CR_STAMP(ev)                                                                                          \
  do {                                                                                                        \
    if (blockIdx.x == 300 && lane == 0 && cr_step >= 3 && cr_step < 9) cr_stamps[wid][cr_step - 3][ev] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define CR_STAMP(ev)
#endif
// WPF (round 4): the filter tile is double-buffered (the second stage takes the place of the row-store pads, which alias the X
// tile during the epilogue: same 65 KiB, still two workgroups per CU).  The W tile of step s+1 is issued right after the barrier
// that opens step s, so the kw = 1, 2 steps find their operands already in LDS (one barrier per step instead of two); only the
// X reload of a kw = 0 step (every third) still waits on the memory side, covered by the partner workgroup as before.
template <typename L, bool BUF, int EPI = ffvc_gemm_detail::EPI_GN, bool WPF = false>
__global__ __launch_bounds__(256, 2) void conv_row_kernel(const ffvc_gemm_desc p, int tiles_n, int n_tiles, int vec_ok,
                                                          const uint16_t* zero) {
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
  const int W = p.conv_W, Cin = p.conv_Cin;
  typename std::conditional<BUF, ConvRowDmaB, ConvRowDma>::type sx;
  typename std::conditional<BUF, KMajorDmaB<BN, 4>, KMajorDma<BN, 4>>::type sw;
  sx.init((const uint16_t*)p.x, m0, p.conv_H, W, Cin, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
  if constexpr (BUF)
    sw.init((const uint16_t*)p.w + (int64_t)n0 * p.ldw, p.ldw, p.N - n0, tid);
  else
    sw.init((const uint16_t*)p.w, p.ldw, n0, p.N, 0, 0, tid, 0, 0);

  constexpr bool M16 = FFVC_MFMA16_CONVROW;         // v_mfma_f32_16x16x32 (see gemm2_kernel)
  f32x16_t acc[M16 ? 1 : 2][M16 ? 1 : MT];
  f32x4_t acc16[M16 ? 4 : 1][M16 ? 2 * MT : 1];
  if constexpr (M16) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2 * MT; ++b) acc16[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;
  }

  // LDS row of this lane's pixel for tap kw = 0 (add kw for the others): 32-row blocks (32x32x16) or 16-row blocks (16x16x32)
  // (16x16x32: the two 16-row halves of a 32-pixel block lie in the same image row (tile rows are 64 / 128 / 256 wide), so the
  // second half is the first + 16)
  int xrow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int px = wm * 128 + t * 32 + (M16 ? (lane & 15) : l31);
    const int Wt = W < 256 ? W : 256;               // tile row width (a segment of the image row when W > 256)
    xrow[t] = (px / Wt) * (Wt + 2) + (px % Wt);
  }
  unsigned char* sX = smem;
  unsigned char* sW = smem + XTILE;
  const int nblk = Cin / 64;
  int cr_step = 0;
  (void)cr_step;
  if constexpr (WPF) {
    sx.issue(sX, 0, 0, zero, tid);
    sw.issue(sW, 0, p.K, zero, tid);
  }
  for (int kh = 0; kh < 3; ++kh) {
    for (int cb = 0; cb < nblk; ++cb) {
#pragma unroll 1
      for (int kw = 0; kw < 3; ++kw, ++cr_step) {
        CR_STAMP(0);
        if constexpr (!WPF) {
          if (kw == 0) sx.issue(sX, kh, cb * 64, zero, tid);
          sw.issue(sW, (kh * 3 + kw) * Cin + cb * 64, p.K, zero, tid);
        }
        CR_STAMP(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CR_STAMP(2);
        __syncthreads();
        CR_STAMP(3);
        if constexpr (WPF) {
          // everyone has finished step s-1 (the last reader of the other W stage): stage the next step's filter tile now
          sW = smem + XTILE + (cr_step & 1) * WTILE;
          int kw2 = kw + 1, cb2 = cb, kh2 = kh;
          if (kw2 == 3) {
            kw2 = 0;
            if (++cb2 == nblk) {
              cb2 = 0;
              ++kh2;
            }
          }
          if (kh2 < 3) sw.issue(smem + XTILE + ((cr_step + 1) & 1) * WTILE, (kh2 * 3 + kw2) * Cin + cb2 * 64, p.K, zero, tid);
        }
        if constexpr (M16) {
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            u32x4_t fa[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = frag16_kmajor(sW, wn * 64 + t * 16 + (lane & 15), sub, lane);
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {          // X fragments in two halves of 4 blocks: 32 instead of 48 live registers
              u32x4_t fb[MT];
#pragma unroll
              for (int t = 0; t < MT; ++t) {
                const int tb = bh * MT + t;           // 16-row block of the wave tile
                fb[t] = frag16_kmajor(sX, xrow[tb >> 1] + 16 * (tb & 1) + kw, sub, lane);
              }
#pragma unroll
              for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int t = 0; t < MT; ++t) mma16_lo<L>(acc16[a][bh * MT + t], fa[a], fb[t]);
              __builtin_amdgcn_sched_barrier(0);      // keep the halves sequential: the scheduler must not hoist the next half's reads
            }
            if (sub == 0) CR_STAMP(4);
          }
          CR_STAMP(5);
        } else {
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) {
            u32x4_t fa[2], fb[MT];
#pragma unroll
            for (int t = 0; t < 2; ++t) fa[t] = frag_kmajor(sW, wn * 64 + t * 32 + l31, sub, lane);
#pragma unroll
            for (int t = 0; t < MT; ++t) fb[t] = frag_kmajor(sX, xrow[t] + kw, sub, lane);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < MT; ++b) mma_lo<L>(acc[a][b], fa[a], fb[b]);
          }
        }
        if constexpr (WPF) {
          if (kw == 2) {                      // the X tile is single-buffered: reload it once every wave is done with it
            __syncthreads();
            int cb2 = cb + 1, kh2 = kh;
            if (cb2 == nblk) {
              cb2 = 0;
              ++kh2;
            }
            if (kh2 < 3) sx.issue(sX, kh2, cb2 * 64, zero, tid);
          }
        } else {
          __syncthreads();
        }
        CR_STAMP(6);
      }
    }
  }
  if constexpr (WPF) {
    // the row-store pads alias the X tile (all X / W reads retired behind the barrier that closed the last step)
    if constexpr (M16) {
      if (vec_ok == 2)
        ffvc_gemm_detail::gemm_epilogue_out16<L, MT, EPI>(p, acc16, m0, n0, wm, wn, lane, 0, 0, smem + wid * 4096);
      else
        ffvc_gemm_detail::gemm_epilogue16<L, MT, true>(p, acc16, m0, n0, wm, wn, lane, 0, 0, 1);
    } else {
      if (vec_ok == 2)
        ffvc_gemm_detail::gemm_epilogue_rows<L, MT, false, EPI>(p, acc, m0, n0, wm, wn, lane, 0, 0, smem + wid * 4096);
      else
        ffvc_gemm_detail::gemm_epilogue<L, MT, true>(p, acc, m0, n0, wm, wn, lane, 0, 0, 1);
    }
    return;
  }
  if constexpr (M16) {
    if (vec_ok == 2)
      ffvc_gemm_detail::gemm_epilogue_out16<L, MT, EPI>(p, acc16, m0, n0, wm, wn, lane, 0, 0, smem + XTILE + WTILE + wid * 4096);
    else
      ffvc_gemm_detail::gemm_epilogue16<L, MT, true>(p, acc16, m0, n0, wm, wn, lane, 0, 0, 1);
  } else {
    if (vec_ok == 2)
      ffvc_gemm_detail::gemm_epilogue_rows<L, MT, false, EPI>(p, acc, m0, n0, wm, wn, lane, 0, 0, smem + XTILE + WTILE + wid * 4096);
    else
      ffvc_gemm_detail::gemm_epilogue<L, MT, true>(p, acc, m0, n0, wm, wn, lane, 0, 0, 1);
  }
}


// ---- 3x3 conv, haloed row tile, software-pipelined (round 3) -------------------------------------------------------------
// conv_row_kernel above spends 59 % of its time outside MFMAs (tools/cr_timing.py, profiles/r03_conv_row_timing.txt): every K
// step pays the DMA issue (100-190 cycles per 1-KiB piece, 13 pieces when the X tile is reloaded), the wait for the data, two
// barriers and four exposed LDS round trips, and the second workgroup on the CU does not cover them.  Here ONE workgroup owns
// the CU (4 waves = one per SIMD, 512 registers each) and the K steps form a pipeline, as in tools/probe kernel G:
//   * X tile double-buffered (2 x 33 KiB), filter tile in a ring of three (3 x 16 KiB): W of step s+2 and the X tile of the next
//     (kh, channel block) group stream in under the MFMAs of step s — 4 pieces after the barrier-free first half, the X pieces
//     in the second half of the kw = 0 and kw = 1 steps; ONE barrier per step with a counted vmcnt (only this step's own four
//     filter pieces may still be in flight);
//   * fragments double-buffered in registers: the second 32-deep half of a step is fetched under the first half's MFMAs, the
//     first half of step s+1 under the second half of step s (behind the barrier that publishes its tiles);
//   * accumulators pinned to the AGPR file through inline-asm MFMAs (hipcc otherwise shuffles 16x16x32 accumulators between the
//     two register files of a 512-register kernel, tools/probe/gemm_probe.hip).
// Same tile geometry, DMA address generators, LDS images and epilogue as conv_row_kernel.
// MEASURED (round 3, tools/cr_timing.py): a K step takes ~2500 cycles here against ~4500 in conv_row_kernel (two workgroups), i.e.
// the pipeline works, but a half of 32 MFMAs (512 pipe cycles) still takes 750-1260 cycles because every LDS-DMA piece costs the
// issuing wave ~100 cycles of address VALU + issue with nothing else on the SIMD to cover it, and with ONE workgroup per CU the
// prologue (~6k cycles) and the row-store epilogue (~12k cycles of a 63k-cycle tile) are exposed: 635 vs 777 TFLOP/s at 256^2
// 128->128, cfg2 step 148.8 vs 136.5 ms.  Kept behind FFVC_CONV_ROW2=1 for the next round (a persistent tile loop that overlaps
// the epilogue with the next tile's first steps is what it needs), off by default.
template <typename L>
__device__ __forceinline__ void mma16_agpr(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b);
template <>
__device__ __forceinline__ void mma16_agpr<uint16_t>(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <>
__device__ __forceinline__ void mma16_agpr<f16_t>(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <typename L, int EPI = ffvc_gemm_detail::EPI_GN>
__global__ __launch_bounds__(256, 1) void conv_row2_kernel(const ffvc_gemm_desc p, int tiles_n, int n_tiles, int vec_ok) {
  constexpr int MT = 4, BM = 256, BN = 128;
  constexpr int XTILE = 264 * 128, WTILE = BN * 128;
  constexpr int XOFF = 0, WOFF = 2 * XTILE;                 // [X0 | X1 | W0 | W1 | W2]; the epilogue pads alias X0
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;
  const int l15 = lane & 15;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int W = p.conv_W, Cin = p.conv_Cin;
  ConvRowDmaB sx;
  KMajorDmaB<BN, 4> sw;
#ifdef FFVC_CR_TIMING
  if (blockIdx.x == 300 && lane == 0) cr_stamps[wid][0][5] = __builtin_amdgcn_s_memtime();
#endif
  sx.init((const uint16_t*)p.x, m0, p.conv_H, W, Cin, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
  sw.init((const uint16_t*)p.w + (int64_t)n0 * p.ldw, p.ldw, p.N - n0, tid);

  f32x4_t acc[4][2 * MT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2 * MT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int xrow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int px = wm * 128 + t * 32 + l15;
    const int Wt = W < 256 ? W : 256;
    xrow[t] = (px / Wt) * (Wt + 2) + (px % Wt);
  }
  const int nblk = Cin / 64;
  u32x4_t fw[2][4], fx[2][2 * MT];
  auto read_frag = [&](int i, const unsigned char* sX, const unsigned char* sW, int kw, int sub, u32x4_t (&w4)[4], u32x4_t (&x8)[2 * MT]) {
    // i = 0..11: the 8 X blocks first (every MFMA of a half needs them), then the 4 W blocks
    if (i < 8) x8[i] = frag16_kmajor(sX, xrow[i >> 1] + 16 * (i & 1) + kw, sub, lane);
    else w4[i - 8] = frag16_kmajor(sW, wn * 64 + (i - 8) * 16 + l15, sub, lane);
  };

  // prologue: X tile of group 0, filter tiles of steps 0 and 1 (K offsets in the [Cout][kh][kw][Cin] filter: (3 kh + kw) Cin + 64 cb)
#pragma unroll
  for (int j = 0; j < 9; ++j) sx.issue1(smem + XOFF, j, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) sw.issue1(smem + WOFF, j, 0, p.K);
#pragma unroll
  for (int j = 0; j < 4; ++j) sw.issue1(smem + WOFF + WTILE, j, Cin, p.K);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 12; ++i) read_frag(i, smem + XOFF, smem + WOFF, 0, 0, fw[0], fx[0]);

  int cr_step = 0;                                           // (debug stamps only)
  (void)cr_step;
  // One K step.  KW is compile-time (it selects which X pieces of the NEXT group travel in this step and where step s + 2's
  // filter tile lies); wslot = s % 3 (ring slot of this step's filter tile), xcur / xnxt = the two X buffers, k2 = K offset of
  // step s + 2 (< 0: none), (khn, cin) = next group (khn < 0: none), more = a step s + 1 exists.
  auto step = [&](auto kw_tag, int wslot, const unsigned char* xcur, unsigned char* xnxt, int k2, int khn, int cin, bool more) {
    constexpr int KW = decltype(kw_tag)::value;
    const unsigned char* sW = smem + WOFF + wslot * WTILE;
    const int s1 = wslot == 2 ? 0 : wslot + 1, s2 = wslot == 0 ? 2 : wslot - 1;     // (s + 1) % 3, (s + 2) % 3
    unsigned char* wnext = smem + WOFF + s2 * WTILE;
    // ---- first half: MFMAs (s, k 0..31) | fragments (s, k 32..63) | the four filter pieces of step s + 2
    CR_STAMP(0);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2 * MT; ++b) {
        mma16_agpr<L>(acc[a][b], fw[0][a], fx[0][b]);
        const int i = a * 8 + b;
        if (i < 12) read_frag(i, xcur, sW, KW, 1, fw[1], fx[1]);
        if (i >= 14 && i < 30 && ((i - 14) & 3) == 0 && k2 >= 0) sw.issue1(wnext, (i - 14) >> 2, k2, p.K);
      }
    // everything older than this step's own filter pieces has landed (W of step s + 1, the X pieces issued in earlier steps)
    CR_STAMP(1);
    if (k2 >= 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    CR_STAMP(2);
    __builtin_amdgcn_s_barrier();
    CR_STAMP(3);
    // ---- second half: MFMAs (s, k 32..63) | fragments (s + 1, k 0..31) | X pieces of the next group (kw = 0: 0..4, kw = 1: 5..8)
    const unsigned char* sXn = (KW == 2) ? (const unsigned char*)xnxt : xcur;
    const unsigned char* sWn = smem + WOFF + s1 * WTILE;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2 * MT; ++b) {
        mma16_agpr<L>(acc[a][b], fw[1][a], fx[1][b]);
        const int i = a * 8 + b;
        if (i < 12 && more) read_frag(i, sXn, sWn, KW == 2 ? 0 : KW + 1, 0, fw[0], fx[0]);
        if constexpr (KW == 0) {
          if (i >= 13 && i < 28 && ((i - 13) % 3) == 0 && khn >= 0) sx.issue1(xnxt, (i - 13) / 3, khn, cin);
        } else if constexpr (KW == 1) {
          if (i >= 13 && i < 25 && ((i - 13) % 3) == 0 && khn >= 0) sx.issue1(xnxt, 5 + (i - 13) / 3, khn, cin);
        }
      }
    CR_STAMP(4);
    ++cr_step;
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  int gpar = 0;                                              // parity of the group index: which X buffer is current
  for (int kh = 0; kh < 3; ++kh) {
#pragma unroll 1
    for (int cb = 0; cb < nblk; ++cb) {
      const bool last = (kh == 2 && cb == nblk - 1);
      const int cbn = cb + 1 < nblk ? cb + 1 : 0;
      const int khn = last ? -1 : (cb + 1 < nblk ? kh : kh + 1);
      const unsigned char* xcur = smem + XOFF + gpar * XTILE;
      unsigned char* xnxt = smem + XOFF + (gpar ^ 1) * XTILE;
      // a group is three steps, so the filter ring (three slots) is at slot 0 at the start of every group
      step(K0{}, 0, xcur, xnxt, (kh * 3 + 2) * Cin + cb * 64, khn, cbn * 64, true);
      step(K1{}, 1, xcur, xnxt, last ? -1 : (khn * 3 + 0) * Cin + cbn * 64, khn, cbn * 64, true);
      step(K2{}, 2, xcur, xnxt, last ? -1 : (khn * 3 + 1) * Cin + cbn * 64, khn, cbn * 64, !last);
      gpar ^= 1;
    }
  }
#ifdef FFVC_CR_TIMING
  if (blockIdx.x == 300 && lane == 0) cr_stamps[wid][0][6] = __builtin_amdgcn_s_memtime();
#endif
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // the last MFMAs' results before the VALU reads them
  __builtin_amdgcn_s_barrier();                              // every wave is done with the X tiles: the pads may alias X0
  if (vec_ok == 2)
    ffvc_gemm_detail::gemm_epilogue_out16<L, MT, EPI>(p, acc, m0, n0, wm, wn, lane, 0, 0, smem + wid * 4096);
  else
    ffvc_gemm_detail::gemm_epilogue16<L, MT, true>(p, acc, m0, n0, wm, wn, lane, 0, 0, 1);
#ifdef FFVC_CR_TIMING
  if (blockIdx.x == 300 && lane == 0) cr_stamps[wid][0][7] = __builtin_amdgcn_s_memtime();
#endif
}


// ---- 3x3 conv with a handful of output channels (the decoder's conv_out: 128 -> 3) -------------------------------------------
// N = 3 wastes 125 of the 128 columns of every other tile (round 2: 1.4 ms at 256^2 x 64 images on the register-staged kernel,
// 20 TFLOP/s).  The work is a stream: 1 GB of activations in, 50 MB out.  Same haloed X row tile as conv_row_kernel (one load per
// (kernel row, 64-channel block), three kw taps served from it), the filter tile is 16 rows (rows >= N read as zeros), every wave
// owns 64 pixels x one 16-wide MFMA column block; 36 KiB of LDS and < 64 registers -> four workgroups per CU hide the loads.
template <typename L>
__global__ __launch_bounds__(256, 4) void conv_row_n16_kernel(const ffvc_gemm_desc p, int n_tiles, const uint16_t* zero) {
  constexpr int XTILE = 264 * 128;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int m0 = tile * 256;
  const int W = p.conv_W, Cin = p.conv_Cin;
  ConvRowDmaB sx;
  sx.init((const uint16_t*)p.x, m0, p.conv_H, W, Cin, (p.flags & FFVC_F_UPSAMPLE2X) ? 1 : 0, tid);
  // filter tile: 16 rows x 128 B = two 1-KiB pieces (waves 0 and 1), K-major image of gemm2_kernels.h
  const rsrc_t rsw = make_rsrc(p.w);
  uint32_t voffw;
  {
    const int line = 4 * (wid & 1) + (lane >> 4);
    const int cp = (lane & 15) ^ (line & 15);
    const int row = 2 * line + (cp >> 3);
    voffw = row < p.N ? (uint32_t)((int64_t)row * p.ldw + (cp & 7) * EPC) * 2u : DMA_OOB;
  }
  unsigned char* sX = smem;
  unsigned char* sW = smem + XTILE;
  f32x4_t acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int xrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int px = wid * 64 + t * 32 + l15;
    const int Wt = W < 256 ? W : 256;
    xrow[t] = (px / Wt) * (Wt + 2) + (px % Wt);
  }
  const int nblk = Cin / 64;
  for (int kh = 0; kh < 3; ++kh) {
    for (int cb = 0; cb < nblk; ++cb) {
#pragma unroll 1
      for (int kw = 0; kw < 3; ++kw) {
        if (kw == 0) sx.issue(sX, kh, cb * 64, zero, tid);
        if (wid < 2) dma16bs(rsw, voffw, (uint32_t)(((kh * 3 + kw) * Cin + cb * 64) * 2), sW + wid * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const u32x4_t fw = frag16_kmajor(sW, l15, sub, lane);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const u32x4_t fx = frag16_kmajor(sX, xrow[b >> 1] + 16 * (b & 1) + kw, sub, lane);
            mma16_lo<L>(acc[b], fw, fx);
          }
        }
        __syncthreads();
      }
    }
  }
  // lane: pixel m = 16 b + (lane & 15) of the wave's 64, output channels n = 4 (lane >> 4) .. + 3
  const bool out32 = p.flags & FFVC_F_OUT_F32;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int m = m0 + wid * 64 + b * 16 + l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = 4 * g4 + i;
      if (n >= p.N) continue;
      const float v = acc[b][i] * p.alpha + (p.bias ? p.bias[n] : 0.0f);
      if (out32) ((float*)p.y)[(int64_t)m * p.y_sm + n] = v;
      else ElemTraits<L>::store((L*)p.y + (int64_t)m * p.y_sm + n, v);
    }
  }
}

uint16_t* g_zero_page[16] = {nullptr};

// run-time options (ffvc_set_option); -1 = not initialised (take the environment variable, else the default)
int g_opt_gemm2_tile = -1;   // FFVC_GEMM2_BM: 0 off | 1 heuristic | 128 | 256 | 512 (= 256x256)
int g_opt_conv_row = -1;     // FFVC_CONV_ROW: 0 off | 1 heuristic | 2 force whenever the geometry allows

int opt_value(int& slot, const char* env, int dflt) {
  if (slot < 0) {
    const char* e = getenv(env);
    slot = e ? atoi(e) : dflt;
  }
  return slot;
}

const uint16_t* zero_page() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!g_zero_page[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    // one-time: the null-stream memset is not ordered against torch's non-blocking streams -> wait for it here
    if (hipMemset(p, 0, 4096) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    g_zero_page[dev] = (uint16_t*)p;
  }
  return g_zero_page[dev];
}

inline bool m8(int64_t v) { return (v % 8) == 0; }

}  // namespace

// 1 = enqueued here, 0 = shape not eligible (caller falls back to gemm.hip), < 0 = launch error.
int ffvc_gemm2_try(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok) {
  if ((d.in_dtype != FFVC_BF16 && d.in_dtype != FFVC_F16) || (d.flags & FFVC_F_TR_SAFE)) return 0;
  if ((d.flags & FFVC_F_GNB_SUMS) && !(d.x_mode == FFVC_OP_CONV3X3 && d.w_mode == FFVC_OP_KMAJOR && d.N > 16)) return 0;   // conv3 only
  if (d.x_mode == FFVC_OP_CONV3X3 && d.w_mode == FFVC_OP_KMAJOR && d.N <= 16) {
    // a 3x3 conv with <= 16 output channels (conv_out): the narrow-N row-tile kernel
    static int n16 = -1;
    if (n16 < 0) {
      const char* e = getenv("FFVC_CONV_N16");
      n16 = e ? atoi(e) : 1;
    }
    const int W = d.conv_W;
    const uint16_t* zero = zero_page();
    const bool ok = n16 && zero && (W == 64 || W == 128 || (W >= 256 && W % 256 == 0)) && ((int64_t)d.conv_H * W) % 256 == 0 && d.batch == 1 &&
                    d.split_k <= 1 && (d.M % 256) == 0 && (d.conv_Cin % 64) == 0 && d.K == 9 * d.conv_Cin && d.act == FFVC_ACT_NONE &&
                    !d.residual && !d.aux && d.y_mi == 0 && (d.ldw % 8) == 0 && ((uintptr_t)d.x % 16) == 0 && ((uintptr_t)d.w % 16) == 0 &&
                    !(d.flags & (FFVC_F_GN_SUMS | FFVC_F_COLSUM | FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_MUL_ACT_GRAD |
                                 FFVC_F_WRITE_PREACT | FFVC_F_BIAS_ALONG_M)) &&
                    g8_offsets_ok<FFVC_OP_CONV3X3>(d);
    if (ok) {
      const int n_tiles = d.M / 256;
      constexpr int lds = 264 * 128 + 2048;
      if (d.in_dtype == FFVC_F16) hipLaunchKernelGGL((conv_row_n16_kernel<f16_t>), dim3(n_tiles), dim3(256), lds, st, d, n_tiles, zero);
      else hipLaunchKernelGGL((conv_row_n16_kernel<uint16_t>), dim3(n_tiles), dim3(256), lds, st, d, n_tiles, zero);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        ffvc_set_error("conv_row_n16 launch failed: %s", hipGetErrorString(e));
        return -(int)e - 1000;
      }
      return 1;
    }
  }
  if (!vec_ok || (d.N % 4) != 0) return 0;     // this path carries the vectorised epilogue only
  // the DMA moves whole 16-byte chunks from 16-byte aligned addresses
  if (((uintptr_t)d.x % 16) || ((uintptr_t)d.w % 16)) return 0;
  if (!m8(d.xbo) || !m8(d.xbi) || !m8(d.wbo) || !m8(d.wbi) || !m8(d.ldw)) return 0;
  if (d.x_mode != FFVC_OP_CONV3X3 && !m8(d.ldx)) return 0;
  if (d.x_mode == FFVC_OP_KMAJOR && (!m8(d.K) || (d.x_mi && !m8(d.x_so)))) return 0;
  if (d.w_mode == FFVC_OP_KMAJOR && !m8(d.K)) return 0;
  // reduction-major operands: whole chunks along M / N — either the extent is a multiple of 8 or the row stride leaves room to read
  // the last chunk in full (gemm2_kernels.h tr_cols)
  // FFVC_TT_PAD=1 admits the second form (0, the default, insists on multiples of 8: see ops._SLN_INPLACE / DESIGN.md section 5)
  static int tt_pad = -1;
  if (tt_pad < 0) {
    const char* e = getenv("FFVC_TT_PAD");
    tt_pad = e ? atoi(e) : 0;
  }
  if (d.x_mode == FFVC_OP_TRANS && !m8(d.M) && (!tt_pad || d.ldx < ((d.M + 7) & ~7))) return 0;
  if (d.w_mode == FFVC_OP_TRANS && !m8(d.N) && (!tt_pad || d.ldw < ((d.N + 7) & ~7))) return 0;
  if (d.kseg && (!m8(d.xkso) || !m8(d.wkso))) return 0;
  const uint16_t* zero = zero_page();
  if (!zero) return 0;
  {
    // row-store epilogue (vec_ok = 2): 8 consecutive n per lane -> N % 8 and 16-byte aligned rows everywhere
    static int rows_opt = -1;
    if (rows_opt < 0) {
      const char* e = getenv("FFVC_EPI_ROWS");
      rows_opt = e ? atoi(e) : 1;
    }
    const bool out32 = d.flags & FFVC_F_OUT_F32;
    bool ok = rows_opt && m8(d.N) && (out32 ? (d.y_sm % 4 == 0 && d.y_so % 4 == 0 && d.ybo % 4 == 0 && d.ybi % 4 == 0 && d.slab_stride % 4 == 0)
                                            : (m8(d.y_sm) && m8(d.y_so) && m8(d.ybo) && m8(d.ybi) && m8(d.slab_stride)));
    if (d.residual) {
      const bool r32 = d.flags & FFVC_F_RES_F32;
      ok = ok && (r32 ? (d.r_sm % 4 == 0 && d.r_so % 4 == 0 && d.rbo % 4 == 0 && d.rbi % 4 == 0)
                      : (m8(d.r_sm) && m8(d.r_so) && m8(d.rbo) && m8(d.rbi)));
    }
    if (d.aux) ok = ok && m8(d.ldaux) && m8(d.abo) && m8(d.abi);
    if (ok) vec_ok = 2;
  }
  if (d.flags & FFVC_F_GN_SUMS) {
    const bool gok = vec_ok == 2 && d.gn_sums && d.gn_hw > 0 && (d.gn_hw % 256) == 0 && d.gn_cpg >= 4 &&
                     (d.gn_cpg % 4) == 0 && (d.N % d.gn_cpg) == 0 && d.batch == 1 && d.split_k <= 1 && d.y_mi == 0 &&
                     (d.M % d.gn_hw) == 0 && !(d.flags & (FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT));
    if (!gok) {
      ffvc_set_error("ffvc_gemm: FFVC_F_GN_SUMS needs the row-store epilogue, gn_hw %% 256 == 0, gn_cpg %% 4 == 0 "
                     "(gn_hw=%d gn_cpg=%d N=%d)", d.gn_hw, d.gn_cpg, d.N);
      return FFVC_E_BADARG;
    }
  }
  if (d.flags & FFVC_F_COLSUM) {
    if (!(vec_ok == 2 && d.colsum && d.batch == 1 && d.split_k <= 1 && !(d.flags & (FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT)))) {
      ffvc_set_error("ffvc_gemm: FFVC_F_COLSUM needs the row-store epilogue (N %% 8 == 0, aligned rows), batch 1, no split-K");
      return FFVC_E_BADARG;
    }
  }
  if (d.flags & FFVC_F_VQ_ARGMIN) {
    // vector-quantisation distances with the argmin folded in: the 256x256 ring kernel only, whatever tile the heuristic would pick
    // (launch2 refuses — 0 — when the operands cannot take its buffer-descriptor form)
    const bool ok = d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_KMAJOR && d.batch == 1 && d.split_k <= 1 && d.slab_stride == 0 &&
                    d.alpha == 1.0f && !d.bias && !d.residual && !d.aux && d.act == FFVC_ACT_NONE && d.kseg == 0 && d.x_mi == 0 &&
                    d.vq_xn && d.vq_cn && d.vq_out &&
                    !(d.flags & ~(FFVC_F_VQ_ARGMIN | FFVC_F_OUT_F32));
    return ok ? ffvc_gemm2_launch_kk(d, st, vec_ok, zero, 512) : 0;
  }
  // tile selection: FFVC_GEMM2_BM = 0 (disable this path) | 128 | 256 | 512 (= 256x256) | unset (heuristic)
  const int env_bm = opt_value(g_opt_gemm2_tile, "FFVC_GEMM2_BM", 1);
  if (env_bm == 0) return 0;
  int cfg = (env_bm == 64 || env_bm == 128 || env_bm == 256 || env_bm == 512) ? env_bm : 0;
  if (cfg == 64 && !(d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_KMAJOR)) cfg = 128;     // the 64x128 tile exists K-major x K-major only
  if (!cfg) {
    // Every variant occupies a CU with 8 waves; what differs is FLOP per L2 byte (64 / 85 / 128) and grid granularity.
    // Pick the largest tile whose grid still fills whole rounds of workgroup slots (profiles/r01_gemm_tile_ab.txt).
    const int64_t zmul = (int64_t)d.batch * (d.split_k < 1 ? 1 : d.split_k);
    const int64_t t512 = (int64_t)ceil_div(d.M, 256) * ceil_div(d.N, 256) * zmul;
    const int64_t t256 = (int64_t)ceil_div(d.M, 256) * ceil_div(d.N, 128) * zmul;
    const int64_t t128 = (int64_t)ceil_div(d.M, 128) * ceil_div(d.N, 128) * zmul;
    auto eff = [](int64_t t, int slots) { return (double)t / (double)(((t + slots - 1) / slots) * slots); };
    const double e512 = (d.N >= 256 && t512 >= 192) ? eff(t512, 256) * 1.25 : 0.0;
    const double e256 = t256 >= 256 ? eff(t256, 512) * 1.07 : 0.0;
    const double e128 = eff(t128, 512);
    cfg = (e512 >= e256 && e512 >= e128) ? 512 : (e256 >= e128 ? 256 : 128);
    // K-major x K-major (Linear forward / dgrad): since the 16x16x32 MFMA switch the 128x128 kernel runs at about half the
    // per-FLOP rate of the 256-row kernels (16384x1024x4096: 268 vs 113 us), so a partly filled last round of big tiles beats
    // a well-filled grid of small ones: ViT's 768-wide outputs (300 tiles of 256x256 on 256 CUs) 253 -> 141 us
    // (profiles/r03_gemm_tile_sweep.txt).  Small grids (< 96 big tiles) stay on 128x128.  FFVC_TILE_RULE=0: previous rule.
    static int rule = -1;
    if (rule < 0) {
      const char* e = getenv("FFVC_TILE_RULE");
      rule = e ? atoi(e) : 1;
    }
    if (rule && d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_KMAJOR) {
      // 70+ big tiles: the 256x256 kernel even with a third of the CUs idle (6400x768x3072, 75 tiles: 68 vs 102 us; at 64 tiles and a
      // short reduction the small tile still wins: 8192x512x512 12 vs 17 us); 24..69 of
      // them with a long reduction: still the 256x256 kernel, launch2 splits it along K inside the launch (4928x512x6144, 40 tiles:
      // 60 vs 94 us; 3200x768x3072: 46 vs 50)
      const bool longk = d.K >= 2048 && d.batch == 1 && (d.split_k <= 1) && d.slab_stride == 0;
      if (d.N > 128 && (t512 >= 70 || (t512 >= 24 && longk))) cfg = 512;
      else if (t256 >= 192) cfg = 256;
      else cfg = 128;
      // small-M linears (VitGAN / x-transformer mappers at 16-32 samples per GPU: 512 rows x 1024..4096 columns): 128x128 tiles
      // leave half of the chip idle or need a deeper in-kernel split; 64-row tiles double the grid (two workgroups per CU).
      // FFVC_SMALLM=0: off (A/B)
      static int smallm = -1;
      if (smallm < 0) {
        const char* e = getenv("FFVC_SMALLM");
        smallm = e ? atoi(e) : 1;
      }
      if (smallm && cfg == 128 && d.M > 64 && d.M <= 2048 && t128 <= 160 && d.batch == 1 && (d.split_k <= 1) && d.slab_stride == 0) cfg = 64;
    }
    // short reductions are epilogue-dominated: keep two (smaller) workgroups per CU so one's epilogue overlaps the
    // other's K loop (threshold via FFVC_SHORTK for A/B runs)
    static int shortk = -1;
    if (shortk < 0) {
      const char* e = getenv("FFVC_SHORTK");
      shortk = e ? atoi(e) : 0;
    }
    if (d.K <= shortk && cfg != 128) cfg = 128;
  }
  if (d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_KMAJOR) {
    if (env_bm == 1) {         // no tile forced: the two-workgroups-per-CU kernel takes the launches it is built for (gemm3.hip)
      static int g3_env = -100;
      if (g3_env == -100) {
        const char* e = getenv("FFVC_GEMM3");
        g3_env = e ? atoi(e) : -1;   // default OFF: measured slower than the 256x256 kernel on every cfg2 shape (profiles/r06_gemm3_ab.txt)
      }
      const int r3 = ffvc_gemm3_try(d, st, vec_ok, g_opt_gemm3 != -100 ? g_opt_gemm3 : g3_env);
      if (r3 != 0) return r3;
    }
    return ffvc_gemm2_launch_kk(d, st, vec_ok, zero, cfg);
  }
  if (d.x_mode == FFVC_OP_CONV3X3 && d.w_mode == FFVC_OP_KMAJOR) {
    const int env_row = opt_value(g_opt_conv_row, "FFVC_CONV_ROW", 1);
    const int W = d.conv_W;
    const bool geom_ok = (W == 64 || W == 128 || (W >= 256 && W % 256 == 0)) && ((int64_t)d.conv_H * W) % 256 == 0 && d.batch == 1 &&
                         d.split_k <= 1 && (d.N % 128) == 0 && (d.M % 256) == 0 && d.act == FFVC_ACT_NONE &&
                         !(d.flags & (FFVC_F_MUL_ACT_GRAD | FFVC_F_WRITE_PREACT | FFVC_F_COLSUM));   // its epilogue class is GN-only
    const bool fills = (int64_t)(d.M / 256) * (d.N / 128) >= 256;
    // round 6: the pipelined row-tile kernel (conv3.hip) also beats the generic 256x256 implicit-GEMM tile on the 256-wide outputs
    // (two column tiles per pixel tile; 128^2 256->256 1054 -> 1149 TFLOP/s, 64^2 1118 -> 1198; the 512-wide ones lose 5 %:
    // profiles/r06_conv3_ab.txt), so with it the row tile takes every filling launch of up to 256 output channels
    static int row3_env = -100;
    if (row3_env == -100) {
      const char* e = getenv("FFVC_CONV_ROW3");
      row3_env = e ? atoi(e) : 1;
    }
    const int row3 = g_opt_conv_row3 != -100 ? g_opt_conv_row3 : row3_env;
    const bool wide_ok = row3 > 0 && d.N <= 256 && (d.conv_Cin % 64) == 0 && d.in_dtype != FFVC_F32 && vec_ok == 2;
    if (geom_ok && (env_row == 2 || (env_row == 1 && fills && (cfg != 512 || wide_ok)))) {
      const int tiles_n = d.N / 128, n_tiles = (d.M / 256) * tiles_n;
      constexpr int lds = 264 * 128 + 128 * 128 + 4 * 4096;
      static bool attr = false;
      static int use_buf = 1;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<uint16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<f16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const char* e = getenv("FFVC_DMA_BUF");
        use_buf = e ? atoi(e) : 1;
        attr = true;
      }
      const bool buf = use_buf && g8_offsets_ok<FFVC_OP_CONV3X3>(d);
      {
        if (row3 > 0 && buf) {       // round 6: the pipelined kernel (conv3.hip) takes every launch whose channels come in blocks of 64
          const int r3 = ffvc_conv3_try(d, st, vec_ok);
          if (r3 != 0) return r3;
        }
        if (d.flags & FFVC_F_GNB_SUMS) return 0;     // only conv3 folds the GroupNorm-backward statistics: never launch a kernel that would skip them
      }
      // epilogue class: forward convolutions accumulate the next GroupNorm's moments, dgrad convolutions do not -> lean
      const bool gnv = (d.flags & FFVC_F_GN_SUMS) != 0;
      constexpr int EG = ffvc_gemm_detail::EPI_GN, EL = ffvc_gemm_detail::EPI_LEAN;
      static bool attr2 = false;
      if (!attr2) {
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<uint16_t, true, EL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<f16_t, true, EL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr2 = true;
      }
      static int wpf = -1;
      if (wpf < 0) {
        const char* e = getenv("FFVC_CR_WPF");
        wpf = e ? atoi(e) : 0;
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<uint16_t, true, EL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<f16_t, true, EL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<uint16_t, true, EG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void*)conv_row_kernel<f16_t, true, EG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      }
      static int row2 = -1;
      if (row2 < 0) {
        const char* e = getenv("FFVC_CONV_ROW2");
        row2 = e ? atoi(e) : 0;     // measured (profiles/r03_conv_row_timing.txt): 635 vs 777 TFLOP/s at 256^2 128->128, step 148.8 vs 136.5 ms
                                    // -> the pipelined single-workgroup kernel is opt-in (FFVC_CONV_ROW2=1)
        constexpr int lds2 = 2 * 264 * 128 + 3 * 128 * 128;
        (void)hipFuncSetAttribute((const void*)conv_row2_kernel<uint16_t, EG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        (void)hipFuncSetAttribute((const void*)conv_row2_kernel<f16_t, EG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        (void)hipFuncSetAttribute((const void*)conv_row2_kernel<uint16_t, EL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        (void)hipFuncSetAttribute((const void*)conv_row2_kernel<f16_t, EL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
      }
      if (row2 && buf) {       // the software-pipelined kernel: one workgroup per CU
        constexpr int lds2 = 2 * 264 * 128 + 3 * 128 * 128;
        if (d.in_dtype == FFVC_F16) {
          if (!gnv) hipLaunchKernelGGL((conv_row2_kernel<f16_t, EL>), dim3(n_tiles), dim3(256), lds2, st, d, tiles_n, n_tiles, vec_ok);
          else hipLaunchKernelGGL((conv_row2_kernel<f16_t, EG>), dim3(n_tiles), dim3(256), lds2, st, d, tiles_n, n_tiles, vec_ok);
        } else {
          if (!gnv) hipLaunchKernelGGL((conv_row2_kernel<uint16_t, EL>), dim3(n_tiles), dim3(256), lds2, st, d, tiles_n, n_tiles, vec_ok);
          else hipLaunchKernelGGL((conv_row2_kernel<uint16_t, EG>), dim3(n_tiles), dim3(256), lds2, st, d, tiles_n, n_tiles, vec_ok);
        }
      } else if (wpf && buf) {     // double-buffered filter tile (FFVC_CR_WPF)
        if (d.in_dtype == FFVC_F16) {
          if (!gnv) hipLaunchKernelGGL((conv_row_kernel<f16_t, true, EL, true>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
          else hipLaunchKernelGGL((conv_row_kernel<f16_t, true, EG, true>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
        } else {
          if (!gnv) hipLaunchKernelGGL((conv_row_kernel<uint16_t, true, EL, true>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
          else hipLaunchKernelGGL((conv_row_kernel<uint16_t, true, EG, true>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
        }
      } else if (d.in_dtype == FFVC_F16) {
        if (buf && !gnv) hipLaunchKernelGGL((conv_row_kernel<f16_t, true, EL>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
        else if (buf) hipLaunchKernelGGL((conv_row_kernel<f16_t, true, EG>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
        else hipLaunchKernelGGL((conv_row_kernel<f16_t, false, EG>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
      } else {
        if (buf && !gnv) hipLaunchKernelGGL((conv_row_kernel<uint16_t, true, EL>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
        else if (buf) hipLaunchKernelGGL((conv_row_kernel<uint16_t, true, EG>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
        else hipLaunchKernelGGL((conv_row_kernel<uint16_t, false, EG>), dim3(n_tiles), dim3(256), lds, st, d, tiles_n, n_tiles, vec_ok, zero);
      }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        ffvc_set_error("conv_row launch failed: %s", hipGetErrorString(e));
        return -(int)e - 1000;
      }
      return 1;
    }
    if (d.flags & FFVC_F_GNB_SUMS) return 0;
    return ffvc_gemm2_launch_conv(d, st, vec_ok, zero, cfg);
  }
  if (d.x_mode == FFVC_OP_TRANS && d.w_mode == FFVC_OP_TRANS) {
    // measured: wgrad (both operands through ds_read_b64_tr_b16) runs 745 TFLOP/s on the register-staged kernel
    // (3 workgroups/CU) vs 551 on the 128x128 DMA tile -> keep it on gemm.hip unless forced
    // round 2 (profiles/r02_wgrad_ab.txt): with 256x256 tiles and a 4-way slab split-K the DMA path passes it on the big
    // square-ish weight gradients (4096x1024x16384: 856 vs 640 TFLOP/s); narrow outputs / short reductions stay on v1.
    // grouped weight gradients (grp_n layers in one launch, full K per tile) FIRST: only the 256x256 tile reads the per-entry offset
    // table, so neither a forced tile (FFVC_GEMM2_BM) nor any heuristic below may route them elsewhere — a descriptor the 256x256
    // kernel cannot take is refused (0 -> ffvc_gemm raises FFVC_E_UNSUPPORTED), never run on layer 0's operands
    if (d.grp_n > 0) {
      const bool gok = d.grp_n == d.batch && d.grp_n <= 8 && (d.M % 256) == 0 && (d.N % 256) == 0 && d.split_k <= 1 && d.slab_stride == 0 &&
                       vec_ok == 2;      // (launch2 itself picks buffer-descriptor or global-address DMA per operand range)
      return gok ? ffvc_gemm2_launch_tt(d, st, vec_ok, zero, 512) : 0;
    }
    if (env_bm == 128 || env_bm == 256 || env_bm == 512) return ffvc_gemm2_launch_tt(d, st, vec_ok, zero, cfg);
    // FFVC_F_SPLITK_INKERNEL: the caller asked for the kernel that combines its K slices itself — the 256x256 tile, whatever the
    // fill heuristic below thinks of the grid
    if ((d.flags & FFVC_F_SPLITK_INKERNEL) && d.split_k > 1 && env_bm == 1 && (d.M % 256) == 0 && (d.N % 256) == 0 && d.batch == 1)
      return ffvc_gemm2_launch_tt(d, st, vec_ok, zero, 512);
    if (env_bm == 1 && d.M >= 1024 && d.N >= 1024 && (d.M % 256) == 0 && (d.N % 256) == 0 && d.K >= 8192 && d.batch == 1 &&
        (int64_t)(d.M / 256) * (d.N / 256) * (d.split_k < 1 ? 1 : d.split_k) >= 192)
      return ffvc_gemm2_launch_tt(d, st, vec_ok, zero, 512);
    // short reductions (the weight gradients of the small-M mappers: 512 rows per GPU): the 128x128 LDS-DMA tile beats the
    // register-staged kernel there (4096x1024 from 512 rows: 15.6 vs 19.5 us, 1024x1024: 12.0 vs 17.0; profiles/r05_small_m_gemm.txt);
    // FFVC_SMALLM=0 keeps them on gemm.hip
    {
      static int smallm_tt = -1;
      if (smallm_tt < 0) {
        const char* e = getenv("FFVC_SMALLM_TT");
        if (!e) e = getenv("FFVC_SMALLM");
        smallm_tt = e ? atoi(e) : 1;
      }
      if (smallm_tt && env_bm == 1 && d.K <= 1024 && d.batch == 1 && d.split_k <= 1 && d.slab_stride == 0 && d.grp_n == 0 &&
          (int64_t)ceil_div(d.M, 128) * ceil_div(d.N, 128) >= 16)
        return ffvc_gemm2_launch_tt(d, st, vec_ok, zero, 128);
    }
    return 0;
  }
  if (d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_TRANS) return ffvc_gemm2_launch_nn(d, st, vec_ok, zero, cfg);
  if (d.x_mode == FFVC_OP_TRANS && d.w_mode == FFVC_OP_KMAJOR) return ffvc_gemm2_launch_tn(d, st, vec_ok, zero, cfg);
  return 0;
}

#ifdef FFVC_CR_TIMING
extern "C" int ffvc_debug_cr_stamps(unsigned long long* host_out) {   // debug builds only (tools/cr_timing.py)
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cr_stamps), sizeof(unsigned long long) * 4 * 6 * 8);
}
#endif
#ifdef FFVC_G8_TIMING
extern "C" int ffvc_debug_g8_stamps(unsigned long long* host_out) {   // debug builds only (tools/g8_timing.py)
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g8_stamps), sizeof(unsigned long long) * 128);
}
#endif

int g_force_gemm8 = 0;

extern "C" int ffvc_set_option(const char* name, int value) {
  if (!name) return FFVC_E_BADARG;
  if (!strcmp(name, "gemm8")) {
    g_force_gemm8 = value;
    return 0;
  }
  if (!strcmp(name, "conv_row3")) {
    g_opt_conv_row3 = value;
    return 0;
  }
  if (!strcmp(name, "gemm3")) {
    g_opt_gemm3 = value;
    return 0;
  }
  if (!strcmp(name, "gemm2_tile")) {
    g_opt_gemm2_tile = value;
    return 0;
  }
  if (!strcmp(name, "conv_row")) {
    g_opt_conv_row = value;
    return 0;
  }
  ffvc_set_error("ffvc_set_option: unknown option '%s'", name);
  return FFVC_E_BADARG;
}
