// gemm3.hip — 256x128 LDS-DMA ring kernel, TWO workgroups per CU, for the epilogue-heavy K-major x K-major launches (round 6).
//
// Why.  The 256x256 ring kernel (gemm2_kernel) owns a CU: 8 waves = 2 per SIMD run the same schedule in lock step, so a tile is
// [K loop with the matrix pipes busy] then [epilogue with the matrix pipes idle].  For the launches whose epilogue is long next to a
// short reduction — the channel-MLP fc1 forward (bias + GELU + act' = two 16-bit tensors out, K = 1024), its aux-multiply dgrad,
// the projections back into the fp32 residual stream (fp32 read + fp32 write), ViT's QuickGELU kinds — the epilogue is 35-45 % of
// the launch (profiles/r05_isolated_sum.txt: 16384x4096x1024 173-179 us against 101 us for the plain kernel on the same M.N.K).
// Nothing inside ONE workgroup can hide it: the accumulators of the finished tile fill the register file.
//
// What.  Half the tile (256 x 128, 4 waves = ONE per SIMD, wave tile 128 x 64 as before), a 3-slot ring of 32-deep stages
// (3 x 24 KiB = 72 KiB) and the register-exchange epilogue (gemm_epilogue_perm16: no LDS pads), so that TWO workgroups fit a CU
// (144 of 160 KiB LDS, 2 x 256 registers per SIMD lane).  The two are independent programs: while one runs its epilogue (VALU,
// global loads / stores) the other's wave on the same SIMD keeps the matrix pipe fed.  Equal tiles would keep the pair in lock step
// (both in the K loop, then both in the epilogue), so the workgroups that take a CU's SECOND slot in the first round start late by
// about half a K loop (`stagger`, 100 MHz ticks): from then on the pair alternates.
//
// K loop (tools/probe/gemm_probe.hip kernel G, re-cut for a 3-slot ring): one 32-deep step = 32 v_mfma_f32_16x16x32 per wave,
// fragments of step t + 1 fetched under the second half of step t's MFMAs, the six 1-KiB LDS-DMA pieces of stage t + 2 issued
// three per half, ONE barrier per step with a counted vmcnt (the three youngest pieces may still be in flight).
// LDS image of a stage: [X 256 rows | W 128 rows] x 64 B; 16-byte chunk c of row r sits at slot c ^ swz4(r >> 2).
#include "gemm2_kernels.h"

namespace {

__device__ __forceinline__ int g3_swz4(int g) { return (0x78 >> (2 * (g & 3))) & 3; }   // 0,2,3,1: conflict-free 16-row b128 reads

constexpr int G3_BM = 256, G3_BN = 128, G3_BK = 32;
constexpr int G3_XT = G3_BM * 64, G3_WT = G3_BN * 64, G3_STAGE = G3_XT + G3_WT;        // bytes: 16 KiB + 8 KiB
constexpr int G3_SLOTS = 3;
constexpr int G3_LDS = G3_SLOTS * G3_STAGE;                                              // 72 KiB

template <typename L, int EPI>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(const ffvc_gemm_desc p, int tiles_n, int n_tiles, int gm, int stagger) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;
  const int l15 = lane & 15, g4 = lane >> 4;
  if (stagger > 0 && blockIdx.y == 0) {
    // workgroups are dealt round-robin to the 8 XCDs and then to an XCD's 32 CUs: the second slot of every CU is filled by the
    // workgroups 32..63 of that XCD's sequence
    const int k = blockIdx.x >> 3;
    if (k >= 32 && k < 64) {
      const uint64_t t0 = wall_clock64();
      while ((int64_t)(wall_clock64() - t0) < stagger) __builtin_amdgcn_s_sleep(16);
    }
  }
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int tm, tn;
  if (gm > 1) {     // groups of gm tile rows, column-major inside a group (see gemm2_kernel)
    const int width = gm * tiles_n;
    const int grp = tile / width, rem = tile - grp * width;
    const int first = grp * gm;
    const int gsz = min(n_tiles / tiles_n - first, gm);
    tn = rem / gsz;
    tm = first + (rem - tn * gsz);
  } else {
    tm = tile / tiles_n;
    tn = tile - tm * tiles_n;
  }
  const int m0 = tm * G3_BM, n0 = tn * G3_BN;
  const int z = blockIdx.y;
  const int zo = z / p.batch_inner, zi = z - zo * p.batch_inner;
  const uint16_t* xb = (const uint16_t*)p.x + zo * p.xbo + zi * p.xbi;
  const uint16_t* wb = (const uint16_t*)p.w + zo * p.wbo + zi * p.wbi;

  // ---- DMA: 24 pieces of 16 rows x 64 B per stage, six per wave: pieces 6 w .. 6 w + 5 of [X 0..15 | W 16..23]
  //      (waves 0, 1 and the first four pieces of wave 2 stage X rows, the rest W rows); lane -> row (lane >> 2) of the piece,
  //      source k-chunk (lane & 3) ^ swz4(row >> 2)
  const rsrc_t rsx = make_rsrc(xb + (int64_t)m0 * p.ldx), rsw = make_rsrc(wb + (int64_t)n0 * p.ldw);
  uint32_t voff[6];
  {
    const int rowp = lane >> 2;
    const int c = (lane & 3) ^ g3_swz4(lane >> 4);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int pc = 6 * wid + j;
      if (pc < 16) {
        const int row = pc * 16 + rowp;
        voff[j] = (m0 + row < p.M) ? (uint32_t)(((int64_t)row * p.ldx + c * 8) * 2) : DMA_OOB;
      } else {
        const int row = (pc - 16) * 16 + rowp;
        voff[j] = (n0 + row < p.N) ? (uint32_t)(((int64_t)row * p.ldw + c * 8) * 2) : DMA_OOB;
      }
    }
  }
  const int nk = p.K / G3_BK;
  // piece j of K stage `st` into ring slot `slot`; stages beyond the reduction fetch from an out-of-range offset (the DMA writes
  // zeros into a slot nobody reads) so the loop has ONE body
  auto issue1 = [&](int st, int slot, int j) {
    const int pc = 6 * wid + j;                                   // wave-uniform
    const bool live = st < nk;
    const uint32_t so = live ? (uint32_t)(st * (G3_BK * 2)) : 0u;
    const uint32_t vo = live ? voff[j] : DMA_OOB;
    dma16bs(pc < 16 ? rsx : rsw, vo, so, smem + slot * G3_STAGE + pc * 1024);
  };

  // ---- fragments: row = base + 16 blk + l15, k-chunk g4 -> byte row * 64 + ((g4 ^ swz4(row >> 2)) << 4); + 1024 per 16-row block
  const int rx = wm * 128 + l15, rw = wn * 64 + l15;
  const uint32_t xa = rx * 64 + ((g4 ^ g3_swz4(rx >> 2)) << 4);
  const uint32_t wa = G3_XT + rw * 64 + ((g4 ^ g3_swz4(rw >> 2)) << 4);

  f32x4_t acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t fw[2][4], fx[2][8];

  // prologue: stages 0 and 1
#pragma unroll
  for (int j = 0; j < 6; ++j) issue1(0, 0, j);
#pragma unroll
  for (int j = 0; j < 6; ++j) issue1(1, 1, j);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const unsigned char* px = smem + xa;
    const unsigned char* pw = smem + wa;
#pragma unroll
    for (int b = 0; b < 8; ++b) fx[0][b] = *(const u32x4_t*)(px + b * 1024);
#pragma unroll
    for (int a = 0; a < 4; ++a) fw[0][a] = *(const u32x4_t*)(pw + a * 1024);
  }

  auto step = [&](int t, int slot_next, int slot_dma, auto par_tag) {
    constexpr int P = decltype(par_tag)::value;
    // first half: W blocks 0, 1 x all X blocks (16 MFMAs) | three pieces of stage t + 2
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        mma16_lo<L>(acc[a][b], fw[P][a], fx[P][b]);
        if (b == 2 || b == 6) {
          const int j = a * 2 + (b == 6);
          if (j < 3) issue1(t + 2, slot_dma, j);
        }
      }
    }
    // stage t + 1 has landed (my pieces: everything but the three youngest), then everyone's
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // second half: W blocks 2, 3 (16 MFMAs) | the 12 fragment reads of step t + 1 | the other three pieces
    {
      const unsigned char* px = smem + (xa + slot_next * G3_STAGE);
      const unsigned char* pw = smem + (wa + slot_next * G3_STAGE);
#pragma unroll
      for (int a = 2; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          mma16_lo<L>(acc[a][b], fw[P][a], fx[P][b]);
          const int i = (a - 2) * 8 + b;                    // 0..15
          if (i < 8) fx[P ^ 1][i] = *(const u32x4_t*)(px + i * 1024);
          else if (i < 12) fw[P ^ 1][i - 8] = *(const u32x4_t*)(pw + (i - 8) * 1024);
          if (i == 3 || i == 7 || i == 11) issue1(t + 2, slot_dma, 3 + i / 4);
        }
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // slot of stage s = s % 3, tracked incrementally (no division in the loop)
  int s1 = 1, s2 = 2;                      // slots of stages t + 1, t + 2
  int t = 0;
  for (; t + 1 < nk; t += 2) {
    step(t, s1, s2, P0{});
    { const int n = s2 == 2 ? 0 : s2 + 1; s1 = s2; s2 = n; }
    step(t + 1, s1, s2, P1{});
    { const int n = s2 == 2 ? 0 : s2 + 1; s1 = s2; s2 = n; }
  }
  if (t < nk) {                            // odd number of stages: the last one, fragments in set 0
    step(t, s1, s2, P0{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-fill pieces of the stages beyond the reduction must not outlive the workgroup
  int lane_e = lane;                                   // (opaque copy: keeps the epilogue's addressing out of the K loop's live ranges)
  asm volatile("" : "+v"(lane_e));
  ffvc_gemm_detail::gemm_epilogue_perm16<L, 4, EPI>(p, acc, m0, n0, wm, wn, lane_e, zo, zi, 0);
}

template <typename L, int EPI>
int g3_launch(const ffvc_gemm_desc& d, hipStream_t st, int gm, int stagger) {
  const int tiles_m = ceil_div(d.M, G3_BM), tiles_n = ceil_div(d.N, G3_BN);
  const int n_tiles = tiles_m * tiles_n;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm3_kernel<L, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, G3_LDS);
    attr = true;
  }
  if (gm > tiles_m) gm = tiles_m;
  hipLaunchKernelGGL((gemm3_kernel<L, EPI>), dim3(n_tiles, d.batch), dim3(256), G3_LDS, st, d, tiles_n, n_tiles, gm, stagger);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    ffvc_set_error("gemm3 launch failed: %s", hipGetErrorString(e));
    return -(int)e - 1000;
  }
  return 1;
}

template <typename L>
int g3_dispatch(const ffvc_gemm_desc& d, hipStream_t st, int gm, int stagger) {
  using namespace ffvc_gemm_detail;
  const bool plain_out = !d.residual && !(d.flags & (FFVC_F_OUT_F32 | FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT)) && d.slab_stride == 0 && d.alpha == 1.0f;
  const bool wants_act = d.act != FFVC_ACT_NONE || (d.flags & (FFVC_F_MUL_ACT_GRAD | FFVC_F_WRITE_PREACT | FFVC_F_COLSUM));
  if (d.flags & (FFVC_F_GN_SUMS | FFVC_F_ATOMIC_OUT | FFVC_F_ACCUM_OUT | FFVC_F_BIAS_ALONG_M | FFVC_F_SPLITK_INKERNEL)) return 0;
  if (d.slab_stride != 0 || d.alpha != 1.0f) return 0;
  if (!wants_act) {
    if (!d.residual && !(d.flags & FFVC_F_OUT_F32)) return g3_launch<L, EPI_LEAN | EPI_O_T>(d, st, gm, stagger);
    if (d.residual && (d.flags & FFVC_F_RES_F32) && (d.flags & FFVC_F_OUT_F32)) return g3_launch<L, EPI_LEAN | EPI_O_F32R>(d, st, gm, stagger);
    return 0;
  }
  if (!(d.act == FFVC_ACT_GELU || d.act == FFVC_ACT_QUICKGELU) || !plain_out) return 0;
  const bool bwd = d.flags & FFVC_F_MUL_ACT_GRAD;
  const bool bias_ok = bwd ? d.bias == nullptr : (d.bias != nullptr);
  if (!bias_ok) return 0;
  if (d.flags & FFVC_F_AUX_ACTGRAD) {
    if (bwd && !(d.flags & FFVC_F_WRITE_PREACT)) return g3_launch<L, EPI_K_MULAUX>(d, st, gm, stagger);      // (+ optional column sums)
    if (!bwd && (d.flags & FFVC_F_WRITE_PREACT) && !(d.flags & FFVC_F_COLSUM)) {
      const int r = d.act == FFVC_ACT_GELU ? g3_launch<L, EPI_K_GELU_FWDG>(d, st, gm, stagger) : g3_launch<L, EPI_K_QGELU_FWDG>(d, st, gm, stagger);
      return r == 1 ? 2 : r;                 // 2 = aux already holds the derivative (no conversion pass needed)
    }
    return 0;
  }
  if (!bwd && !(d.flags & FFVC_F_COLSUM)) {
    if (d.act == FFVC_ACT_GELU) return g3_launch<L, EPI_K_GELU_FWD>(d, st, gm, stagger);
    return g3_launch<L, EPI_K_QGELU_FWD>(d, st, gm, stagger);
  }
  if (bwd && !(d.flags & FFVC_F_WRITE_PREACT)) {
    if (d.act == FFVC_ACT_GELU) return g3_launch<L, EPI_K_GELU_BWD>(d, st, gm, stagger);
    return g3_launch<L, EPI_K_QGELU_BWD>(d, st, gm, stagger);
  }
  return 0;
}

}  // namespace

// 0 = not taken (the caller carries on with the 256x256 / 128x128 kernels), 1 = launched, 2 = launched and aux holds act'(pre),
// < 0 = launch error.  mode: 0 = heuristic, 1 = every eligible launch (tests / A-B), -1 = never.
int ffvc_gemm3_try(const ffvc_gemm_desc& d, hipStream_t st, int vec_ok, int mode) {
  if (mode < 0) return 0;
  if (!(d.x_mode == FFVC_OP_KMAJOR && d.w_mode == FFVC_OP_KMAJOR) || d.in_dtype == FFVC_F32) return 0;
  if (vec_ok != 2 || d.kseg != 0 || d.x_mi != 0 || d.split_k > 1 || (d.K % G3_BK) != 0 || d.K < 2 * G3_BK) return 0;
  if (!dma_operand_ok<FFVC_OP_KMAJOR>(d, true) || !dma_operand_ok<FFVC_OP_KMAJOR>(d, false)) return 0;
  const int64_t tiles = (int64_t)ceil_div(d.M, G3_BM) * ceil_div(d.N, G3_BN) * d.batch;
  static int n_cu = 0, stagger_env = -2, gm_env = -1;
  if (!n_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    const char* e = getenv("FFVC_G3_STAGGER");        // 100 MHz ticks; -1 (default) = half a K loop, from K; 0 = off
    stagger_env = e ? atoi(e) : -1;
    const char* g = getenv("FFVC_G3_GM");
    gm_env = g ? atoi(g) : 0;
  }
  if (mode == 0) {
    // the launches this kernel is for: at least one full round of two workgroups per CU, and an epilogue worth hiding —
    // an activation kind, the aux multiply, or the fp32 residual projection
    const bool heavy = d.act != FFVC_ACT_NONE || (d.flags & (FFVC_F_MUL_ACT_GRAD | FFVC_F_WRITE_PREACT)) ||
                       (d.residual && (d.flags & FFVC_F_RES_F32) && (d.flags & FFVC_F_OUT_F32));
    if (!heavy || tiles < 2 * n_cu) return 0;
  }
  // half a K loop of one workgroup running alone: 256 x 128 x K x 2 FLOP on a CU's 4 x 1024 FLOP/clk at ~2 GHz and ~65 % busy
  int stagger = stagger_env;
  if (stagger < 0) stagger = (int)((double)G3_BM * G3_BN * d.K * 2.0 / (4096.0 * 0.65) / 2.0e9 * 1.0e8 * 0.5);
  if (tiles < 2 * n_cu) stagger = 0;
  const int gm = gm_env > 0 ? gm_env : 1;
  if (d.in_dtype == FFVC_F16) return g3_dispatch<f16_t>(d, st, gm, stagger);
  return g3_dispatch<uint16_t>(d, st, gm, stagger);
}
