// gemm_probe.hip — standalone developer probe (no torch): candidate main loops for the NT GEMM  Y[M,N] = X[M,K] . W[N,K]^T
// (16-bit operands, fp32 accumulate) on gfx950, each checked against a naive fp32 kernel and timed in interleaved rounds.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/gemm_probe.hip -o tools/probe/gemm_probe
//   run  : tools/probe/gemm_probe [M N K] ...
// Kernel F ("one wave per SIMD"): 256x256 tile, 4 waves x (128x128), 512 registers per lane, BK = 32 stages in a 4-deep
// LDS-DMA ring with counted vmcnt, ONE barrier per 32 MFMAs, fragments fetched one 16-deep sub-step ahead.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <type_traits>
#include <vector>

typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void* lds_vp;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

#define HIPCHECK(x)                                                                      \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

template <bool BF>
__device__ __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (BF)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7FFFFF00, 0x00020000);
}
__device__ __forceinline__ void dma16(rsrc_t rs, uint32_t voff, uint32_t soff, unsigned char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vp)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ uint32_t pack_f16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 h2;
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel F.  VAR bits: 1 = no DMA inside the loop (timing only)   2 = no barrier / waits (timing only)
//                      4 = lookahead 3 instead of 2               8 = no sched hints
template <bool BF, int VAR>
__global__ __launch_bounds__(256, 1) void gemmF(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                uint16_t* __restrict__ Y, int M, int N, int K, int tiles_n, int n_tiles, unsigned long long* stamps) {
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int LA = (VAR & 4) ? 3 : 2;
  constexpr int STAGE = 32768, XT = 16384;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;
  const int l31 = lane & 31, h = lane >> 5;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;

  // ---- DMA state: waves 0,1 stage the X tile (rows 128*w ..), waves 2,3 the W tile; 8 pieces of 16 rows x 64 B per stage
  const bool isw = wid >= 2;
  const uint16_t* obase = isw ? W + (int64_t)n0 * K : X + (int64_t)m0 * K;
  const rsrc_t rs = make_rsrc(obase);
  uint32_t voff[8];
  {
    const int rowp = (lane >> 4) * 4 + ((lane & 15) >> 2);            // row inside the piece
    const int c = (lane & 3) ^ ((lane >> 4) & 3);                     // source k-chunk (8 elements) of this lane
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (wid & 1) * 128 + j * 16 + rowp;
      voff[j] = (uint32_t)(row * K + c * 8) * 2u;
    }
  }
  unsigned char* dma_base = smem + (isw ? XT : 0) + (wid & 1) * 8192;
  u32x4_t dummy[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dummy[j] = u32x4_t{0, 0, 0, 0};
  auto issue = [&](int stage_k, int buf, int j0, int j1) {   // pieces j0..j1-1 of K stage `stage_k` into ring slot `buf`
#pragma unroll
    for (int j = j0; j < j1; ++j) {
      if constexpr (VAR & 16) {          // timing only: the same request as a register load (no LDS write)
        dummy[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[j], (uint32_t)(stage_k * 64), 0);
      } else if constexpr (VAR & 32) {   // timing only: one lane per piece
        if (lane == 0) dma16(rs, voff[j], (uint32_t)(stage_k * 64), dma_base + buf * STAGE + j * 1024);
      } else {
        dma16(rs, voff[j], (uint32_t)(stage_k * 64), dma_base + buf * STAGE + j * 1024);
      }
    }
  };

  // ---- fragment addresses: byte = row * 64 + ((2 s + h) ^ ((row >> 2) & 3)) * 16
  uint32_t xa[2], wa[2];
  {
    const int rx = wm * 128 + l31, rw = wn * 128 + l31;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xa[s] = rx * 64 + (((2 * s + h) ^ ((rx >> 2) & 3)) << 4);
      wa[s] = XT + rw * 64 + (((2 * s + h) ^ ((rw >> 2) & 3)) << 4);
    }
  }
  f32x16_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  u32x4_t fw[2][4], fx[2][4];
  auto reads = [&](uint32_t sbase, int s, u32x4_t (&w)[4], u32x4_t (&x)[4]) {
    const unsigned char* pw = smem + (wa[s] + sbase);
    const unsigned char* px = smem + (xa[s] + sbase);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      w[b] = *(const u32x4_t*)(pw + b * 2048);
      x[b] = *(const u32x4_t*)(px + b * 2048);
    }
  };
  auto mmas = [&](const u32x4_t (&w)[4], const u32x4_t (&x)[4]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) mma<BF>(acc[a][b], w[a], x[b]);
  };
  auto hints = [&](bool rd, bool dma) {   // 16 MFMA | 8 DS reads right behind the first four | 4 LDS-DMA pieces spread behind
    if constexpr (!(VAR & 8)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (rd) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        if (dma) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    }
  };

  const int nk = K / 32;
  // prologue: stages 0 .. LA-1
#pragma unroll
  for (int s = 0; s < LA; ++s)
    if (s < nk) issue(s, s, 0, 8);
  if (nk < LA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (LA == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  reads(0, 0, fw[0], fx[0]);

  // one K step of 32: DMA = stage t + LA is issued (steady state), NEXT = the first fragments of stage t + 1 are fetched
  auto step = [&](int t, auto dma_tag, auto next_tag) {
    constexpr bool DMA = decltype(dma_tag)::value && !(VAR & 1), NEXT = decltype(next_tag)::value;
    const uint32_t sb = (uint32_t)(t & 3) << 15, sbn = (uint32_t)((t + 1) & 3) << 15;
    const int bl = (t + LA) & 3;
    // first half: MFMA (t, 0) | reads (t, 1) | 4 pieces of stage t + LA   (memory operations in the order of the hints)
    {
      const unsigned char* pw = smem + (wa[1] + sb);
      const unsigned char* px = smem + (xa[1] + sb);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr ((VAR & 64) != 0) {
          if (t == 0) {
            fw[1][b] = *(const u32x4_t*)(pw + b * 2048);
            fx[1][b] = *(const u32x4_t*)(px + b * 2048);
          }
        } else {
          fw[1][b] = *(const u32x4_t*)(pw + b * 2048);
          fx[1][b] = *(const u32x4_t*)(px + b * 2048);
        }
      }
      if constexpr (DMA) issue(t + LA, bl, 0, 4);
    }
    mmas(fw[0], fx[0]);
    hints(true, DMA);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(VAR & 2)) {
      // stage t + 1 landed (my pieces), every fragment read retired, then the workgroup barrier
      if constexpr (DMA) {
        if (LA == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // 4 ring slots, lookahead 2: a slot is rewritten two barriers after its last read
        else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // second half: MFMA (t, 1) | reads (t + 1, 0) | the other 4 pieces
    {
      const unsigned char* pw = smem + (wa[0] + sbn);
      const unsigned char* px = smem + (xa[0] + sbn);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if constexpr (NEXT && !(VAR & 64)) {
          fw[0][b] = *(const u32x4_t*)(pw + b * 2048);
          fx[0][b] = *(const u32x4_t*)(px + b * 2048);
        }
      }
      if constexpr (DMA) issue(t + LA, bl, 4, 8);
    }
    mmas(fw[1], fx[1]);
    hints(NEXT, DMA);
    __builtin_amdgcn_sched_barrier(0);
  };
  using T1 = std::true_type;
  using T0 = std::false_type;
  int t = 0;
  for (; t < nk - LA; ++t) step(t, T1{}, T1{});
  for (; t < nk - 1; ++t) step(t, T0{}, T1{});
  step(t, T0{}, T0{});

  if constexpr (VAR & 16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[0][0][j] += __uint_as_float(dummy[j][0] & 1u);
  }
  // ---- epilogue (probe): direct 8-byte stores, lane = row m, 4 consecutive n per register quad
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = m0 + wm * 128 + b * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 128 + a * 32 + 8 * q + 4 * h;
        u32x2_t o;
        if constexpr (BF) {
          o[0] = pack_bf16x2(acc[a][b][4 * q], acc[a][b][4 * q + 1]);
          o[1] = pack_bf16x2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        } else {
          o[0] = pack_f16x2(acc[a][b][4 * q], acc[a][b][4 * q + 1]);
          o[1] = pack_f16x2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
        }
        *(u32x2_t*)(Y + (int64_t)m * N + n) = o;
      }
    }
  if (blockIdx.x == 0 && threadIdx.x == 0 && stamps) {
    stamps[0] = __builtin_amdgcn_s_memtime() - st_c0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
}



// ------------------------------------------------------------------------------------------------------------------
// Kernel G = kernel F on v_mfma_f32_16x16x32 (the pure-MFMA loops above: the 16x16x32 shape sustains ~2.0 GHz where
// 32x32x16 sustains ~1.7 GHz under the power cap).  One K step of 32 = 64 MFMAs per wave, 16 fragment reads (all of the
// NEXT step, issued behind the barrier), 8 LDS-DMA pieces, one barrier.  VAR bits as kernel F (1, 2, 4).
template <bool BF>
__device__ __forceinline__ void mma16(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (BF)
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  else
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ void mma16a(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {   // accumulator pinned to the AGPR file
  if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ int swz4(int g) { return (0x78 >> (2 * (g & 3))) & 3; }   // 0,2,3,1: conflict-free 16-row b128 reads

template <bool BF, int VAR>
__global__ __launch_bounds__(256, 1) void gemmG(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                uint16_t* __restrict__ Y, int M, int N, int K, int tiles_n, int n_tiles, unsigned long long* stamps) {
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int LA = (VAR & 4) ? 3 : 2;
  constexpr int STAGE = 32768, XT = 16384;
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
  int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  if constexpr ((VAR & 32) != 0) {        // groups of 4 tile rows, column-major inside a group (tiles_m % 4 == 0 assumed)
    const int width = 4 * tiles_n, grp = tile / width, rem = tile - grp * width;
    tn = rem / 4;
    tm = grp * 4 + (rem - tn * 4);
  }
  const int m0 = tm * 256, n0 = tn * 256;

  const bool isw = wid >= 2;
  const int nkk = K / 32;
  const uint16_t* obase = (VAR & 16) ? (isw ? W : X) : (isw ? W + (int64_t)n0 * K : X + (int64_t)m0 * K);
  const rsrc_t rs = make_rsrc(obase);
  uint32_t voff[8];
  {
    const int rowp = (lane >> 4) * 4 + ((lane & 15) >> 2);
    const int c = (lane & 3) ^ swz4(lane >> 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (wid & 1) * 128 + j * 16 + rowp;
      voff[j] = (uint32_t)(row * K + c * 8) * 2u;
    }
  }
  unsigned char* dma_base = smem + (isw ? XT : 0) + (wid & 1) * 8192;
  auto issue = [&](int stage_k, int buf, int j0, int j1) {
#pragma unroll
    for (int j = j0; j < j1; ++j)
      dma16(rs, stage_k < nkk ? voff[j] : 0xFFFFFFF0u, stage_k < nkk ? (uint32_t)(stage_k * 64) : 0u, dma_base + buf * STAGE + j * 1024);
  };
  // fragment address: row = base + 16 blk + l15, chunk g4: byte = row * 64 + ((g4 ^ swz4(row >> 2)) << 4); + 1024 per block
  const int rx = wm * 128 + l15, rw = wn * 128 + l15;
  const uint32_t xa = rx * 64 + ((g4 ^ swz4(rx >> 2)) << 4);
  const uint32_t wa = XT + rw * 64 + ((g4 ^ swz4(rw >> 2)) << 4);

  f32x4_t acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t fw[2][8], fx[2][8];
  auto reads = [&](uint32_t sbase, u32x4_t (&w)[8], u32x4_t (&x)[8]) {
    const unsigned char* pw = smem + (wa + sbase);
    const unsigned char* px = smem + (xa + sbase);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      w[b] = *(const u32x4_t*)(pw + b * 1024);
      x[b] = *(const u32x4_t*)(px + b * 1024);
    }
  };
  const int nk = K / 32;
#pragma unroll
  for (int s = 0; s < LA; ++s)
    if (s < nk) issue(s, s, 0, 8);
  if (nk < LA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (LA == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  reads(0, fw[0], fx[0]);

  auto step = [&](int t, auto par_tag, auto dma_tag, auto next_tag) {
    constexpr int P = decltype(par_tag)::value;
    constexpr bool DMA = decltype(dma_tag)::value && !(VAR & 1), NEXT = decltype(next_tag)::value;
    const uint32_t sbn = (uint32_t)((t + 1) & 3) << 15;
    const int bl = (t + LA) & 3;
    // first half: W blocks 0..3 x all X blocks (32 MFMAs) | 4 pieces, one every 8 MFMAs
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if constexpr (DMA) issue(t + LA, bl, a, a + 1);
#pragma unroll
      for (int b = 0; b < 8; ++b) mma16a<BF>(acc[a][b], fw[P][a], fx[P][b]);
    }
    if constexpr (!(VAR & 2)) {
      if constexpr (DMA) {
        if (LA == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    // second half: W blocks 4..7 (32 MFMAs) | the 16 fragment reads of step t + 1, one per MFMA | the other 4 pieces
    {
      const unsigned char* pw = smem + (wa + sbn);
      const unsigned char* px = smem + (xa + sbn);
#pragma unroll
      for (int a = 4; a < 8; ++a) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          mma16a<BF>(acc[a][b], fw[P][a], fx[P][b]);
          if constexpr (NEXT) {
            if (a < 6) {
              const int i = (a - 4) * 8 + b;     // 0..15: X blocks first (all needed by the first MFMAs of the next step)
              if (i < 8) fx[P ^ 1][i] = *(const u32x4_t*)(px + i * 1024);
              else fw[P ^ 1][i - 8] = *(const u32x4_t*)(pw + (i - 8) * 1024);
            }
          }
        }
        if constexpr (DMA) issue(t + LA, bl, a, a + 1);
      }
    }
  };
  using T1 = std::true_type;
  using T0 = std::false_type;
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // one loop body for every step: beyond the last stage the pieces are fetched from an out-of-range offset (the DMA writes
  // zeros into a ring slot nobody reads) and the extra fragment reads of the last step are never used
  for (int t = 0; t < nk; t += 2) {
    step(t, P0{}, T1{}, T1{});
    step(t + 1, P1{}, T1{}, T1{});
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the last MFMAs' results before the VALU reads them
  // epilogue (probe): lane = row m (l15), 4 consecutive n per accumulator
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int m = m0 + wm * 128 + b * 16 + l15;
      const int n = n0 + wn * 128 + a * 16 + 4 * g4;
      u32x2_t o;
      if constexpr (BF) {
        o[0] = pack_bf16x2(acc[a][b][0], acc[a][b][1]);
        o[1] = pack_bf16x2(acc[a][b][2], acc[a][b][3]);
      } else {
        o[0] = pack_f16x2(acc[a][b][0], acc[a][b][1]);
        o[1] = pack_f16x2(acc[a][b][2], acc[a][b][3]);
      }
      *(u32x2_t*)(Y + (int64_t)m * N + n) = o;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0 && stamps) {
    stamps[0] = __builtin_amdgcn_s_memtime() - st_c0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Kernel R = the library's ring kernel (gemm2_kernel 256x256: 8 waves x (128 x 64), 2 waves / SIMD, BK = 64, 2-stage LDS-DMA
// ring, vmcnt(0) + barrier per K step, fragments one sub-step ahead), with the MFMA shape as a parameter:
//   MF = 32: v_mfma_f32_32x32x16 (what the library ships in round 2)    MF = 16: v_mfma_f32_16x16x32
template <bool BF, int MF>
__global__ __launch_bounds__(512, 2) void gemmR(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                uint16_t* __restrict__ Y, int M, int N, int K, int tiles_n, int n_tiles, unsigned long long* stamps) {
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int XT = 32768, STAGE = 65536;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid & 1, wn = wid >> 1;
  int tile;
  {
    const int bid = blockIdx.x;
    const int q = n_tiles >> 3, r = n_tiles & 7, xcd = bid & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 256, n0 = tn * 256;
  // DMA: K-major tile [256 rows][128 B]; piece p = lines 4p..4p+3 (a line = 2 rows), slot' = slot ^ (line & 15)
  const rsrc_t rsx = make_rsrc(X + (int64_t)m0 * K), rsw = make_rsrc(W + (int64_t)n0 * K);
  uint32_t voff[4];
  {
    const int line = 4 * wid + (lane >> 4);
    const int cp = (lane & 15) ^ (line & 15);
    const int r0 = 2 * line + (cp >> 3), kc = (cp & 7) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) voff[j] = (uint32_t)((r0 + 64 * j) * K + kc) * 2u;
  }
  auto issue_x = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsx, voff[j], (uint32_t)(kt * 128), smem + buf * STAGE + (8 * j + wid) * 1024);
  };
  auto issue_w = [&](int kt, int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(rsw, voff[j], (uint32_t)(kt * 128), smem + buf * STAGE + XT + (8 * j + wid) * 1024);
  };
  const int nk = K / 64;
  if constexpr (MF == 32) {
    const int l31 = lane & 31, h = lane >> 5;
    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    auto frag = [&](const unsigned char* t, int row, int sub) {
      const int line = row >> 1;
      const int cp = (((row & 1) << 3) | (2 * sub + h)) ^ (line & 15);
      return *(const u32x4_t*)(t + line * 256 + cp * 16);
    };
    issue_x(0, 0);
    issue_w(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned char* sX = smem + (kt & 1) * STAGE;
      const unsigned char* sW = sX + XT;
      const bool more = kt + 1 < nk;
      u32x4_t fa[2][2], fb[2][4];
      auto fetch = [&](int sub, u32x4_t (&a)[2], u32x4_t (&b)[4]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = frag(sW, wn * 64 + t * 32 + l31, sub);
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = frag(sX, wm * 128 + t * 32 + l31, sub);
      };
      fetch(0, fa[0], fb[0]);
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub < 3) fetch(sub + 1, fa[(sub + 1) & 1], fb[(sub + 1) & 1]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) mma<BF>(acc[a][b], fa[sub & 1][a], fb[sub & 1][b]);
        if (more && sub == 0) issue_x(kt + 1, (kt + 1) & 1);
        if (more && sub == 1) issue_w(kt + 1, (kt + 1) & 1);
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = m0 + wm * 128 + b * 32 + l31;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wn * 64 + a * 32 + 8 * q + 4 * h;
          u32x2_t o;
          o[0] = BF ? pack_bf16x2(acc[a][b][4 * q], acc[a][b][4 * q + 1]) : pack_f16x2(acc[a][b][4 * q], acc[a][b][4 * q + 1]);
          o[1] = BF ? pack_bf16x2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]) : pack_f16x2(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
          *(u32x2_t*)(Y + (int64_t)m * N + n) = o;
        }
      }
  } else {
    const int l15 = lane & 15, g4 = lane >> 4;
    f32x4_t acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    auto frag = [&](const unsigned char* t, int row, int sub32) {
      const int line = row >> 1;
      const int cp = (((row & 1) << 3) | (4 * sub32 + g4)) ^ (line & 15);
      return *(const u32x4_t*)(t + line * 256 + cp * 16);
    };
    issue_x(0, 0);
    issue_w(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned char* sX = smem + (kt & 1) * STAGE;
      const unsigned char* sW = sX + XT;
      const bool more = kt + 1 < nk;
      u32x4_t fa[2][4], fb[2][8];
      auto fetch = [&](int sub, u32x4_t (&a)[4], u32x4_t (&b)[8]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = frag(sW, wn * 64 + t * 16 + l15, sub);
#pragma unroll
        for (int t = 0; t < 8; ++t) b[t] = frag(sX, wm * 128 + t * 16 + l15, sub);
      };
      fetch(0, fa[0], fb[0]);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (sub < 1) fetch(1, fa[1], fb[1]);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
          for (int b = 0; b < 8; ++b) mma16<BF>(acc[a][b], fa[sub][a], fb[sub][b]);
          if (more && sub == 0 && a == 1) issue_x(kt + 1, (kt + 1) & 1);
          if (more && sub == 0 && a == 3) issue_w(kt + 1, (kt + 1) & 1);
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int m = m0 + wm * 128 + b * 16 + l15;
        const int n = n0 + wn * 64 + a * 16 + 4 * g4;
        u32x2_t o;
        o[0] = BF ? pack_bf16x2(acc[a][b][0], acc[a][b][1]) : pack_f16x2(acc[a][b][0], acc[a][b][1]);
        o[1] = BF ? pack_bf16x2(acc[a][b][2], acc[a][b][3]) : pack_f16x2(acc[a][b][2], acc[a][b][3]);
        *(u32x2_t*)(Y + (int64_t)m * N + n) = o;
      }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && stamps) {
    stamps[0] = __builtin_amdgcn_s_memtime() - st_c0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
}

// ---- pure MFMA loops (no LDS, no DMA, no barrier): what the matrix pipes deliver on random operands under the power cap.
// MODE 0: 32x32x16, 16 accumulators in rotation   1: 32x32x16, each accumulator twice in a row (k-inner)
//      2: 16x16x32, 64 accumulators in rotation   3: 16x16x32, k-inner pairs
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <int MODE>
__global__ __launch_bounds__(256, 1) void pure_mfma(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                    uint16_t* __restrict__ Y, int M, int N, int K, int tiles_n, int n_tiles,
                                                    unsigned long long* stamps) {
  unsigned long long st_c0 = 0, st_r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int tid = threadIdx.x;
  u32x4_t fa[2][8], fb[2][8];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      fa[s][i] = *(const u32x4_t*)(X + ((size_t)((blockIdx.x & 63) * 16 + s * 8 + i) * 256 + tid) * 8);
      fb[s][i] = *(const u32x4_t*)(W + ((size_t)((blockIdx.x & 63) * 16 + s * 8 + i) * 256 + tid) * 8);
    }
  const int nk = K / 32;
  float out = 0.f;
  if constexpr (MODE < 2) {
    f32x16_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    for (int t = 0; t < nk; ++t) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) mma<false>(acc[a][b], fa[s][a], fb[s][b]);
      } else {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) mma<false>(acc[a][b], fa[s][a], fb[s][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) out += acc[a][b][i];
  } else {
    f32x4v acc[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < nk; ++t) {
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int a2 = MODE == 2 ? a : a, b2 = b;
          acc[a2][b2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[0][a2]), __builtin_bit_cast(f16x8_t, fb[0][b2]), acc[a2][b2], 0, 0, 0);
          if constexpr (MODE == 3)
            acc[a2][b2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[1][a2]), __builtin_bit_cast(f16x8_t, fb[1][b2]), acc[a2][b2], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MODE == 3) ++t;   // two k32 halves per pass
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) out += acc[a][b][i];
  }
  Y[(size_t)blockIdx.x * 256 + tid] = (uint16_t)__float_as_uint(out);
  if (blockIdx.x == 0 && threadIdx.x == 0 && stamps) {
    stamps[0] = __builtin_amdgcn_s_memtime() - st_c0;
    stamps[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
}

// ---- naive reference: fp32 accumulate over the stored 16-bit operands --------------------------------------------------
template <bool BF>
__device__ __forceinline__ float ld16(const uint16_t* p) {
  if constexpr (BF) return __uint_as_float(((uint32_t)*p) << 16);
  else return (float)__builtin_bit_cast(f16_t, *p);
}
template <bool BF>
__global__ void ref_kernel(const uint16_t* X, const uint16_t* W, float* R, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= M) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += ld16<BF>(X + (int64_t)m * K + k) * ld16<BF>(W + (int64_t)n * K + k);
  R[(int64_t)m * N + n] = s;
}
template <bool BF>
__global__ void cmp_kernel(const uint16_t* Y, const float* R, int64_t n, float* maxerr, float* maxref) {
  float e = 0.f, r = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    e = fmaxf(e, fabsf(ld16<BF>(Y + i) - R[i]));
    r = fmaxf(r, fabsf(R[i]));
  }
  atomicMax((unsigned int*)maxerr, __float_as_uint(e));
  atomicMax((unsigned int*)maxref, __float_as_uint(r));
}

// ---- host ------------------------------------------------------------------------------------------------------------
static uint16_t f2h(float f, bool bf) {
  if (bf) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
  }
  f16_t hh = (f16_t)f;
  uint16_t r;
  memcpy(&r, &hh, 2);
  return r;
}
static uint64_t rng_state = 0x1234567887654321ull;
static inline float urand() {   // uniform [-1, 1)
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (float)((rng_state >> 40) * (1.0 / 8388608.0)) - 1.0f;
}

typedef void (*kern_t)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, int, int, unsigned long long*);
struct Variant {
  const char* name;
  kern_t fn;
  int threads, lds;
  bool exact;    // results expected correct
};

int main(int argc, char** argv) {
  const bool bf = getenv("PROBE_BF16") != nullptr;
  std::vector<Variant> vars;
  constexpr int LDSF = 4 * 32768;
#define ADDF(name, BFv, VARv, exact) vars.push_back({name, (kern_t)gemmF<BFv, VARv>, 256, LDSF, exact})
  if (bf) {
    ADDF("F bf16 la2", true, 0, true);
    ADDF("F bf16 la3", true, 4, true);
  } else {
    ADDF("F la2", false, 0, true);
    ADDF("F la3", false, 4, true);
    vars.push_back({"R ring 32x32x16", (kern_t)gemmR<false, 32>, 512, 131072, true});
    vars.push_back({"R ring 16x16x32", (kern_t)gemmR<false, 16>, 512, 131072, true});
    vars.push_back({"G la2", (kern_t)gemmG<false, 0>, 256, LDSF, true});
    vars.push_back({"G la3", (kern_t)gemmG<false, 4>, 256, LDSF, true});
    vars.push_back({"G la3 gm4", (kern_t)gemmG<false, 36>, 256, LDSF, true});
    vars.push_back({"G la3 same-tile src*", (kern_t)gemmG<false, 20>, 256, LDSF, false});
    vars.push_back({"G la2 noDMA*", (kern_t)gemmG<false, 1>, 256, LDSF, false});
    vars.push_back({"G la2 nobar*", (kern_t)gemmG<false, 2>, 256, LDSF, false});
    ADDF("F la2 noDMA*", false, 1, false);
    ADDF("F pure MFMA*", false, 67, false);
    vars.push_back({"pure 32x32x16 rot*", (kern_t)pure_mfma<0>, 256, 0, false});
    vars.push_back({"pure 32x32x16 kinner*", (kern_t)pure_mfma<1>, 256, 0, false});
    vars.push_back({"pure 16x16x32 rot*", (kern_t)pure_mfma<2>, 256, 0, false});
    vars.push_back({"pure 16x16x32 kinner*", (kern_t)pure_mfma<3>, 256, 0, false});
  }
  for (auto& v : vars) if (v.lds) HIPCHECK(hipFuncSetAttribute((const void*)v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));

  unsigned long long* dstamps;
  HIPCHECK(hipMalloc(&dstamps, 16));
  HIPCHECK(hipMemset(dstamps, 0, 16));
  std::vector<std::array<int, 3>> shapes;
  for (int i = 1; i + 2 < argc; i += 3) shapes.push_back({atoi(argv[i]), atoi(argv[i + 1]), atoi(argv[i + 2])});
  if (shapes.empty()) shapes = {{4096, 4096, 4096}, {8192, 8192, 8192}, {16384, 4096, 1024}, {16384, 1024, 4096}};

  // correctness on a small multi-tile problem
  {
    const int M = 512, N = 768, K = 1024;
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
    for (auto& v : hx) v = f2h(urand(), bf);
    for (auto& v : hw) v = f2h(urand(), bf);
    uint16_t *dx, *dw, *dy;
    float *dr, *dm;
    HIPCHECK(hipMalloc(&dx, hx.size() * 2));
    HIPCHECK(hipMalloc(&dw, hw.size() * 2));
    HIPCHECK(hipMalloc(&dy, (size_t)M * N * 2));
    HIPCHECK(hipMalloc(&dr, (size_t)M * N * 4));
    HIPCHECK(hipMalloc(&dm, 8));
    HIPCHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    if (bf) hipLaunchKernelGGL(ref_kernel<true>, dim3((N + 255) / 256, M), dim3(256), 0, 0, dx, dw, dr, M, N, K);
    else hipLaunchKernelGGL(ref_kernel<false>, dim3((N + 255) / 256, M), dim3(256), 0, 0, dx, dw, dr, M, N, K);
    for (auto& v : vars) {
      if (!v.exact) continue;
      for (int rep = 0; rep < 3; ++rep) {
        HIPCHECK(hipMemset(dy, 0xff, (size_t)M * N * 2));
        HIPCHECK(hipMemset(dm, 0, 8));
        const int tn = N / 256, nt = (M / 256) * tn;
        hipLaunchKernelGGL(v.fn, dim3(nt), dim3(v.threads), v.lds, 0, dx, dw, dy, M, N, K, tn, nt, dstamps);
        if (bf) hipLaunchKernelGGL(cmp_kernel<true>, dim3(256), dim3(256), 0, 0, dy, dr, (int64_t)M * N, dm, dm + 1);
        else hipLaunchKernelGGL(cmp_kernel<false>, dim3(256), dim3(256), 0, 0, dy, dr, (int64_t)M * N, dm, dm + 1);
        float hm[2];
        HIPCHECK(hipMemcpy(hm, dm, 8, hipMemcpyDeviceToHost));
        const bool ok = hm[0] <= hm[1] * (bf ? 1.2e-2f : 2e-3f) && hm[1] > 1.0f;
        if (rep == 0 || !ok) printf("check %-22s maxerr %.4g (max |ref| %.4g) %s\n", v.name, hm[0], hm[1], ok ? "OK" : "FAIL");
      }
    }
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dr); hipFree(dm);
  }

  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
    for (auto& v : hx) v = f2h(urand(), bf);
    for (auto& v : hw) v = f2h(urand(), bf);
    uint16_t *dx, *dw, *dy;
    HIPCHECK(hipMalloc(&dx, hx.size() * 2));
    HIPCHECK(hipMalloc(&dw, hw.size() * 2));
    HIPCHECK(hipMalloc(&dy, (size_t)M * N * 2));
    HIPCHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const int tn = N / 256, nt = (M / 256) * tn;
    const int rounds = 7, iters = 10;
    std::vector<std::vector<float>> ms(vars.size());
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (int r = 0; r < rounds + 1; ++r)
      for (size_t vi = 0; vi < vars.size(); ++vi) {
        auto& v = vars[vi];
        HIPCHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(v.fn, dim3(nt), dim3(v.threads), v.lds, 0, dx, dw, dy, M, N, K, tn, nt, dstamps);
        HIPCHECK(hipEventRecord(e1, 0));
        HIPCHECK(hipEventSynchronize(e1));
        float t;
        HIPCHECK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) ms[vi].push_back(t / iters);
      }
    printf("shape %d x %d x %d (%s, uniform [-1,1))\n", M, N, K, bf ? "bf16" : "f16");
    for (size_t vi = 0; vi < vars.size(); ++vi) {
      std::sort(ms[vi].begin(), ms[vi].end());
      const double med = ms[vi][ms[vi].size() / 2], mn = ms[vi][0];
      const double fl = 2.0 * M * N * K;
      hipLaunchKernelGGL(vars[vi].fn, dim3(nt), dim3(vars[vi].threads), vars[vi].lds, 0, dx, dw, dy, M, N, K, tn, nt, dstamps);
      unsigned long long hs[2];
      HIPCHECK(hipMemcpy(hs, dstamps, 16, hipMemcpyDeviceToHost));
      const double mfma_cyc = (double)(K / 16) * 16 * 32 * (vars[vi].threads == 512 ? 0.5 : 1.0);   // per wave and tile: K/16 sub-steps x 16 MFMAs x 32 cycles
      printf("  %-22s median %8.1f TF  best %8.1f TF  (%.3f ms)  tile0: %llu cyc, %.2f GHz, MFMA busy %.1f %%\n", vars[vi].name,
             fl / med * 1e-9, fl / mn * 1e-9, med, hs[0], hs[0] / (hs[1] * 10.0), 100.0 * mfma_cyc / hs[0]);
    }
    hipFree(dx); hipFree(dw); hipFree(dy);
  }
  return 0;
}
