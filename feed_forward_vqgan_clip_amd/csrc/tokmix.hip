// tokmix.hip — fused token-mixing MLP of the MLP-Mixer mapper (mlp_mixer_pytorch.py:28,34: Conv1d(k=1) over the token
// axis -> GELU -> Conv1d(k=1), inside PreNormResidual :7-14):
//
//     y[b] = W2 @ gelu(W1 @ xn[b] + b1) + b2 + res[b]          xn[b]: [T, D]   W1: [O, T]   W2: [T, O]   (O = 4T)
//
// As two batched GEMMs this is the worst-shaped work of the step: K = T = 256 is only 4 K-steps, so the first GEMM is all
// prologue + epilogue (142 TFLOP/s, round 1) and the hidden activation [B, O, D] (2 x 134 MB per layer with its
// pre-activation) crosses HBM four times per step.  Here ONE workgroup owns (sample b, 256 columns of D):
//
//   * the 256 x 256 slice of xn[b] is staged once (LDS-DMA, transposed reads) into MFMA B fragments that stay in
//     registers for the whole kernel (T/16 fragments = 64 VGPRs at T = 256): every wave owns 32 columns;
//   * the hidden dimension is walked in chunks of 32 rows: W1[32, T] and W2[T, 32] chunks stream through a 4-deep LDS ring
//     (counted vmcnt, one barrier per chunk); GEMM-1 gives the 32 x 32 hidden block in accumulators, bias + erf-GELU are
//     applied in registers, and — since MFMA sums over k in any order as long as both operands agree — the accumulator
//     registers ARE the B fragment of GEMM-2 after four v_permlane32_swap (which make each lane's 8 hidden rows
//     consecutive, so the W2 fragment is a plain 16-byte K-major read).  The hidden activation never leaves the CU;
//   * the output tile goes through LDS once so that residual reads / y stores are full 1 KiB rows (16 B per lane).
//
// Backward recomputes the hidden block instead of reading it back (`tokmix_bwd_hidden_kernel`): h = gelu(W1 xn + b1) and
// dh = (W2^T dy) * gelu'(W1 xn + b1) are produced chunk by chunk and written once (16-bit) for the weight-gradient GEMMs;
// dx = W1^T dh stays a plain batched GEMM (K = O is long).  Nothing of the forward is saved besides xn.
//
// LDS images (16-byte slots, source-side XOR swizzle as in gemm2.hip; verified conflict-free for T in {128, 256}):
//   W1 / W2^T chunk [32 rows][T]:  slot' = slot ^ (row & 15)
//   W2 chunk        [T rows][32]:  4 slots per row, 4 rows per 256-B line:  q' = q ^ (line & 15)
#include "gemm2_kernels.h"

namespace {

constexpr int TM_DT = 256;   // D columns per workgroup (8 waves x 32)
constexpr int TM_OC = 32;    // hidden rows per chunk
constexpr int TM_NST = 4;    // ring depth

template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 12 && (N % 2) == 0, "add the vmcnt literal");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
}
typedef const __attribute__((address_space(4))) float* const_f32p;   // constant address space: uniform loads become s_load

// per-thread DMA sources of one [32 rows][T] K-major chunk (row stride `ld` elements) and of one [T rows][32] chunk, as byte
// offsets against a buffer descriptor of the weight matrix; the chunk index travels in the scalar offset
template <int T_>
struct ChunkDma {
  static constexpr int NP = T_ / 128;          // 1-KiB pieces per thread per chunk half (512 threads x 16 B = 8 KiB)
  uint32_t offA[NP];                           // [32][T] image
  uint32_t offB[NP];                           // [T][32] image
  int w;
  __device__ __forceinline__ void init(int tid, int ldA, int ldB) {
    const int lane = tid & 63;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int SPR = T_ / 8;                // 16-byte slots per [32][T] row
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int p = (8 * j + w) * 64 + lane;
      const int row = p / SPR, slot = (p % SPR) ^ (row & 15);
      offA[j] = (uint32_t)(row * ldA + slot * 8) * 2u;
      const int e = (p & ~15) | ((p & 15) ^ ((p >> 4) & 15));
      offB[j] = (uint32_t)((e >> 2) * ldB + (e & 3) * 8) * 2u;
    }
  }
  // rows [32c, 32c+32) of a K-major [rows][T] matrix: soff = c * 32 * T * 2 bytes
  __device__ __forceinline__ void issueA(unsigned char* dst, rsrc_t rs, uint32_t soff) const {
#pragma unroll
    for (int j = 0; j < NP; ++j) dma16bs(rs, offA[j], soff, dst + (8 * j + w) * 1024);
  }
  // columns [32c, 32c+32) of a K-major [T][cols] matrix: soff = c * 32 * 2 bytes
  __device__ __forceinline__ void issueB(unsigned char* dst, rsrc_t rs, uint32_t soff) const {
#pragma unroll
    for (int j = 0; j < NP; ++j) dma16bs(rs, offB[j], soff, dst + (8 * j + w) * 1024);
  }
};

template <int T_>
__device__ __forceinline__ u32x4_t fragA(const unsigned char* s, int row, int kstep, int hh) {   // [32][T] image
  // slot (2 kstep + hh) ^ (row & 15) == (2 kstep) ^ (hh ^ (row & 15)): one xor per fragment on a per-lane constant
  return *(const u32x4_t*)(s + row * (2 * T_) + (((2 * kstep) ^ (hh ^ (row & 15))) << 4));
}
__device__ __forceinline__ u32x4_t fragB(const unsigned char* s, int row, int u, int hh) {       // [T][32] image
  const int e = row * 4 + 2 * u + hh;
  const int p = (e & ~15) | ((e & 15) ^ ((e >> 4) & 15));
  return *(const u32x4_t*)(s + p * 16);
}

// stage the [T][256] slice of an activation (rows t, columns d0..d0+255 of a [T][D] matrix) and return this wave's B
// fragments (column 32*wid + lane%32, k = t)
template <typename L, int T_>
__device__ __forceinline__ void load_bfrags(u32x4_t (&f)[T_ / 16], unsigned char* smem, const uint16_t* act, int D, int d0,
                                            const uint16_t* zero, int tid) {
  const int lane = tid & 63, wid = tid >> 6;
  TransDmaB<256, 8> sx;
  sx.init(act, D, d0, D, tid);
#pragma unroll
  for (int kt = 0; kt < T_ / 64; ++kt) sx.issue(smem + kt * 32768, kt * 64, T_, zero, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int kt = 0; kt < T_ / 64; ++kt)
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) f[kt * 4 + sub] = frag_trans<256>(smem + kt * 32768, 32 * wid + (lane & 31), sub, lane);
  __syncthreads();
}

__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b) {
  // a[lanes 32..63] <-> b[lanes 0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// SAVE (round 4): the forward also writes the hidden activation h = gelu(pre) and the derivative act'(pre) as [B][O][D] 16-bit
// tensors (hout / gout), staged through LDS into whole 512-byte rows exactly like tokmix_bwd_hidden_kernel's output.  The
// backward pass then needs no recomputation: dh = (W2^T dy) * act' is a plain batched GEMM with the aux-multiply epilogue and
// the weight gradients read the saved h (the recomputing kernel spends its time in erf / pdf arithmetic: 242 us per layer against
// ~90 for the GEMM).  LDS: 3-stage ring (96 KiB) + two 32-KiB staging buffers; the biases come through scalar loads.
template <typename L, int T_, bool SAVE = false>
__global__ __launch_bounds__(512, 2) void tokmix_fwd_kernel(const uint16_t* __restrict__ xn, const uint16_t* __restrict__ w1,
                                                            const float* __restrict__ b1, const uint16_t* __restrict__ w2,
                                                            const float* __restrict__ b2, const float* __restrict__ res,
                                                            float* __restrict__ y, int D, int O, const uint16_t* zero,
                                                            uint16_t* __restrict__ hout = nullptr,
                                                            uint16_t* __restrict__ gout = nullptr) {
  constexpr int NK1 = T_ / 16;                 // k-steps of GEMM-1
  constexpr int NTB = T_ / 32;                 // 32-row output blocks of GEMM-2
  constexpr int HALF = 64 * T_;                // bytes of one [32][T] / [T][32] chunk image
  constexpr int STAGE = 2 * HALF;
  constexpr int PPC = 2 * ChunkDma<T_>::NP;    // DMA pieces per thread per chunk
  constexpr int NST = SAVE ? 3 : TM_NST;       // ring depth
  constexpr int OUTB = 2 * TM_OC * TM_DT * 2;  // SAVE: h and act' tiles of one chunk: 2 x [32][256] 16-bit = 32 KiB
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // ring (>= xn staging) | b1 (O floats)  /  SAVE: ring | outb[2]
  constexpr int RING = SAVE ? NST * STAGE : ((NST * STAGE > (T_ / 64) * 32768) ? NST * STAGE : (T_ / 64) * 32768);
  float* b1s = (float*)(smem + RING);
  unsigned char* outb = smem + RING;           // SAVE only (the prologue's fragment staging may run into it: barrier below)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, d0 = blockIdx.x * TM_DT;
  const int64_t act0 = (int64_t)b * T_ * D;
  const int64_t out0 = (int64_t)b * O * D;

  if constexpr (!SAVE)
    for (int i = tid; i < O; i += 512) b1s[i] = b1[i];
  u32x4_t xf[NK1];
  load_bfrags<L, T_>(xf, smem, xn + act0, D, d0, zero, tid);

  ChunkDma<T_> dm;
  dm.init(tid, T_, O);
  const rsrc_t rs1 = make_rsrc(w1), rs2 = make_rsrc(w2);
  const int NC = O / TM_OC;
  auto issue = [&](int c) {
    unsigned char* st = smem + (c % NST) * STAGE;
    dm.issueA(st, rs1, (uint32_t)c * (TM_OC * T_ * 2));
    dm.issueB(st + HALF, rs2, (uint32_t)c * (TM_OC * 2));
  };
#pragma unroll
  for (int c = 0; c < NST - 1; ++c)
    if (c < NC) issue(c);

  f32x16_t acc2[NTB];
#pragma unroll
  for (int t = 0; t < NTB; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[t][i] = 0.0f;

  auto flush = [&](int c) {        // SAVE: coalesced store of chunk c's staged tiles: 64 rows of 512 B, 16 B per lane
    const unsigned char* ob = outb + (c & 1) * OUTB;
    const int d = d0 + 8 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * i + 2 * wid + (lane >> 5);            // 0..31: h rows, 32..63: act' rows
      const u32x4_t v = *(const u32x4_t*)(ob + row * 512 + (lane & 31) * 16);
      uint16_t* dst = (row < 32 ? hout : gout) + out0 + (int64_t)(c * TM_OC + (row & 31)) * D + d;
      if (d < D) *(u32x4_t*)dst = v;
    }
  };

  for (int c = 0; c < NC; ++c) {
    if constexpr (SAVE) {
      // vmcnt retires in issue order and counts the flush stores: behind chunk c's pieces were issued the flush of iteration
      // c-3's tiles (4 stores, in iteration c-2), chunk c+1 (PPC pieces), the flush in iteration c-1 (4 stores).  Lanes beyond D
      // skip their stores, so the counts below are upper bounds of what may stay in flight only when every lane stores: keep to
      // the conservative side (fewer allowed outstanding) when D is ragged — D % 256 == 0 in every shipped shape
      const bool more = c + 1 < NC;
      if (D % TM_DT) {
        wait_vm<0>();
      } else if (c >= 3) {
        if (more) wait_vm<PPC + 8>(); else wait_vm<8>();
      } else if (c == 2) {
        if (more) wait_vm<PPC + 4>(); else wait_vm<4>();
      } else {
        if (more) wait_vm<PPC>(); else wait_vm<0>();
      }
    } else {
      // chunk c has landed when at most the pieces of the (up to two) later chunks are still in flight
      if (c + 2 < NC)
        wait_vm<2 * PPC>();
      else if (c + 1 < NC)
        wait_vm<PPC>();
      else
        wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (c + NST - 1 < NC) issue(c + NST - 1);      // into the stage everyone finished reading before this barrier
    if constexpr (SAVE)
      if (c > 0) flush(c - 1);                     // staged by everyone before this barrier
    const unsigned char* s1 = smem + (c % NST) * STAGE;
    const unsigned char* s2 = s1 + HALF;
    f32x16_t a1;
#pragma unroll
    for (int i = 0; i < 16; ++i) a1[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < NK1; ++s) mma_lo<L>(a1, fragA<T_>(s1, l31, s, hh), xf[s]);
    // bias + exact-erf GELU in registers; register r of the block is hidden row 8 (r / 4) + 4 hh + r % 4
    uint32_t P[4][2];
    if constexpr (SAVE) {
      // the chunk's 32 biases through scalar loads (wave-uniform address; a vector load would drain the LDS-DMA ring)
      const const_f32p bc = (const_f32p)(b1 + __builtin_amdgcn_readfirstlane(c) * TM_OC);
      unsigned char* ob = outb + (c & 1) * OUTB;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = 4 * m + 2 * q;
          f32x2_t pre, cdf, e;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ru = r + u;
            const float blo = bc[8 * (ru >> 2) + (ru & 3)], bhi = bc[8 * (ru >> 2) + 4 + (ru & 3)];
            pre[u] = a1[ru] + (hh ? bhi : blo);
          }
          gelu_parts_fast2(pre, cdf, e);
          const f32x2_t hv = pre * cdf;
          const f32x2_t gv = cdf + pre * 0.39894228040143267794f * e;
          P[m][q] = lo_pack2<L>(hv[0], hv[1]);
          const uint32_t gp = lo_pack2<L>(gv[0], gv[1]);
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int ru = r + u;
            const int ol = 8 * (ru >> 2) + 4 * hh + (ru & 3);
            *(uint16_t*)(ob + ol * 512 + (32 * wid + l31) * 2) = (uint16_t)(u ? (P[m][q] >> 16) : (P[m][q] & 0xffffu));
            *(uint16_t*)(ob + (32 + ol) * 512 + (32 * wid + l31) * 2) = (uint16_t)(u ? (gp >> 16) : (gp & 0xffffu));
          }
        }
      }
    } else {
    const float* bc = b1s + c * TM_OC + 4 * hh;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4_t bb = *(const f32x4_t*)(bc + 8 * m);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const f32x2_t g = act_gelu_fast2(f32x2_t{a1[4 * m + 2 * q] + bb[2 * q], a1[4 * m + 2 * q + 1] + bb[2 * q + 1]});
        P[m][q] = lo_pack2<L>(g[0], g[1]);
      }
    }
    }
    // make each lane's 8 rows per k-step consecutive: half 0 keeps rows 0-3 and receives 4-7, half 1 gets 8-11 and keeps 12-15
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      swap_halves(P[0][q], P[1][q]);
      swap_halves(P[2][q], P[3][q]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const u32x4_t hb = {P[2 * u][0], P[2 * u][1], P[2 * u + 1][0], P[2 * u + 1][1]};
#pragma unroll
      for (int t = 0; t < NTB; ++t) mma_lo<L>(acc2[t], fragB(s2, 32 * t + l31, u, hh), hb);
    }
  }

  if constexpr (SAVE) {
    __syncthreads();
    flush(NC - 1);
  }
  // epilogue: [T/2 rows][256 cols] fp32 through LDS per half, then whole 1 KiB rows: y = acc + b2[t] + res
  constexpr int RH = T_ / 2;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    __syncthreads();
#pragma unroll
    for (int tb = 0; tb < NTB / 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tl = 32 * tb + 8 * (r >> 2) + 4 * hh + (r & 3);
        *(float*)(smem + tl * 1024 + (32 * wid + l31) * 4) = acc2[hf * (NTB / 2) + tb][r];
      }
    __syncthreads();
    const int d = d0 + 4 * lane;
    if (d < D) {
#pragma unroll 4
      for (int i = 0; i < RH / 8; ++i) {
        const int tl = wid + 8 * i, t = hf * RH + tl;
        f32x4_t v = *(const f32x4_t*)(smem + tl * 1024 + lane * 16);
        const int64_t off = act0 + (int64_t)t * D + d;
        const f32x4_t r = *(const f32x4_t*)(res + off);
        const float bt = b2[t];
        v += r + bt;
        *(f32x4_t*)(y + off) = v;
      }
    }
  }
}

// Backward, hidden part: recompute pre = W1 xn + b1 and g = W2^T dy chunk by chunk; write h = gelu(pre) and
// dh = g * gelu'(pre) as [B][O][D] (16-bit) for the weight-gradient GEMMs and the dx GEMM.
#ifdef FFVC_TM_TIMING
__device__ unsigned long long tm_stamps[8][6][8];      // [wave][chunk - 4][event], workgroup (1, 7)
#define TM_STAMP(ev)                                                                                                   \
  do {                                                                                                                 \
    if (blockIdx.x == 1 && blockIdx.y == 7 && lane == 0 && c >= 4 && c < 10) tm_stamps[wid][c - 4][ev] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define TM_STAMP(ev)
#endif
template <typename L, int T_>
__global__ __launch_bounds__(512, 2) void tokmix_bwd_hidden_kernel(const uint16_t* __restrict__ xn, const uint16_t* __restrict__ dy,
                                                                   const uint16_t* __restrict__ w1, const float* __restrict__ b1,
                                                                   const uint16_t* __restrict__ w2t, uint16_t* __restrict__ hout,
                                                                   uint16_t* __restrict__ dhout, float* __restrict__ db1,
                                                                   int D, int O, const uint16_t* zero) {
  constexpr int NK1 = T_ / 16;
  constexpr int HALF = 64 * T_;
  constexpr int STAGE = 2 * HALF;
  constexpr int NST = 3;                       // ring depth here: 3 stages + 2 output staging buffers fill the LDS
  constexpr int PPC = 2 * ChunkDma<T_>::NP;
  constexpr int OUTB = 2 * TM_OC * TM_DT * 2;  // h and dh tiles of one chunk: 2 x [32][256] 16-bit = 32 KiB
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int RING = NST * STAGE;            // the fragment staging of the prologue ((T/64) x 32 KiB) may run into outb[0]
  unsigned char* outb = smem + RING;           // [2][OUTB]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, d0 = blockIdx.x * TM_DT;
  const int64_t act0 = (int64_t)b * T_ * D;

  u32x4_t xf[NK1], yf[NK1];
  load_bfrags<L, T_>(xf, smem, xn + act0, D, d0, zero, tid);
  load_bfrags<L, T_>(yf, smem, dy + act0, D, d0, zero, tid);

  ChunkDma<T_> dm;
  dm.init(tid, T_, T_);
  const rsrc_t rs1 = make_rsrc(w1), rs2 = make_rsrc(w2t);
  const int NC = O / TM_OC;
  auto issue = [&](int c) {
    unsigned char* st = smem + (c % NST) * STAGE;
    dm.issueA(st, rs1, (uint32_t)c * (TM_OC * T_ * 2));
    dm.issueA(st + HALF, rs2, (uint32_t)c * (TM_OC * T_ * 2));
  };
#pragma unroll
  for (int c = 0; c < NST; ++c)
    if (c < NC) issue(c);

  const int64_t out0 = (int64_t)b * O * D;
  auto flush = [&](int c) {        // coalesced store of chunk c's staged tiles: 64 rows of 512 B, 16 B per lane
    const unsigned char* ob = outb + (c & 1) * OUTB;
    const int d = d0 + 8 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 16 * i + 2 * wid + (lane >> 5);            // 0..31: h rows, 32..63: dh rows
      const u32x4_t v = *(const u32x4_t*)(ob + row * 512 + (lane & 31) * 16);
      if (d < D) {
        uint16_t* dst = (row < 32 ? hout : dhout) + out0 + (int64_t)(c * TM_OC + (row & 31)) * D + d;
        *(u32x4_t*)dst = v;
      }
      if (db1 && i >= 2) {     // rows 32..63 are dh: its sum over (sample, d) is the first Conv1d's bias gradient
        float sum = 0.0f;
        if (d < D) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            sum += lo_unpack<L>((uint16_t)(v[e] & 0xffffu)) + lo_unpack<L>((uint16_t)(v[e] >> 16));
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) sum += __shfl_xor(sum, o, 64);
        // scalar base + 32-bit lane offset, spelled out: left to the compiler the loop-invariant per-lane 64-bit address is
        // hoisted, spilled, and reloaded from scratch behind a vmcnt(0) that drains the LDS-DMA ring every chunk
        if ((lane & 31) == 0) {
          const float* base = db1 + __builtin_amdgcn_readfirstlane(c) * TM_OC;
          asm volatile("global_atomic_add_f32 %0, %1, %2" ::"v"((uint32_t)(row & 31) * 4u), "v"(sum), "s"(base) : "memory");
        }                      // (2 extra vm ops per flush: the counted waits below stay conservative)
      }
    }
  };

  // Software pipeline over the chunks: iteration c runs the erf-GELU / staging VALU work of chunk c INTERLEAVED with the
  // MFMAs of chunk c+1 (into the other accumulator pair), so the matrix pipe works under the VALU phase instead of after
  // it (measured before: ~2200 cycles of MFMA then ~2500-3000 cycles of VALU per chunk, both waves of a SIMD in lockstep).
  // Ring: reading c+1, landing c+2, issuing c+3 into the slot of c (whose MFMAs every wave finished before the barrier).
  auto mfma_k = [&](int s, const unsigned char* s1, const unsigned char* s2, f32x16_t& n1, f32x16_t& n2) {
    mma_lo<L>(n1, fragA<T_>(s1, l31, s, hh), xf[s]);
    mma_lo<L>(n2, fragA<T_>(s2, l31, s, hh), yf[s]);
  };
  auto gelu_pair = [&](int r, const f32x16_t& a1, const f32x16_t& a2, unsigned char* ob, const_f32p bc) {
    f32x2_t pre, cdf, e;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ru = r + u;
      const float blo = bc[8 * (ru >> 2) + (ru & 3)], bhi = bc[8 * (ru >> 2) + 4 + (ru & 3)];
      pre[u] = a1[ru] + (hh ? bhi : blo);
    }
    gelu_parts_fast2(pre, cdf, e);
    const f32x2_t hv = pre * cdf;
    const f32x2_t dv = f32x2_t{a2[r], a2[r + 1]} * (cdf + pre * 0.39894228040143267794f * e);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ru = r + u;
      const int ol = 8 * (ru >> 2) + 4 * hh + (ru & 3);
      const uint32_t pk = lo_pack2<L>(hv[u], dv[u]);
      *(uint16_t*)(ob + ol * 512 + (32 * wid + l31) * 2) = (uint16_t)(pk & 0xffffu);
      *(uint16_t*)(ob + (32 + ol) * 512 + (32 * wid + l31) * 2) = (uint16_t)(pk >> 16);
    }
  };
  f32x16_t accA1, accA2, accB1, accB2;
  // chunk 0's MFMAs ahead of the loop
  if (NC >= 3) wait_vm<2 * PPC>(); else if (NC == 2) wait_vm<PPC>(); else wait_vm<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) accA1[i] = accA2[i] = 0.0f;
#pragma unroll
  for (int s = 0; s < NK1; ++s) mfma_k(s, smem, smem + HALF, accA1, accA2);

  auto body = [&](auto has_next, int c, f32x16_t& c1, f32x16_t& c2, f32x16_t& n1, f32x16_t& n2) {
    constexpr bool NEXT = decltype(has_next)::value;
    TM_STAMP(0);
    if constexpr (NEXT) {
      // chunk c+1 must have landed.  vmcnt retires in issue order and counts the flush stores too; issued after chunk
      // c+1's pieces: the flush of iteration c-2 (4 stores, c >= 3), chunk c+2 (PPC pieces), the flush of iteration c-1
      // (4 stores, c >= 2)
      const bool more = c + 2 < NC;
      if (c >= 3) {
        if (more) wait_vm<PPC + 8>(); else wait_vm<8>();
      } else if (c == 2) {
        if (more) wait_vm<PPC + 4>(); else wait_vm<4>();
      } else {
        if (more) wait_vm<PPC>(); else wait_vm<0>();
      }
    }
    TM_STAMP(1);
    __builtin_amdgcn_s_barrier();
    TM_STAMP(2);
    if (c + NST < NC) issue(c + NST);          // into chunk c's slot
    TM_STAMP(3);
    if (c > 0) flush(c - 1);                   // staged by everyone before this barrier
    TM_STAMP(4);
    unsigned char* ob = outb + (c & 1) * OUTB;
    // the chunk's 32 biases come through SCALAR loads (wave-uniform addresses): an ordinary vector load here would make
    // hipcc drain the LDS-DMA ring with vmcnt(0) at its first use
    const const_f32p bc = (const_f32p)(b1 + __builtin_amdgcn_readfirstlane(c) * TM_OC);
    if constexpr (NEXT) {
      const unsigned char* s1 = smem + ((c + 1) % NST) * STAGE;
      const unsigned char* s2 = s1 + HALF;
#pragma unroll
      for (int i = 0; i < 16; ++i) n1[i] = n2[i] = 0.0f;
      constexpr int KPG = NK1 / 8;             // k-steps of the next chunk per GELU pair
#pragma unroll
      for (int r = 0; r < 16; r += 2) {        // two hidden rows at a time: the erf / pdf polynomial runs on v_pk_*_f32
        // the next chunk's fragments are read one VALU block ahead of the MFMAs that consume them: the LDS latency hides
        // under the GELU arithmetic instead of sitting in front of every MFMA (one k-step = 8 VGPRs in flight)
        static_assert(KPG == 1 || KPG == 2, "k-steps per GELU pair");
        const int s0 = (r / 2) * KPG;
        u32x4_t f0 = fragA<T_>(s1, l31, s0, hh), f1 = fragA<T_>(s2, l31, s0, hh);
        __builtin_amdgcn_sched_barrier(0);
        f32x2_t pre, cdf, e;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ru = r + u;
          const float blo = bc[8 * (ru >> 2) + (ru & 3)], bhi = bc[8 * (ru >> 2) + 4 + (ru & 3)];
          pre[u] = c1[ru] + (hh ? bhi : blo);
        }
        gelu_parts_fast2(pre, cdf, e);
        __builtin_amdgcn_sched_barrier(0);
        mma_lo<L>(n1, f0, xf[s0]);
        mma_lo<L>(n2, f1, yf[s0]);
        if constexpr (KPG == 2) {
          f0 = fragA<T_>(s1, l31, s0 + 1, hh);
          f1 = fragA<T_>(s2, l31, s0 + 1, hh);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x2_t hv = pre * cdf;
        const f32x2_t dv = f32x2_t{c2[r], c2[r + 1]} * (cdf + pre * 0.39894228040143267794f * e);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int ru = r + u;
          const int ol = 8 * (ru >> 2) + 4 * hh + (ru & 3);
          const uint32_t pk = lo_pack2<L>(hv[u], dv[u]);
          *(uint16_t*)(ob + ol * 512 + (32 * wid + l31) * 2) = (uint16_t)(pk & 0xffffu);
          *(uint16_t*)(ob + (32 + ol) * 512 + (32 * wid + l31) * 2) = (uint16_t)(pk >> 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KPG == 2) {
          mma_lo<L>(n1, f0, xf[s0 + 1]);
          mma_lo<L>(n2, f1, yf[s0 + 1]);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; r += 2) gelu_pair(r, c1, c2, ob, bc);
    }
    TM_STAMP(6);
  };
  {
    std::true_type yes;
    std::false_type no;
    int c = 0;
    for (; c + 2 < NC; c += 2) {
      body(yes, c, accA1, accA2, accB1, accB2);
      body(yes, c + 1, accB1, accB2, accA1, accA2);
    }
    if (c + 1 < NC) {
      body(yes, c, accA1, accA2, accB1, accB2);
      body(no, c + 1, accB1, accB2, accA1, accA2);
    } else {
      body(no, c, accA1, accA2, accB1, accB2);
    }
  }
  __syncthreads();
  flush(NC - 1);
}
#ifdef FFVC_TM_TIMING
extern "C" int ffvc_debug_tm_stamps(unsigned long long* host_out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tm_stamps), sizeof(unsigned long long) * 8 * 6 * 8);
}
#endif

const uint16_t* tm_zero_page() {
  static uint16_t* page[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!page[dev]) {
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    page[dev] = (uint16_t*)p;
  }
  return page[dev];
}

template <typename L, int T_>
int launch_fwd_save(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const float* res, float* y,
                    void* h, void* g, int B, int D, int O, hipStream_t st, const uint16_t* zero) {
  constexpr int lds_main = 3 * 128 * T_ + 2 * (2 * TM_OC * TM_DT * 2);
  constexpr int lds_stage = (T_ / 64) * 32768;
  constexpr int lds_epi = (T_ / 2) * 1024;
  constexpr int lds = (lds_main > lds_stage ? lds_main : lds_stage) > lds_epi ? (lds_main > lds_stage ? lds_main : lds_stage) : lds_epi;
  static_assert(lds <= 163840, "LDS budget");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)tokmix_fwd_kernel<L, T_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr = true;
  }
  hipLaunchKernelGGL((tokmix_fwd_kernel<L, T_, true>), dim3(ceil_div(D, TM_DT), B), dim3(512), lds, st, (const uint16_t*)xn,
                     (const uint16_t*)w1, b1, (const uint16_t*)w2, b2, res, y, D, O, zero, (uint16_t*)h, (uint16_t*)g);
  FFVC_LAUNCH_CHECK();
  return 0;
}

template <typename L, int T_>
int launch_fwd(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2, const float* res, float* y,
               int B, int D, int O, hipStream_t st, const uint16_t* zero) {
  constexpr int STAGE = 128 * T_;
  constexpr int RING = (TM_NST * STAGE > (T_ / 64) * 32768) ? TM_NST * STAGE : (T_ / 64) * 32768;
  const int lds = RING + O * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)tokmix_fwd_kernel<L, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr = true;
  }
  hipLaunchKernelGGL((tokmix_fwd_kernel<L, T_>), dim3(ceil_div(D, TM_DT), B), dim3(512), lds, st, (const uint16_t*)xn,
                     (const uint16_t*)w1, b1, (const uint16_t*)w2, b2, res, y, D, O, zero);
  FFVC_LAUNCH_CHECK();
  return 0;
}

template <typename L, int T_>
int launch_bwd(const void* xn, const void* dy, const void* w1, const float* b1, const void* w2t, void* h, void* dh, float* db1,
               int B, int D, int O, hipStream_t st, const uint16_t* zero) {
  constexpr int lds_main = 3 * 128 * T_ + 2 * (2 * TM_OC * TM_DT * 2);
  constexpr int lds_stage = (T_ / 64) * 32768;
  constexpr int lds = lds_main > lds_stage ? lds_main : lds_stage;
  static_assert(lds <= 163840, "LDS budget");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)tokmix_bwd_hidden_kernel<L, T_>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr = true;
  }
  hipLaunchKernelGGL((tokmix_bwd_hidden_kernel<L, T_>), dim3(ceil_div(D, TM_DT), B), dim3(512), lds, st, (const uint16_t*)xn,
                     (const uint16_t*)dy, (const uint16_t*)w1, b1, (const uint16_t*)w2t, (uint16_t*)h, (uint16_t*)dh, db1, D, O, zero);
  FFVC_LAUNCH_CHECK();
  return 0;
}

int tm_check(const char* who, int dtype, int B, int T, int D, int O) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16, "%s: 16-bit storage only (dtype %d)", who, dtype);
  FFVC_CHECK_ARG(T == 128 || T == 256, "%s: T=%d unsupported (128 | 256; use the unfused GEMM path)", who, T);
  FFVC_CHECK_ARG(B > 0 && B <= 65535 && D > 0 && D % 32 == 0 && O > 0 && O % TM_OC == 0 && O * 4 <= 16384,
                 "%s: bad dims B=%d D=%d O=%d (D %% 32, O %% 32, O <= 4096)", who, B, D, O);
  return 0;
}

}  // namespace

extern "C" int ffvc_tokmix_supported(int dtype, int T, int D, int O) {
  return (dtype == FFVC_BF16 || dtype == FFVC_F16) && (T == 128 || T == 256) && D > 0 && D % 32 == 0 && O > 0 &&
         O % TM_OC == 0 && O <= 4096;
}

extern "C" int ffvc_tokmix_fwd(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2,
                               const float* residual, float* y, int dtype, int B, int T, int D, int O, void* stream) {
  FFVC_CHECK_ARG(xn && w1 && b1 && w2 && b2 && residual && y, "ffvc_tokmix_fwd: null pointer");
  if (int e = tm_check("ffvc_tokmix_fwd", dtype, B, T, D, O)) return e;
  FFVC_CHECK_ARG(((uintptr_t)xn % 16) == 0 && ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)w2 % 16) == 0 &&
                     ((uintptr_t)residual % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)b1 % 16) == 0,
                 "ffvc_tokmix_fwd: pointers must be 16-byte aligned");
  const uint16_t* zero = tm_zero_page();
  FFVC_CHECK_ARG(zero != nullptr, "ffvc_tokmix_fwd: zero page allocation failed");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FFVC_F16)
    return T == 256 ? launch_fwd<f16_t, 256>(xn, w1, b1, w2, b2, residual, y, B, D, O, st, zero)
                    : launch_fwd<f16_t, 128>(xn, w1, b1, w2, b2, residual, y, B, D, O, st, zero);
  return T == 256 ? launch_fwd<uint16_t, 256>(xn, w1, b1, w2, b2, residual, y, B, D, O, st, zero)
                  : launch_fwd<uint16_t, 128>(xn, w1, b1, w2, b2, residual, y, B, D, O, st, zero);
}

extern "C" int ffvc_tokmix_fwd_save(const void* xn, const void* w1, const float* b1, const void* w2, const float* b2,
                                    const float* residual, float* y, void* h, void* gact, int dtype, int B, int T, int D, int O,
                                    void* stream) {
  FFVC_CHECK_ARG(xn && w1 && b1 && w2 && b2 && residual && y && h && gact, "ffvc_tokmix_fwd_save: null pointer");
  if (int e = tm_check("ffvc_tokmix_fwd_save", dtype, B, T, D, O)) return e;
  FFVC_CHECK_ARG(((uintptr_t)xn % 16) == 0 && ((uintptr_t)w1 % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
                     ((uintptr_t)y % 16) == 0 && ((uintptr_t)b1 % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)gact % 16) == 0,
                 "ffvc_tokmix_fwd_save: pointers must be 16-byte aligned");
  const uint16_t* zero = tm_zero_page();
  FFVC_CHECK_ARG(zero != nullptr, "ffvc_tokmix_fwd_save: zero page allocation failed");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FFVC_F16)
    return T == 256 ? launch_fwd_save<f16_t, 256>(xn, w1, b1, w2, b2, residual, y, h, gact, B, D, O, st, zero)
                    : launch_fwd_save<f16_t, 128>(xn, w1, b1, w2, b2, residual, y, h, gact, B, D, O, st, zero);
  return T == 256 ? launch_fwd_save<uint16_t, 256>(xn, w1, b1, w2, b2, residual, y, h, gact, B, D, O, st, zero)
                  : launch_fwd_save<uint16_t, 128>(xn, w1, b1, w2, b2, residual, y, h, gact, B, D, O, st, zero);
}

extern "C" int ffvc_tokmix_bwd_hidden(const void* xn, const void* dy, const void* w1, const float* b1, const void* w2t,
                                      void* h, void* dh, float* db1, int dtype, int B, int T, int D, int O, void* stream) {
  FFVC_CHECK_ARG(xn && dy && w1 && b1 && w2t && h && dh, "ffvc_tokmix_bwd_hidden: null pointer");
  if (int e = tm_check("ffvc_tokmix_bwd_hidden", dtype, B, T, D, O)) return e;
  FFVC_CHECK_ARG(((uintptr_t)xn % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)w1 % 16) == 0 &&
                     ((uintptr_t)w2t % 16) == 0 && ((uintptr_t)h % 16) == 0 && ((uintptr_t)dh % 16) == 0,
                 "ffvc_tokmix_bwd_hidden: pointers must be 16-byte aligned");
  const uint16_t* zero = tm_zero_page();
  FFVC_CHECK_ARG(zero != nullptr, "ffvc_tokmix_bwd_hidden: zero page allocation failed");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FFVC_F16)
    return T == 256 ? launch_bwd<f16_t, 256>(xn, dy, w1, b1, w2t, h, dh, db1, B, D, O, st, zero)
                    : launch_bwd<f16_t, 128>(xn, dy, w1, b1, w2t, h, dh, db1, B, D, O, st, zero);
  return T == 256 ? launch_bwd<uint16_t, 256>(xn, dy, w1, b1, w2t, h, dh, db1, B, D, O, st, zero)
                  : launch_bwd<uint16_t, 128>(xn, dy, w1, b1, w2t, h, dh, db1, B, D, O, st, zero);
}
