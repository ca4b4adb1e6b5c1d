// attention.hip — fused softmax attention for SHORT sequences (T <= 64 tokens, head dim 64, bf16 storage):
// the CLIP ViT-B/32 image tower has 50 tokens per cutout (cloob.py:199-200, nn.MultiheadAttention inside
// ResidualAttentionBlock), i.e. 6144 independent (cutout, head) problems of 50x50x64 per layer.  As batched GEMMs +
// softmax passes those cost ~10 ms per step (128x128 tiles are 85 % padding, scores and probabilities round-trip
// through HBM); here ONE wave owns one (cutout, head): scores, softmax and both products stay in registers.
//
// Layout trick: scores are computed TRANSPOSED (S^T = K Q^T), so that a lane owns one query column and the keys run
// over the accumulator registers — row max / row sum are in-lane reductions plus one exchange between the two half
// waves, and 8 consecutive accumulator registers are exactly the B fragment (k = key) of the next MFMA.  MFMA sums
// over k in any order as long as both operands agree, so the A operand (V^T, K^T, dO^T or Q^T, staged transposed in
// LDS) is simply read in the accumulator's key order (two 8-byte reads per fragment) instead of shuffling P.
//
// Numerics match the GEMM + softmax path it replaces: fp32 scores and statistics, probabilities rounded to bf16
// before the second product and before the softmax gradient, fp32 accumulation everywhere.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int TS = 68;            // LDS row stride (bf16) of a transposed 64 x 64 panel: 136 B, 8-byte aligned rows
constexpr int PANEL = 64 * TS;
constexpr float LOG2E = 1.44269504088896341f;

// L = 16-bit storage format tag (uint16_t = bf16, f16_t = IEEE half); data moves as raw 16-bit words
#define mma mma_lo<L>

// 8 consecutive features (k = 16 s + 8 h ...) of token `row` straight from global memory; rows >= T read as zero.
__device__ __forceinline__ u32x4_t rowfrag(const uint16_t* base, int64_t ld, int row, int T, int s, int h) {
  const u32x4_t z = {0u, 0u, 0u, 0u};
  return row < T ? *(const u32x4_t*)(base + (int64_t)row * ld + 16 * s + 8 * h) : z;
}

// acc[a][b][r] += sum_d A[rowA0 + 32 a + i(r)][d] * B[rowB0 + 32 b + lane%32][d],  i(r) = 8 (r/4) + 4 (lane/32) + r%4
template <typename L, int NA, int NB>
__device__ __forceinline__ void rows_product(f32x16_t (&acc)[NA][NB], const uint16_t* A, int rowA0, const uint16_t* B,
                                             int rowB0, int64_t ld, int T, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    u32x4_t fa[NA], fb[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a) fa[a] = rowfrag(A, ld, rowA0 + 32 * a + l31, T, s, h);
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = rowfrag(B, ld, rowB0 + 32 * b + l31, T, s, h);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) mma(acc[a][b], fa[a], fb[b]);
  }
}

// Xt[d][tok] (LDS, stride TS) <- X[tok][d] of one head (64 features), rows >= T zero.
__device__ __forceinline__ void stage_transposed(uint16_t* Xt, const uint16_t* X, int64_t ld, int T, int lane) {
  const int c = lane & 7;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 8 + (lane >> 3);
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (row < T) v = *(const u32x4_t*)(X + (int64_t)row * ld + 8 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Xt[(8 * c + 2 * e) * TS + row] = (uint16_t)(v[e] & 0xffffu);
      Xt[(8 * c + 2 * e + 1) * TS + row] = (uint16_t)(v[e] >> 16);
    }
  }
}

template <typename L>
__device__ __forceinline__ u32x4_t pack8(const f32x16_t& t, int first) {
  u32x4_t r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = lo_pack2<L>(t[first + 2 * e], t[first + 2 * e + 1]);
  return r;
}

// acc[dt][bt][r] += sum_c Xt[32 dt + i(r)][c] * src[c / 32][bt]{c % 32, lane},  c over 64 (contraction index of the
// source tiles' REGISTER dimension); NC = number of 32-wide contraction tiles present in src (1 or 2).
template <typename L, int NC, int NB>
__device__ __forceinline__ void lds_product(f32x16_t (&acc)[2][NB], const uint16_t* Xt, const f32x16_t (&src)[NC][NB],
                                            int c0, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int sp = 0; sp < 2 * NC; ++sp) {
    const int ct = sp >> 1, sg = sp & 1;
    u32x4_t fa[2], fb[NB];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      const uint16_t* p = Xt + (32 * dt + l31) * TS + c0 + 32 * ct + 16 * sg + 4 * h;
      const u32x2_t lo = *(const u32x2_t*)p;
      const u32x2_t hi = *(const u32x2_t*)(p + 8);
      fa[dt] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) fb[b] = pack8<L>(src[ct][b], 8 * sg);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int b = 0; b < NB; ++b) mma(acc[dt][b], fa[dt], fb[b]);
  }
}

template <int NA, int NB>
__device__ __forceinline__ void zero_tiles(f32x16_t (&t)[NA][NB]) {
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) t[a][b][r] = 0.0f;
}

__device__ __forceinline__ int reg_index(int r, int h) { return 8 * (r >> 2) + 4 * h + (r & 3); }   // row inside a 32-tile

// Softmax over the keys of transposed score tiles ST[kt][qt] (lane = query, registers = keys), in place, probabilities
// rounded to bf16 precision (kept as float).  Returns per (lane, qt) the row maximum and 1 / row sum.
template <typename L>
__device__ __forceinline__ void softmax_transposed(f32x16_t (&ST)[2][2], int T, float scale, int lane, float (&mx)[2],
                                                   float (&linv)[2]) {
  const int h = lane >> 5;
  const float c = scale * LOG2E;
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float m = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * kt + reg_index(r, h) < T) m = fmaxf(m, ST[kt][qt][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (32 * kt + reg_index(r, h) < T) ? __builtin_amdgcn_exp2f((ST[kt][qt][r] - m) * c) : 0.0f;
        ST[kt][qt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ST[kt][qt][r] = lo_round<L>(ST[kt][qt][r] * inv);
    mx[qt] = m;
    linv[qt] = inv;
  }
}

// out[(row0 + 32 bt + lane%32) * ld + 32 dt + i(r)] = acc[dt][bt][r]  (4 consecutive features per store), rows < T
template <typename L, int NB>
__device__ __forceinline__ void store_transposed(uint16_t* out, int64_t ld, const f32x16_t (&acc)[2][NB], int row0, int T,
                                                 int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int row = row0 + 32 * b + l31;
    if (row >= T) continue;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4_t v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[dt][b][4 * g + j];
        store4((L*)out + (int64_t)row * ld + 32 * dt + 8 * g + 4 * h, v);
      }
  }
}

// The same tile as fp8 bytes: out8[(row0 + 32 bt + lane%32) * ld + 32 dt + i(r)] = fp8(round_L(acc) * scale), 4 bytes per store; m collects
// max |round_L(acc)| (producer-side quantisation of the attention output for the fp8 out_proj, see ffvc_attn_flash_fwd_f8)
template <typename L, int NB>
__device__ __forceinline__ void store_transposed_f8(uint8_t* out8, int64_t ld, const f32x16_t (&acc)[2][NB], int row0, int T, int lane,
                                                    float scale, int fmt, float& m) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int row = row0 + 32 * b + l31;
    if (row >= T) continue;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = lo_round<L>(acc[dt][b][4 * g + j]);
          m = fmaxf(m, fabsf(v[j]));
        }
        *(uint32_t*)(out8 + (int64_t)row * ld + 32 * dt + 8 * g + 4 * h) = f8_pack4(fmt, v[0] * scale, v[1] * scale, v[2] * scale, v[3] * scale);
      }
  }
}

template <typename L>
__global__ __launch_bounds__(64) void attn_small_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o,
                                                           int T, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t Vt[PANEL];
  const int lane = threadIdx.x;
  const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  stage_transposed(Vt, V, ld, T, lane);
  f32x16_t ST[2][2];
  zero_tiles(ST);
  rows_product<L, 2, 2>(ST, Kp, 0, Q, 0, ld, T, lane);
  float mx[2], linv[2];
  softmax_transposed<L>(ST, T, scale, lane, mx, linv);
  __syncthreads();
  f32x16_t OT[2][2];
  zero_tiles(OT);
  lds_product<L, 2, 2>(OT, Vt, ST, 0, lane);
  store_transposed<L, 2>(o + (int64_t)b * T * D + hd * 64, D, OT, 0, T, lane);
}

// Backward.  Phase A works on transposed tiles (lane = query): dQ.  Phase B walks the two query tiles in the other
// orientation (lane = key, registers = queries), re-deriving P from the row statistics kept in LDS: dV and dK.
template <typename L>
__global__ __launch_bounds__(64, 2) void attn_small_bwd_kernel(const uint16_t* __restrict__ qkv,
                                                           const uint16_t* __restrict__ dout,
                                                           uint16_t* __restrict__ dqkv, int T, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t bufA[PANEL];   // K^T, later dO^T
  __shared__ __attribute__((aligned(16))) uint16_t bufB[PANEL];   // Q^T
  __shared__ __attribute__((aligned(16))) float stat[3][64];      // row max, 1 / row sum, delta per query
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  const int b = blockIdx.x / heads, hd = blockIdx.x - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const uint16_t* dO = dout + (int64_t)b * T * D + hd * 64;
  uint16_t* dQ = dqkv + (int64_t)b * T * ld + hd * 64;
  uint16_t* dK = dQ + D;
  uint16_t* dV = dQ + 2 * D;
  const float c = scale * LOG2E;

  stage_transposed(bufA, Kp, ld, T, lane);
  stage_transposed(bufB, Q, ld, T, lane);
  {
    f32x16_t PT[2][2], dPT[2][2];
    zero_tiles(PT);
    rows_product<L, 2, 2>(PT, Kp, 0, Q, 0, ld, T, lane);
    float mx[2], linv[2];
    softmax_transposed<L>(PT, T, scale, lane, mx, linv);
    // dP^T[key][query] = sum_d V[key][d] dO[query][d]   (dO rows have their own stride D)
    zero_tiles(dPT);
    {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u32x4_t fa[2], fb[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) fa[a] = rowfrag(V, ld, 32 * a + l31, T, s, h);
#pragma unroll
        for (int q = 0; q < 2; ++q) fb[q] = rowfrag(dO, D, 32 * q + l31, T, s, h);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int q = 0; q < 2; ++q) mma(dPT[a][q], fa[a], fb[q]);
      }
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      float delta = 0.0f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) delta += PT[kt][qt][r] * dPT[kt][qt][r];
      delta += __shfl_xor(delta, 32, 64);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dPT[kt][qt][r] = PT[kt][qt][r] * (dPT[kt][qt][r] - delta) * scale;   // dS^T
      if (h == 0) {
        stat[0][32 * qt + l31] = mx[qt];
        stat[1][32 * qt + l31] = linv[qt];
        stat[2][32 * qt + l31] = delta;
      }
    }
    __syncthreads();
    // dQ^T[d][query] = sum_key K^T[d][key] dS^T[key][query]
    f32x16_t dQT[2][2];
    zero_tiles(dQT);
    lds_product<L, 2, 2>(dQT, bufA, dPT, 0, lane);
    store_transposed<L, 2>(dQ, ld, dQT, 0, T, lane);
  }
  __syncthreads();
  stage_transposed(bufA, dO, D, T, lane);      // dO^T[d][query]
  __syncthreads();

  f32x16_t dVT[2][2], dKT[2][2];               // [d tile][key tile], lane = key
  zero_tiles(dVT);
  zero_tiles(dKT);
#pragma unroll 1
  for (int qt = 0; qt < 2; ++qt) {
    if (32 * qt >= T) break;
    f32x16_t S[1][2], dP[1][2];                // [.][key tile], registers = queries of tile qt
    zero_tiles(S);
    zero_tiles(dP);
    rows_product<L, 1, 2>(S, Q, 32 * qt, Kp, 0, ld, T, lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32x4_t fa = rowfrag(dO, D, 32 * qt + l31, T, s, h);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) mma(dP[0][kt], fa, rowfrag(V, ld, 32 * kt + l31, T, s, h));
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q0 = 32 * qt + 8 * g + 4 * h;
      const f32x4_t m4 = *(const f32x4_t*)&stat[0][q0];
      const f32x4_t i4 = *(const f32x4_t*)&stat[1][q0];
      const f32x4_t d4 = *(const f32x4_t*)&stat[2][q0];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const bool kok = 32 * kt + l31 < T;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 4 * g + j;
          float p = kok ? __builtin_amdgcn_exp2f((S[0][kt][r] - m4[j]) * c) * i4[j] : 0.0f;
          p = lo_round<L>(p);
          S[0][kt][r] = p;                                           // P[query][key]
          dP[0][kt][r] = p * (dP[0][kt][r] - d4[j]) * scale;         // dS[query][key]
        }
      }
    }
    // dV^T[d][key] += sum_query dO^T[d][query] P[query][key];  dK^T[d][key] += sum_query Q^T[d][query] dS[query][key]
    lds_product<L, 1, 2>(dVT, bufA, S, 32 * qt, lane);
    lds_product<L, 1, 2>(dKT, bufB, dP, 32 * qt, lane);
  }
  store_transposed<L, 2>(dV, ld, dVT, 0, T, lane);
  store_transposed<L, 2>(dK, ld, dKT, 0, T, lane);
}


// ---------------------------------------------------------------------------------------------------------------
// Flash-style attention for ANY sequence length (head dim 64): the x-transformer mapper's causal self-attention
// (transformer.py:11-20, 1024 tokens at cfg4) and ViT-L/14's 257-token attention (cfg5).  Same transposed-tile scheme
// as the short-sequence kernels above, with a loop over 64-key blocks and online softmax: no score matrix in HBM,
// causal blocks above the diagonal are never visited.  One wave owns 64 queries (forward, dQ) or 64 keys (dK, dV).
// The forward keeps lse[query] = log2(sum_k exp(scale * s_k)) (base-2 units) so the backward re-derives the
// probabilities as exp2(scale*log2e*s - lse) without a second statistics pass.
// ---------------------------------------------------------------------------------------------------------------
constexpr float NEG_BIG = -3.0e38f;

template <typename L, bool CAUSAL>
__global__ __launch_bounds__(64) void attn_flash_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o,
                                                            float* __restrict__ lse, int T, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t Vt[PANEL];
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const int q0 = 64 * blockIdx.x;
  const float c = scale * LOG2E;

  u32x4_t fq[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int s = 0; s < 4; ++s) fq[qt][s] = rowfrag(Q, ld, q0 + 32 * qt + l31, T, s, h);
  f32x16_t OT[2][2];
  zero_tiles(OT);
  float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.0f, 0.0f};
  const int nkb = CAUSAL ? (int)blockIdx.x + 1 : (T + 63) / 64;
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = 64 * kb;
    __syncthreads();
    stage_transposed(Vt, V + (int64_t)k0 * ld, ld, T - k0, lane);
    f32x16_t ST[2][2];
    zero_tiles(ST);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4_t fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = rowfrag(Kp, ld, k0 + 32 * a + l31, T, s, h);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) mma(ST[a][qt], fa[a], fq[qt][s]);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = q0 + 32 * qt + l31;
      const int klim = CAUSAL ? min(T, q + 1) : T;          // keys < klim are visible to this lane's query
      float mloc = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + 32 * kt + reg_index(r, h) < klim) mloc = fmaxf(mloc, ST[kt][qt][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(m[qt], mloc);
      const float alpha = __builtin_amdgcn_exp2f((m[qt] - mnew) * c);
      float sum = 0.0f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p =
              (k0 + 32 * kt + reg_index(r, h) < klim) ? __builtin_amdgcn_exp2f((ST[kt][qt][r] - mnew) * c) : 0.0f;
          ST[kt][qt][r] = p;
          sum += p;
        }
      sum += __shfl_xor(sum, 32, 64);
      l[qt] = l[qt] * alpha + sum;
      m[qt] = mnew;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) OT[dt][qt][r] *= alpha;
    }
    __syncthreads();
    lds_product<L, 2, 2>(OT, Vt, ST, 0, lane);
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const float inv = l[qt] > 0.0f ? 1.0f / l[qt] : 0.0f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) OT[dt][qt][r] *= inv;
    const int q = q0 + 32 * qt + l31;
    if (h == 0 && q < T) lse[(int64_t)bh * T + q] = m[qt] * c + __builtin_amdgcn_logf(l[qt]);   // v_log_f32 = log2
  }
  store_transposed<L, 2>(o + (int64_t)b * T * D + hd * 64, D, OT, q0, T, lane);
}

// dQ for one 64-query block (lane = query); also writes delta[query] = sum_d dO O for the dK / dV kernel.
template <typename L, bool CAUSAL>
__global__ __launch_bounds__(64) void attn_flash_bwd_dq_kernel(const uint16_t* __restrict__ qkv,
                                                               const uint16_t* __restrict__ out,
                                                               const uint16_t* __restrict__ dout,
                                                               const float* __restrict__ lse, float* __restrict__ delta,
                                                               uint16_t* __restrict__ dqkv, int T, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t Kt[PANEL];
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const uint16_t* O = out + (int64_t)b * T * D + hd * 64;
  const uint16_t* dO = dout + (int64_t)b * T * D + hd * 64;
  const int q0 = 64 * blockIdx.x;
  const float c = scale * LOG2E;

  u32x4_t fq[2][4], fdo[2][4];
  float dl[2], ls[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + 32 * qt + l31;
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fq[qt][s] = rowfrag(Q, ld, q, T, s, h);
      fdo[qt][s] = rowfrag(dO, D, q, T, s, h);
      const u32x4_t fo = rowfrag(O, D, q, T, s, h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += lo_unpack<L>((uint16_t)(fo[e] & 0xffffu)) * lo_unpack<L>((uint16_t)(fdo[qt][s][e] & 0xffffu));
        acc += lo_unpack<L>((uint16_t)(fo[e] >> 16)) * lo_unpack<L>((uint16_t)(fdo[qt][s][e] >> 16));
      }
    }
    acc += __shfl_xor(acc, 32, 64);
    dl[qt] = acc;
    ls[qt] = q < T ? lse[(int64_t)bh * T + q] : 0.0f;
    if (h == 0 && q < T) delta[(int64_t)bh * T + q] = acc;
  }
  f32x16_t dQT[2][2];
  zero_tiles(dQT);
  const int nkb = CAUSAL ? (int)blockIdx.x + 1 : (T + 63) / 64;
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = 64 * kb;
    __syncthreads();
    stage_transposed(Kt, Kp + (int64_t)k0 * ld, ld, T - k0, lane);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t PT[1][2], dPT[1][2];
      zero_tiles(PT);
      zero_tiles(dPT);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4_t fk = rowfrag(Kp, ld, k0 + 32 * kt + l31, T, s, h);
        const u32x4_t fv = rowfrag(V, ld, k0 + 32 * kt + l31, T, s, h);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          mma(PT[0][qt], fk, fq[qt][s]);
          mma(dPT[0][qt], fv, fdo[qt][s]);
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 32 * qt + l31;
        const int klim = q < T ? (CAUSAL ? min(T, q + 1) : T) : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p =
              (k0 + 32 * kt + reg_index(r, h) < klim) ? __builtin_amdgcn_exp2f(PT[0][qt][r] * c - ls[qt]) : 0.0f;
          dPT[0][qt][r] = p * (dPT[0][qt][r] - dl[qt]) * scale;          // dS^T[key][query]
        }
      }
      // dQ^T[d][query] += sum_key K^T[d][key] dS^T[key][query]
      lds_product<L, 1, 2>(dQT, Kt, dPT, 32 * kt, lane);
    }
  }
  store_transposed<L, 2>(dqkv + (int64_t)b * T * ld + hd * 64, ld, dQT, q0, T, lane);
}

// dK, dV for one 64-key block (lane = key), walking the query blocks that see it.
template <typename L, bool CAUSAL>
__global__ __launch_bounds__(64) void attn_flash_bwd_dkv_kernel(const uint16_t* __restrict__ qkv,
                                                                const uint16_t* __restrict__ dout,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta,
                                                                uint16_t* __restrict__ dqkv, int T, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t dOt[PANEL];    // dO^T[d][query]
  __shared__ __attribute__((aligned(16))) uint16_t Qt[PANEL];     // Q^T[d][query]
  __shared__ __attribute__((aligned(16))) float stat[2][64];      // lse, delta of the current query block
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const uint16_t* dO = dout + (int64_t)b * T * D + hd * 64;
  const int k0 = 64 * blockIdx.x;
  const float c = scale * LOG2E;

  u32x4_t fk[2][4], fv[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fk[kt][s] = rowfrag(Kp, ld, k0 + 32 * kt + l31, T, s, h);
      fv[kt][s] = rowfrag(V, ld, k0 + 32 * kt + l31, T, s, h);
    }
  f32x16_t dVT[2][2], dKT[2][2];               // [d tile][key tile]
  zero_tiles(dVT);
  zero_tiles(dKT);
  const int nqb = (T + 63) / 64;
#pragma unroll 1
  for (int qb = CAUSAL ? (int)blockIdx.x : 0; qb < nqb; ++qb) {
    const int q0 = 64 * qb;
    __syncthreads();
    stage_transposed(dOt, dO + (int64_t)q0 * D, D, T - q0, lane);
    stage_transposed(Qt, Q + (int64_t)q0 * ld, ld, T - q0, lane);
    {
      const int q = q0 + lane;
      stat[0][lane] = q < T ? lse[(int64_t)bh * T + q] : 0.0f;
      stat[1][lane] = q < T ? delta[(int64_t)bh * T + q] : 0.0f;
    }
    __syncthreads();
#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
      if (q0 + 32 * qt >= T) break;
      f32x16_t S[1][2], dP[1][2];              // [.][key tile], registers = queries of tile qt
      zero_tiles(S);
      zero_tiles(dP);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4_t fa = rowfrag(Q, ld, q0 + 32 * qt + l31, T, s, h);
        const u32x4_t fb = rowfrag(dO, D, q0 + 32 * qt + l31, T, s, h);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          mma(S[0][kt], fa, fk[kt][s]);
          mma(dP[0][kt], fb, fv[kt][s]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qi = 32 * qt + 8 * g + 4 * h;
        const f32x4_t l4 = *(const f32x4_t*)&stat[0][qi];
        const f32x4_t d4 = *(const f32x4_t*)&stat[1][qi];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int key = k0 + 32 * kt + l31;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const int q = q0 + qi + j;
            const bool ok = key < T && q < T && (!CAUSAL || key <= q);
            const float p = ok ? __builtin_amdgcn_exp2f(S[0][kt][r] * c - l4[j]) : 0.0f;
            S[0][kt][r] = p;                                           // P[query][key]
            dP[0][kt][r] = p * (dP[0][kt][r] - d4[j]) * scale;         // dS[query][key]
          }
        }
      }
      lds_product<L, 1, 2>(dVT, dOt, S, 32 * qt, lane);
      lds_product<L, 1, 2>(dKT, Qt, dP, 32 * qt, lane);
    }
  }
  uint16_t* dKp = dqkv + (int64_t)b * T * ld + hd * 64 + D;
  store_transposed<L, 2>(dKp + D, ld, dVT, k0, T, lane);
  store_transposed<L, 2>(dKp, ld, dKT, k0, T, lane);
}


// ---------------------------------------------------------------------------------------------------------------
// Flash attention, second generation: NWV waves per workgroup (64 queries each) share every 64-key block through LDS —
// K as rows (A operand of S^T = K Q^T, 16-byte fragment reads), V transposed (A operand of O^T += V^T P^T) — loaded
// cooperatively (each 16-byte chunk once per workgroup instead of once per wave) and prefetched into registers one block
// ahead of the MFMAs that consume it.  The per-wave math is that of attn_flash_fwd_kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int KS = 72;                 // row stride (halves) of the K-rows panel: 144 B keeps 16-byte fragment reads aligned
constexpr int KPANEL = 64 * KS;

// 16-byte fragment of row `row` of a rows panel: features 16 s + 8 h ..
__device__ __forceinline__ u32x4_t ldsfrag(const uint16_t* panel, int row, int s, int h) {
  return *(const u32x4_t*)(panel + row * KS + 16 * s + 8 * h);
}

template <int NWV>
struct BlockLoader {                   // 64 rows x 64 features of a [T, ld] matrix = 512 chunks of 16 B over 64*NWV threads
  static constexpr int NCH = 512 / (64 * NWV);
  u32x4_t r[NCH];
  __device__ __forceinline__ void load(const uint16_t* base, int64_t ld, int row0, int T, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 64 * NWV;
      const int row = row0 + (c >> 3);
      const u32x4_t z = {0u, 0u, 0u, 0u};
      r[i] = row < T ? *(const u32x4_t*)(base + (int64_t)row * ld + 8 * (c & 7)) : z;
    }
  }
  __device__ __forceinline__ void store_rows(uint16_t* panel, int tid) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 64 * NWV;
      *(u32x4_t*)(panel + (c >> 3) * KS + 8 * (c & 7)) = r[i];
    }
  }
  __device__ __forceinline__ void store_transposed(uint16_t* Xt, int tid) const {      // Xt[d][row], stride TS
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * 64 * NWV;
      const int row = c >> 3, d0 = 8 * (c & 7);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Xt[(d0 + 2 * e) * TS + row] = (uint16_t)(r[i][e] & 0xffffu);
        Xt[(d0 + 2 * e + 1) * TS + row] = (uint16_t)(r[i][e] >> 16);
      }
    }
  }
};

template <typename L, bool CAUSAL, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void attn_flash2_fwd_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ o,
                                                                   float* __restrict__ lse, int T, int heads, float scale,
                                                                   uint8_t* __restrict__ o8 = nullptr, float* __restrict__ f8_state = nullptr,
                                                                   int f8_fmt = 0) {
  __shared__ __attribute__((aligned(16))) uint16_t Ks[KPANEL];
  __shared__ __attribute__((aligned(16))) uint16_t Vt[PANEL];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const int q0 = 64 * (NWV * blockIdx.x + wv);                 // this wave's queries
  const float c = scale * LOG2E;

  u32x4_t fq[2][4];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int s = 0; s < 4; ++s) fq[qt][s] = rowfrag(Q, ld, q0 + 32 * qt + l31, T, s, h);
  f32x16_t OT[2][2];
  zero_tiles(OT);
  float m[2] = {NEG_BIG, NEG_BIG}, l[2] = {0.0f, 0.0f};
  const int qend = min(T, 64 * NWV * ((int)blockIdx.x + 1));    // one past the workgroup's last query
  const int nkb = CAUSAL ? (qend + 63) / 64 : (T + 63) / 64;
  const int my_last = CAUSAL ? q0 / 64 : nkb - 1;              // last key block this wave needs
  BlockLoader<NWV> lk, lv;
  lk.load(Kp, ld, 0, T, tid);
  lv.load(V, ld, 0, T, tid);
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = 64 * kb;
    __syncthreads();                                           // everyone is done with the previous block's panels
    lk.store_rows(Ks, tid);
    lv.store_transposed(Vt, tid);
    __syncthreads();
    if (kb + 1 < nkb) {                                        // next block's chunks fly under this block's MFMAs
      lk.load(Kp, ld, k0 + 64, T, tid);
      lv.load(V, ld, k0 + 64, T, tid);
    }
    if (kb > my_last || q0 >= T) continue;
    f32x16_t ST[2][2];
    zero_tiles(ST);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4_t fa[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = ldsfrag(Ks, 32 * a + l31, s, h);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) mma(ST[a][qt], fa[a], fq[qt][s]);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int q = q0 + 32 * qt + l31;
      const int klim = CAUSAL ? min(T, q + 1) : T;
      float mloc = NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + 32 * kt + reg_index(r, h) < klim) mloc = fmaxf(mloc, ST[kt][qt][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(m[qt], mloc);
      const float alpha = __builtin_amdgcn_exp2f((m[qt] - mnew) * c);
      float sum = 0.0f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p =
              (k0 + 32 * kt + reg_index(r, h) < klim) ? __builtin_amdgcn_exp2f((ST[kt][qt][r] - mnew) * c) : 0.0f;
          ST[kt][qt][r] = p;
          sum += p;
        }
      sum += __shfl_xor(sum, 32, 64);
      l[qt] = l[qt] * alpha + sum;
      m[qt] = mnew;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) OT[dt][qt][r] *= alpha;
    }
    lds_product<L, 2, 2>(OT, Vt, ST, 0, lane);
  }
  float f8m = 0.0f;
  if (q0 < T) {
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const float inv = l[qt] > 0.0f ? 1.0f / l[qt] : 0.0f;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) OT[dt][qt][r] *= inv;
      const int q = q0 + 32 * qt + l31;
      if (h == 0 && q < T) lse[(int64_t)bh * T + q] = m[qt] * c + __builtin_amdgcn_logf(l[qt]);
    }
    store_transposed<L, 2>(o + (int64_t)b * T * D + hd * 64, D, OT, q0, T, lane);
    if (o8) store_transposed_f8<L, 2>(o8 + (int64_t)b * T * D + hd * 64, D, OT, q0, T, lane, f8_state[0], f8_fmt, f8m);
  }
  if (o8) f8_amax_block(f8m, f8_state);      // every wave of the workgroup arrives here (also those past the sequence end)
}


// dQ, second generation: NWV waves (64 queries each) share the K / V key blocks through LDS (K rows, V rows, K^T).
template <typename L, bool CAUSAL, int NWV>
__global__ __launch_bounds__(64 * NWV, 2) void attn_flash2_bwd_dq_kernel(const uint16_t* __restrict__ qkv,
                                                                         const uint16_t* __restrict__ out,
                                                                         const uint16_t* __restrict__ dout,
                                                                         const float* __restrict__ lse,
                                                                         float* __restrict__ delta, uint16_t* __restrict__ dqkv,
                                                                         int T, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t Ks[KPANEL];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[KPANEL];
  __shared__ __attribute__((aligned(16))) uint16_t Kt[PANEL];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const uint16_t* O = out + (int64_t)b * T * D + hd * 64;
  const uint16_t* dO = dout + (int64_t)b * T * D + hd * 64;
  const int q0 = 64 * (NWV * blockIdx.x + wv);
  const float c = scale * LOG2E;

  u32x4_t fq[2][4], fdo[2][4];
  float dl[2], ls[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + 32 * qt + l31;
    float acc = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fq[qt][s] = rowfrag(Q, ld, q, T, s, h);
      fdo[qt][s] = rowfrag(dO, D, q, T, s, h);
      const u32x4_t fo = rowfrag(O, D, q, T, s, h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc += lo_unpack<L>((uint16_t)(fo[e] & 0xffffu)) * lo_unpack<L>((uint16_t)(fdo[qt][s][e] & 0xffffu));
        acc += lo_unpack<L>((uint16_t)(fo[e] >> 16)) * lo_unpack<L>((uint16_t)(fdo[qt][s][e] >> 16));
      }
    }
    acc += __shfl_xor(acc, 32, 64);
    dl[qt] = acc;
    ls[qt] = q < T ? lse[(int64_t)bh * T + q] : 0.0f;
    if (h == 0 && q < T) delta[(int64_t)bh * T + q] = acc;
  }
  f32x16_t dQT[2][2];
  zero_tiles(dQT);
  const int qend = min(T, 64 * NWV * ((int)blockIdx.x + 1));
  const int nkb = CAUSAL ? (qend + 63) / 64 : (T + 63) / 64;
  const int my_last = CAUSAL ? q0 / 64 : nkb - 1;
  BlockLoader<NWV> lk, lv;
  lk.load(Kp, ld, 0, T, tid);
  lv.load(V, ld, 0, T, tid);
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int k0 = 64 * kb;
    __syncthreads();
    lk.store_rows(Ks, tid);
    lk.store_transposed(Kt, tid);
    lv.store_rows(Vs, tid);
    __syncthreads();
    if (kb + 1 < nkb) {
      lk.load(Kp, ld, k0 + 64, T, tid);
      lv.load(V, ld, k0 + 64, T, tid);
    }
    if (kb > my_last || q0 >= T) continue;
#pragma unroll 1
    for (int kt = 0; kt < 2; ++kt) {
      f32x16_t PT[1][2], dPT[1][2];
      zero_tiles(PT);
      zero_tiles(dPT);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4_t fk = ldsfrag(Ks, 32 * kt + l31, s, h);
        const u32x4_t fv = ldsfrag(Vs, 32 * kt + l31, s, h);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          mma(PT[0][qt], fk, fq[qt][s]);
          mma(dPT[0][qt], fv, fdo[qt][s]);
        }
      }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const int q = q0 + 32 * qt + l31;
        const int klim = q < T ? (CAUSAL ? min(T, q + 1) : T) : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p =
              (k0 + 32 * kt + reg_index(r, h) < klim) ? __builtin_amdgcn_exp2f(PT[0][qt][r] * c - ls[qt]) : 0.0f;
          dPT[0][qt][r] = p * (dPT[0][qt][r] - dl[qt]) * scale;
        }
      }
      lds_product<L, 1, 2>(dQT, Kt, dPT, 32 * kt, lane);
    }
  }
  if (q0 < T) store_transposed<L, 2>(dqkv + (int64_t)b * T * ld + hd * 64, ld, dQT, q0, T, lane);
}

// dK, dV, second generation: NWV waves (64 keys each) share the Q / dO query blocks through LDS (rows and transposed).
template <typename L, bool CAUSAL, int NWV>
__global__ __launch_bounds__(64 * NWV) void attn_flash2_bwd_dkv_kernel(const uint16_t* __restrict__ qkv,
                                                                          const uint16_t* __restrict__ dout,
                                                                          const float* __restrict__ lse,
                                                                          const float* __restrict__ delta,
                                                                          uint16_t* __restrict__ dqkv, int T, int heads,
                                                                          float scale) {
  __shared__ __attribute__((aligned(16))) uint16_t Qs[KPANEL];
  __shared__ __attribute__((aligned(16))) uint16_t dOs[KPANEL];
  __shared__ __attribute__((aligned(16))) uint16_t Qt[PANEL];
  __shared__ __attribute__((aligned(16))) uint16_t dOt[PANEL];
  __shared__ __attribute__((aligned(16))) float stat[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int D = heads * 64;
  const int64_t ld = 3 * (int64_t)D;
  const uint16_t* Q = qkv + (int64_t)b * T * ld + hd * 64;
  const uint16_t* Kp = Q + D;
  const uint16_t* V = Q + 2 * D;
  const uint16_t* dO = dout + (int64_t)b * T * D + hd * 64;
  const int k0 = 64 * (NWV * blockIdx.x + wv);                 // this wave's keys
  const float c = scale * LOG2E;

  u32x4_t fk[2][4], fv[2][4];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fk[kt][s] = rowfrag(Kp, ld, k0 + 32 * kt + l31, T, s, h);
      fv[kt][s] = rowfrag(V, ld, k0 + 32 * kt + l31, T, s, h);
    }
  f32x16_t dVT[2][2], dKT[2][2];
  zero_tiles(dVT);
  zero_tiles(dKT);
  const int nqb = (T + 63) / 64;
  const int qb0 = CAUSAL ? NWV * (int)blockIdx.x : 0;          // first query block that sees any key of this workgroup
  const int my_first = CAUSAL ? k0 / 64 : 0;
  BlockLoader<NWV> lq, ldo;
  lq.load(Q, ld, 64 * qb0, T, tid);
  ldo.load(dO, D, 64 * qb0, T, tid);
#pragma unroll 1
  for (int qb = qb0; qb < nqb; ++qb) {
    const int q0 = 64 * qb;
    __syncthreads();
    lq.store_rows(Qs, tid);
    lq.store_transposed(Qt, tid);
    ldo.store_rows(dOs, tid);
    ldo.store_transposed(dOt, tid);
    if (tid < 64) {
      const int q = q0 + tid;
      stat[0][tid] = q < T ? lse[(int64_t)bh * T + q] : 0.0f;
      stat[1][tid] = q < T ? delta[(int64_t)bh * T + q] : 0.0f;
    }
    __syncthreads();
    if (qb + 1 < nqb) {
      lq.load(Q, ld, q0 + 64, T, tid);
      ldo.load(dO, D, q0 + 64, T, tid);
    }
    if (qb < my_first || k0 >= T) continue;
#pragma unroll 1
    for (int qt = 0; qt < 2; ++qt) {
      if (q0 + 32 * qt >= T) break;
      f32x16_t S[1][2], dP[1][2];
      zero_tiles(S);
      zero_tiles(dP);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32x4_t fa = ldsfrag(Qs, 32 * qt + l31, s, h);
        const u32x4_t fb = ldsfrag(dOs, 32 * qt + l31, s, h);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          mma(S[0][kt], fa, fk[kt][s]);
          mma(dP[0][kt], fb, fv[kt][s]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qi = 32 * qt + 8 * g + 4 * h;
        const f32x4_t l4 = *(const f32x4_t*)&stat[0][qi];
        const f32x4_t d4 = *(const f32x4_t*)&stat[1][qi];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int key = k0 + 32 * kt + l31;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * g + j;
            const int q = q0 + qi + j;
            const bool ok = key < T && q < T && (!CAUSAL || key <= q);
            const float p = ok ? __builtin_amdgcn_exp2f(S[0][kt][r] * c - l4[j]) : 0.0f;
            S[0][kt][r] = p;
            dP[0][kt][r] = p * (dP[0][kt][r] - d4[j]) * scale;
          }
        }
      }
      lds_product<L, 1, 2>(dVT, dOt, S, 32 * qt, lane);
      lds_product<L, 1, 2>(dKT, Qt, dP, 32 * qt, lane);
    }
  }
  if (k0 >= T) return;
  uint16_t* dKp = dqkv + (int64_t)b * T * ld + hd * 64 + D;
  store_transposed<L, 2>(dKp + D, ld, dVT, k0, T, lane);
  store_transposed<L, 2>(dKp, ld, dKT, k0, T, lane);
}

#undef mma

}  // namespace

extern "C" int ffvc_attn_small_fwd(const void* qkv, void* out, int dtype, int B, int T, int heads, int head_dim,
                                   float scale, void* stream) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16, "ffvc_attn_small_fwd: 16-bit storage only (dtype %d)", dtype);
  FFVC_CHECK_ARG(qkv && out && B > 0 && heads > 0, "ffvc_attn_small_fwd: bad args");
  FFVC_CHECK_ARG(T >= 1 && T <= 64 && head_dim == 64, "ffvc_attn_small_fwd: needs T <= 64 and head_dim == 64 (T=%d, dh=%d)",
                 T, head_dim);
  FFVC_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0, "ffvc_attn_small_fwd: misaligned pointers");
  if (dtype == FFVC_F16)
    hipLaunchKernelGGL(attn_small_fwd_kernel<f16_t>, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                       (uint16_t*)out, T, heads, scale);
  else
    hipLaunchKernelGGL(attn_small_fwd_kernel<uint16_t>, dim3(B * heads), dim3(64), 0, (hipStream_t)stream,
                       (const uint16_t*)qkv, (uint16_t*)out, T, heads, scale);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_attn_small_bwd(const void* qkv, const void* dout, void* dqkv, int dtype, int B, int T, int heads,
                                   int head_dim, float scale, void* stream) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16, "ffvc_attn_small_bwd: 16-bit storage only (dtype %d)", dtype);
  FFVC_CHECK_ARG(qkv && dout && dqkv && B > 0 && heads > 0, "ffvc_attn_small_bwd: bad args");
  FFVC_CHECK_ARG(T >= 1 && T <= 64 && head_dim == 64, "ffvc_attn_small_bwd: needs T <= 64 and head_dim == 64 (T=%d, dh=%d)",
                 T, head_dim);
  FFVC_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)dqkv % 8) == 0,
                 "ffvc_attn_small_bwd: misaligned pointers");
  if (dtype == FFVC_F16)
    hipLaunchKernelGGL(attn_small_bwd_kernel<f16_t>, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)qkv,
                       (const uint16_t*)dout, (uint16_t*)dqkv, T, heads, scale);
  else
    hipLaunchKernelGGL(attn_small_bwd_kernel<uint16_t>, dim3(B * heads), dim3(64), 0, (hipStream_t)stream,
                       (const uint16_t*)qkv, (const uint16_t*)dout, (uint16_t*)dqkv, T, heads, scale);
  FFVC_LAUNCH_CHECK();
  return 0;
}

static int flash_gen() {               // FFVC_FLASH_GEN=1: the one-wave first-generation kernels (A/B runs)
  static int g = -1;
  if (g < 0) {
    const char* e = getenv("FFVC_FLASH_GEN");
    g = e ? atoi(e) : 2;
  }
  return g;
}

template <typename L>
static void flash_fwd_launch(const void* qkv, void* out, float* lse, int B, int T, int heads, float scale, int causal,
                             hipStream_t st, uint8_t* o8 = nullptr, float* f8_state = nullptr, int f8_fmt = 0) {
  if (flash_gen() >= 2 || o8) {
    if (T > 320) {         // 4 waves share every key block; short sequences (257 tokens) waste fewer query slots with 2
      const dim3 g4((T + 255) / 256, B * heads);
      if (causal)
        hipLaunchKernelGGL((attn_flash2_fwd_kernel<L, true, 4>), g4, dim3(256), 0, st, (const uint16_t*)qkv, (uint16_t*)out, lse,
                           T, heads, scale, o8, f8_state, f8_fmt);
      else
        hipLaunchKernelGGL((attn_flash2_fwd_kernel<L, false, 4>), g4, dim3(256), 0, st, (const uint16_t*)qkv, (uint16_t*)out, lse,
                           T, heads, scale, o8, f8_state, f8_fmt);
    } else {
      const dim3 g2((T + 127) / 128, B * heads);
      if (causal)
        hipLaunchKernelGGL((attn_flash2_fwd_kernel<L, true, 2>), g2, dim3(128), 0, st, (const uint16_t*)qkv, (uint16_t*)out, lse,
                           T, heads, scale, o8, f8_state, f8_fmt);
      else
        hipLaunchKernelGGL((attn_flash2_fwd_kernel<L, false, 2>), g2, dim3(128), 0, st, (const uint16_t*)qkv, (uint16_t*)out, lse,
                           T, heads, scale, o8, f8_state, f8_fmt);
    }
    return;
  }
  const dim3 grid((T + 63) / 64, B * heads);
  if (causal)
    hipLaunchKernelGGL((attn_flash_fwd_kernel<L, true>), grid, dim3(64), 0, st, (const uint16_t*)qkv, (uint16_t*)out, lse, T,
                       heads, scale);
  else
    hipLaunchKernelGGL((attn_flash_fwd_kernel<L, false>), grid, dim3(64), 0, st, (const uint16_t*)qkv, (uint16_t*)out, lse, T,
                       heads, scale);
}

template <typename L>
static void flash_bwd_launch(const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv,
                             int B, int T, int heads, float scale, int causal, hipStream_t st) {
  if (flash_gen() >= 2) {
#define FFVC_FLASH2_BWD(CAUS, NWV)                                                                                          \
  {                                                                                                                         \
    const dim3 g((T + 64 * NWV - 1) / (64 * NWV), B * heads);                                                               \
    hipLaunchKernelGGL((attn_flash2_bwd_dq_kernel<L, CAUS, NWV>), g, dim3(64 * NWV), 0, st, (const uint16_t*)qkv,            \
                       (const uint16_t*)out, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, T, heads, scale);          \
    hipLaunchKernelGGL((attn_flash2_bwd_dkv_kernel<L, CAUS, NWV>), g, dim3(64 * NWV), 0, st, (const uint16_t*)qkv,           \
                       (const uint16_t*)dout, lse, (const float*)delta, (uint16_t*)dqkv, T, heads, scale);                  \
  }
    if (T > 320) {
      if (causal) FFVC_FLASH2_BWD(true, 4) else FFVC_FLASH2_BWD(false, 4)
    } else {
      if (causal) FFVC_FLASH2_BWD(true, 2) else FFVC_FLASH2_BWD(false, 2)
    }
#undef FFVC_FLASH2_BWD
    return;
  }
  const dim3 grid((T + 63) / 64, B * heads);
  if (causal) {
    hipLaunchKernelGGL((attn_flash_bwd_dq_kernel<L, true>), grid, dim3(64), 0, st, (const uint16_t*)qkv, (const uint16_t*)out,
                       (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, T, heads, scale);
    hipLaunchKernelGGL((attn_flash_bwd_dkv_kernel<L, true>), grid, dim3(64), 0, st, (const uint16_t*)qkv,
                       (const uint16_t*)dout, lse, (const float*)delta, (uint16_t*)dqkv, T, heads, scale);
  } else {
    hipLaunchKernelGGL((attn_flash_bwd_dq_kernel<L, false>), grid, dim3(64), 0, st, (const uint16_t*)qkv,
                       (const uint16_t*)out, (const uint16_t*)dout, lse, delta, (uint16_t*)dqkv, T, heads, scale);
    hipLaunchKernelGGL((attn_flash_bwd_dkv_kernel<L, false>), grid, dim3(64), 0, st, (const uint16_t*)qkv,
                       (const uint16_t*)dout, lse, (const float*)delta, (uint16_t*)dqkv, T, heads, scale);
  }
}

extern "C" int ffvc_attn_flash_fwd(const void* qkv, void* out, void* lse, int dtype, int B, int T, int heads, int head_dim,
                                   float scale, int causal, void* stream) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16, "ffvc_attn_flash_fwd: 16-bit storage only (dtype %d)", dtype);
  FFVC_CHECK_ARG(qkv && out && lse && B > 0 && heads > 0 && T > 0, "ffvc_attn_flash_fwd: bad args");
  FFVC_CHECK_ARG(head_dim == 64, "ffvc_attn_flash_fwd: head_dim must be 64 (got %d)", head_dim);
  FFVC_CHECK_ARG((int64_t)B * heads <= 65535, "ffvc_attn_flash_fwd: B*heads = %lld exceeds the grid limit",
                 (long long)B * heads);
  FFVC_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0, "ffvc_attn_flash_fwd: misaligned pointers");
  if (dtype == FFVC_F16)
    flash_fwd_launch<f16_t>(qkv, out, (float*)lse, B, T, heads, scale, causal, (hipStream_t)stream);
  else
    flash_fwd_launch<uint16_t>(qkv, out, (float*)lse, B, T, heads, scale, causal, (hipStream_t)stream);
  FFVC_LAUNCH_CHECK();
  return 0;
}

// ffvc_attn_flash_fwd whose output ALSO leaves as fp8 bytes out8 [B, T, heads*64] = saturate(round_16(out) * f8_state[0]) (f8_fmt 0 e4m3 |
// 1 e5m2; f8_state[1] collects max |round_16(out)|): the operand of the fp8 out_proj behind it, byte for byte what ffvc_fp8_quant makes
// of `out` (which the backward pass still needs in 16 bits).
extern "C" int ffvc_attn_flash_fwd_f8(const void* qkv, void* out, void* lse, void* out8, float* f8_state, int f8_fmt, int dtype, int B, int T,
                                      int heads, int head_dim, float scale, int causal, void* stream) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16, "ffvc_attn_flash_fwd_f8: 16-bit storage only (dtype %d)", dtype);
  FFVC_CHECK_ARG(qkv && out && lse && out8 && f8_state && B > 0 && heads > 0 && T > 0, "ffvc_attn_flash_fwd_f8: bad args");
  FFVC_CHECK_ARG(head_dim == 64, "ffvc_attn_flash_fwd_f8: head_dim must be 64 (got %d)", head_dim);
  FFVC_CHECK_ARG(f8_fmt == 0 || f8_fmt == 1, "ffvc_attn_flash_fwd_f8: f8_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG((int64_t)B * heads <= 65535, "ffvc_attn_flash_fwd_f8: B*heads = %lld exceeds the grid limit", (long long)B * heads);
  FFVC_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0 && ((uintptr_t)out8 % 4) == 0,
                 "ffvc_attn_flash_fwd_f8: misaligned pointers");
  if (dtype == FFVC_F16)
    flash_fwd_launch<f16_t>(qkv, out, (float*)lse, B, T, heads, scale, causal, (hipStream_t)stream, (uint8_t*)out8, f8_state, f8_fmt);
  else
    flash_fwd_launch<uint16_t>(qkv, out, (float*)lse, B, T, heads, scale, causal, (hipStream_t)stream, (uint8_t*)out8, f8_state, f8_fmt);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_attn_flash_bwd(const void* qkv, const void* out, const void* dout, const void* lse, void* delta_ws,
                                   void* dqkv, int dtype, int B, int T, int heads, int head_dim, float scale, int causal,
                                   void* stream) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16, "ffvc_attn_flash_bwd: 16-bit storage only (dtype %d)", dtype);
  FFVC_CHECK_ARG(qkv && out && dout && lse && delta_ws && dqkv && B > 0 && heads > 0 && T > 0, "ffvc_attn_flash_bwd: bad args");
  FFVC_CHECK_ARG(head_dim == 64, "ffvc_attn_flash_bwd: head_dim must be 64 (got %d)", head_dim);
  FFVC_CHECK_ARG((int64_t)B * heads <= 65535, "ffvc_attn_flash_bwd: B*heads = %lld exceeds the grid limit",
                 (long long)B * heads);
  FFVC_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)dout % 16) == 0 &&
                     ((uintptr_t)dqkv % 8) == 0,
                 "ffvc_attn_flash_bwd: misaligned pointers");
  if (dtype == FFVC_F16)
    flash_bwd_launch<f16_t>(qkv, out, dout, (const float*)lse, (float*)delta_ws, dqkv, B, T, heads, scale, causal,
                            (hipStream_t)stream);
  else
    flash_bwd_launch<uint16_t>(qkv, out, dout, (const float*)lse, (float*)delta_ws, dqkv, B, T, heads, scale, causal,
                               (hipStream_t)stream);
  FFVC_LAUNCH_CHECK();
  return 0;
}
