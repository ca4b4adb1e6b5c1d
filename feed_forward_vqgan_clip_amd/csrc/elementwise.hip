// elementwise.hip — HBM-bound glue kernels of the train step (main.py:715-837): casts and
// transposes, bias-gradient column sums, clamp-with-grad, VQ argmin + gather (STE), cutout
// pooling + noise + CLIP normalisation written straight into ViT patch layout, spherical loss,
// fused Adam.  All arithmetic fp32; coalesced 8/16-byte accesses; grid-stride loops capped at
// ~8 workgroups per CU.
#include <stdlib.h>
#include "common.h"
#include "augment_cj.h"

namespace {

inline int ew_grid(int64_t n, int per_block) {
  int64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (int)g;
}

// ------------------------------- cast --------------------------------------
template <typename ST, typename DT>
__global__ __launch_bounds__(256) void cast_kernel(const ST* __restrict__ s, DT* __restrict__ d, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    store4(d + i * 4, load4(s + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    ElemTraits<DT>::store(d + i, ElemTraits<ST>::load(s + i));
  }
}

// ---------------------- fp32 -> three 16-bit segments ----------------------
// Split-precision operand of an fp32-grade GEMM on the 16-bit matrix pipes: x = hi + lo (+ ~2^-22 |x|) with hi = rn16(x),
// lo = rn16(x - hi).  With the activation written as [hi | lo | hi] and the weight as [hi | hi | lo] along K, ONE 16-bit GEMM
// of depth 3K accumulates x_hi w_hi + x_lo w_hi + x_hi w_lo in fp32: everything but the lo x lo term.  SEG selects the layout:
// 0 = activation order [hi | lo | hi], 1 = weight order [hi | hi | lo].
template <typename DT>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ s, DT* __restrict__ d, int64_t rows, int K,
                                                     int64_t lds, int seg) {
  const int k4 = K >> 2;
  const int64_t n4 = rows * k4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / k4;
    const int c = (int)(i - r * k4) * 4;
    const f32x4_t v = load4(s + r * lds + c);
    f32x4_t hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      hi[j] = lo_round<DT>(v[j]);
      lo[j] = v[j] - hi[j];
    }
    DT* o = d + r * (3 * (int64_t)K) + c;
    store4(o, hi);
    store4(o + K, seg ? hi : lo);
    store4(o + 2 * (int64_t)K, seg ? lo : hi);
  }
}

// ------------------------- batched 2-D transpose ---------------------------
// dst[b][c][r] = src[b][r][c] with dtype conversion (weight shadows W^T, mixer rearrange).
template <typename ST, typename DT>
__global__ __launch_bounds__(256) void transpose_kernel(const ST* __restrict__ s, DT* __restrict__ d, int rows,
                                                        int cols, int64_t sb, int64_t db, int dld) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? ElemTraits<ST>::load(s + b * sb + (int64_t)r * cols + c) : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < dld) ElemTraits<DT>::store(d + b * db + (int64_t)c * dld + r, r < rows ? tile[tx][i] : 0.f);
  }
}

// Many independent transposes in ONE launch (the W^T shadows of every trainable weight after an optimizer step: 177
// matrices for the 32-layer Mixer).  tile_prefix[i] = number of 64x64 tiles of items 0..i-1.
template <typename T>
__global__ __launch_bounds__(256) void transpose_multi_kernel(const ffvc_tr_item* __restrict__ items,
                                                              const int* __restrict__ tile_prefix, int n_items) {
  __shared__ T tile[64][66];
  int lo = 0, hi = n_items - 1;
  const int g = blockIdx.x;
  while (lo < hi) {                       // last item whose prefix <= g
    const int mid = (lo + hi + 1) >> 1;
    if (tile_prefix[mid] <= g) lo = mid; else hi = mid - 1;
  }
  const ffvc_tr_item it = items[lo];
  const int t = g - tile_prefix[lo];
  const int tiles_x = (it.cols + 63) >> 6;
  const int r0 = (t / tiles_x) * 64, c0 = (t % tiles_x) * 64;
  const T* s = (const T*)it.src;
  T* d = (T*)it.dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    if (r < it.rows && c < it.cols) tile[i][tx] = s[(int64_t)r * it.cols + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < it.cols && r < it.rows) d[(int64_t)c * it.rows + r] = tile[tx][i];
  }
}

// ---------------------------- column sums ----------------------------------
// out[c] (+)= sum_r x[r, c]   (bias gradients; reduction of LayerNorm partials)
// block = 64 column-quads x 4 row-lanes (8/16-byte loads); gridDim.y strips of rows, atomics combine strips.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                     int cols, int64_t ld, int rows_per_block, int vec) {
  __shared__ float red[4][256];
  const int tx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    const int c = (blockIdx.x * 64 + tx) * 4;
    if (c < cols) {
      // eight rows in flight per lane: the loop is a chain of dependent global loads otherwise (one ~2 us round trip per 4 rows), and
      // hiding that with ~1024 workgroups meant as many fp32 atomics per output column (75 ns each when they hit one address)
      int64_t r = r0 + ry;
      for (; r + 28 < r1; r += 32) {
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = load4(x + (r + 4 * u) * ld + c);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] += v[u][j];
      }
      for (; r < r1; r += 4) {
        const f32x4_t v = load4(x + r * ld + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += v[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[ry][tx * 4 + j] = a[j];
    __syncthreads();
    if (ry == 0 && c < cols) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        atomicAdd(out + c + j, red[0][tx * 4 + j] + red[1][tx * 4 + j] + red[2][tx * 4 + j] + red[3][tx * 4 + j]);
    }
  } else {
    const int c = blockIdx.x * 64 + tx;
    if (c < cols)
      for (int64_t r = r0 + ry; r < r1; r += 4) a[0] += ElemTraits<T>::load(x + r * ld + c);
    red[ry][tx] = a[0];
    __syncthreads();
    if (ry == 0 && c < cols) atomicAdd(out + c, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
  }
}

// y[i] (+)= sum_s slab[s][i]  — combine of split-K partial slabs (plain vector stores instead of scalar fp32
// atomics, each of which is its own fabric transaction)
// NS > 0: slab count fixed at compile time — every slab's 16-byte load of an element is in flight before the first add
// (a runtime loop issued them one by one: 2.4 TB/s on the 4-slab weight-gradient reduces, round 2); NS == 0: any count.
template <int NS>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ y,
                                                          int64_t n, int nslab, int accumulate) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4_t a = accumulate ? load4(y + i * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (NS > 0) {
      f32x4_t v[NS];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) v[s2] = load4(slabs + (int64_t)s2 * n + i * 4);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) a += v[s2];
    } else {
      for (int s2 = 0; s2 < nslab; ++s2) a += load4(slabs + (int64_t)s2 * n + i * 4);
    }
    store4(y + i * 4, a);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    float a = accumulate ? y[i] : 0.f;
    for (int s2 = 0; s2 < nslab; ++s2) a += slabs[(int64_t)s2 * n + i];
    y[i] = a;
  }
}

// -------------------------- clamp with grad --------------------------------
// main.py:118-132: y = clamp(x*mul + add, lo, hi); backward passes g unless it pushes an
// out-of-range value further out: dx = mul * g * [g * (u - clamp(u)) >= 0], u = x*mul + add.
template <typename XT, typename YT>
__global__ __launch_bounds__(256) void clamp_fwd_kernel(const XT* __restrict__ x, YT* __restrict__ y, int64_t n,
                                                        float mul, float add, float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float u = ElemTraits<XT>::load(x + i) * mul + add;
    ElemTraits<YT>::store(y + i, fminf(fmaxf(u, lo), hi));
  }
}
template <typename XT, typename GT>
__global__ __launch_bounds__(256) void clamp_bwd_kernel(const XT* __restrict__ x, const GT* __restrict__ g,
                                                        XT* __restrict__ dx, int64_t n, float mul, float add,
                                                        float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float u = ElemTraits<XT>::load(x + i) * mul + add;
    const float gv = ElemTraits<GT>::load(g + i);
    const float c = fminf(fmaxf(u, lo), hi);
    ElemTraits<XT>::store(dx + i, (gv * (u - c) >= 0.f) ? gv * mul : 0.f);
  }
}

// ------------------- 2x2 sum pool (backward of nearest 2x upsample) --------
template <typename T>
__global__ __launch_bounds__(256) void sumpool2_kernel(const T* __restrict__ s, T* __restrict__ d, int B, int H, int W,
                                                       int C) {
  // s: [B, 2H, 2W, C] -> d: [B, H, W, C]; 4 channels per thread
  const int c4 = C >> 2;
  const int64_t n = (int64_t)B * H * W * c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    int64_t p = i / c4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int b = (int)(p / H);
    const T* base = s + (((int64_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
    f32x4_t a = load4(base);
    a += load4(base + C);
    a += load4(base + (int64_t)2 * W * C);
    a += load4(base + (int64_t)2 * W * C + C);
    store4(d + (((int64_t)b * H + y) * W + x) * C + c, a);
  }
}

// ------------------------------- VQ ----------------------------------------
// row squared norms (||c_j||^2 once for the frozen codebook, ||x_i||^2 per step)
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                      int dim) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float a = 0.f;
    for (int i = lane; i < dim; i += 64) {
      const float v = x[r * dim + i];
      a += v * v;
    }
    a = wave_sum(a);
    if (lane == 0) out[r] = a;
  }
}
// idx[i] = argmin_j (xn[i] + cn[j]) - 2 * dot[i, j]   — same fp32 expression order as
// main.py:135 `x.pow(2).sum + codebook.pow(2).sum - 2 * x @ codebook.T`; first index wins ties (:136).
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ dot, const float* __restrict__ xn,
                                                        const float* __restrict__ cn, int64_t* __restrict__ idx,
                                                        int64_t rows, int ncodes, int64_t ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    const float x2 = xn[r];
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < ncodes; j += 64) {
      const float d = (x2 + cn[j]) - 2.0f * dot[r * ld + j];
      if (d < best) {
        best = d;
        bi = j;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob < best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (lane == 0) idx[r] = bi;
  }
}
// out[r, :] = table[idx[r], :] (+ pos[r % period, :])  — codebook gather (== one_hot @ codebook,
// main.py:137), token embedding + positional embedding (cloob.py:526-528).
template <typename DT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table,
                                                          const int64_t* __restrict__ idx, const float* __restrict__ pos,
                                                          int period, DT* __restrict__ out, int64_t rows, int dim) {
  const int64_t n = rows * dim;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / dim;
    const int c = (int)(i - r * dim);
    float v = table[idx[r] * dim + c];
    if (pos) v += pos[(r % period) * dim + c];
    ElemTraits<DT>::store(out + i, v);
  }
}
// out[b, :] = x[b, argmax_t tok[b, t], :]   (EOT pooling, cloob.py:536)
template <typename XT>
__global__ __launch_bounds__(64) void eot_gather_kernel(const XT* __restrict__ x, const int64_t* __restrict__ tok,
                                                        float* __restrict__ out, int L, int dim) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int64_t best = INT64_MIN;
  int bi = 0x7fffffff;
  for (int t = lane; t < L; t += 64) {
    const int64_t v = tok[(int64_t)b * L + t];
    if (v > best) {
      best = v;
      bi = t;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int64_t ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      bi = oi;
    }
  }
  for (int c = lane; c < dim; c += 64) out[(int64_t)b * dim + c] = ElemTraits<XT>::load(x + ((int64_t)b * L + bi) * dim + c);
}

// ------------------------------ cutouts ------------------------------------
// main.py:212-229 (pool branch, augs=['R'] at pool_size == cut_size) fused with :797:
//   c = (AdaptiveAvgPool(x) + AdaptiveMaxPool(x)) / 2 ; repeat cutn (cut-major) ;
//   + facs[n] * noise[n] ; (v - mean[ch]) / std[ch]
// and written directly as ViT patch rows: out[n, py*gw+px, ch*P*P + ky*P + kx] (the im2col of the
// stride-P patch-embedding conv, cloob.py:224,237), so the patch GEMM reads it K-major.
// xr is NHWC fp32 [B, H, W, 3]; noise is NCHW [cutn*B, 3, cut, cut] fp32 or NULL.
// (32-bit unsigned arithmetic: a 64-bit division is ~150 instructions on this ISA and the backward does two dozen of them per input element;
// exact for in, out <= 32768 — the launchers check)
__device__ __forceinline__ int apool_start(int o, int in, int out) { return (int)(((uint32_t)o * (uint32_t)in) / (uint32_t)out); }
__device__ __forceinline__ int apool_end(int o, int in, int out) {
  return (int)(((uint32_t)(o + 1) * (uint32_t)in + (uint32_t)out - 1u) / (uint32_t)out);
}
constexpr int APOOL_MAX = 32768;

template <typename OT>
__global__ __launch_bounds__(256) void cutouts_fwd_kernel(const float* __restrict__ xr, const float* __restrict__ noise,
                                                          const float* __restrict__ facs, OT* __restrict__ out, int B,
                                                          int H, int W, int cut, int cutn, int P, float m0, float m1,
                                                          float m2, float s0, float s1, float s2) {
  const int gw = cut / P;
  const int64_t n = (int64_t)B * 3 * cut * cut;
  const int64_t per_img = (int64_t)3 * cut * cut;
  const bool small = n <= 0x7fffffffll;      // 32-bit index arithmetic whenever the tensor allows it
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int ox, oy, ch, b;
    if (small) {
      uint32_t t = (uint32_t)i;
      ox = (int)(t % (uint32_t)cut);
      t /= (uint32_t)cut;
      oy = (int)(t % (uint32_t)cut);
      t /= (uint32_t)cut;
      ch = (int)(t % 3u);
      b = (int)(t / 3u);
    } else {
      ox = (int)(i % cut);
      int64_t t = i / cut;
      oy = (int)(t % cut);
      t /= cut;
      ch = (int)(t % 3);
      b = (int)(t / 3);
    }
    const int y0 = apool_start(oy, H, cut), y1 = apool_end(oy, H, cut);
    const int x0 = apool_start(ox, W, cut), x1 = apool_end(ox, W, cut);
    float sum = 0.f, mx = -INFINITY;
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) {
        const float v = xr[(((int64_t)b * H + y) * W + x) * 3 + ch];
        sum += v;
        mx = fmaxf(mx, v);
      }
    const float pooled = (sum / (float)((y1 - y0) * (x1 - x0)) + mx) * 0.5f;
    const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
    const float istd = 1.0f / (ch == 0 ? s0 : (ch == 1 ? s1 : s2));
    const int py = oy / P, ky = oy - py * P, px = ox / P, kx = ox - px * P;
    const int64_t prow = (int64_t)(py * gw + px) * (3 * P * P) + (int64_t)ch * P * P + ky * P + kx;
    for (int c = 0; c < cutn; ++c) {
      const int64_t nimg = (int64_t)c * B + b;
      float v = pooled;
      if (noise) v += facs[nimg] * noise[nimg * per_img + ((int64_t)ch * cut + oy) * cut + ox];
      ElemTraits<OT>::store(out + nimg * per_img + prow, (v - mean) * istd);
    }
  }
}

// dxr[b, y, x, ch] = sum over cuts and over the pooling windows containing (y, x) of
//   g/std * (0.5/|win| + 0.5*[argmax == (y,x)])    (first max in row-major order, torch semantics)
template <typename GT>
__global__ __launch_bounds__(256) void cutouts_bwd_kernel(const float* __restrict__ xr, const GT* __restrict__ gout,
                                                          float* __restrict__ dxr, int B, int H, int W, int cut,
                                                          int cutn, int P, float s0, float s1, float s2) {
  const int gw = cut / P;
  const int64_t n = (int64_t)B * H * W * 3;
  const int64_t per_img = (int64_t)3 * cut * cut;
  const bool small = n <= 0x7fffffffll;      // 32-bit index arithmetic whenever the tensor allows it
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int ch, x, y, b;
    if (small) {
      uint32_t t = (uint32_t)i;
      ch = (int)(t % 3u);
      t /= 3u;
      x = (int)(t % (uint32_t)W);
      t /= (uint32_t)W;
      y = (int)(t % (uint32_t)H);
      b = (int)(t / (uint32_t)H);
    } else {
      ch = (int)(i % 3);
      int64_t t = i / 3;
      x = (int)(t % W);
      t /= W;
      y = (int)(t % H);
      b = (int)(t / H);
    }
    const float istd = 1.0f / (ch == 0 ? s0 : (ch == 1 ? s1 : s2));
    float acc = 0.f;
    const int oyc = (int)(((uint32_t)y * (uint32_t)cut) / (uint32_t)H), oxc = (int)(((uint32_t)x * (uint32_t)cut) / (uint32_t)W);
    // the (up to three) candidate windows per axis; the column candidates once, not once per row candidate
    int cx0[3], cx1[3], cpx[3], ckx[3];
    bool cin[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ox = oxc - 1 + j;
      const bool ok = ox >= 0 && ox < cut;
      cx0[j] = ok ? apool_start(ox, W, cut) : 0;
      cx1[j] = ok ? apool_end(ox, W, cut) : 0;
      cin[j] = ok && x >= cx0[j] && x < cx1[j];
      cpx[j] = ok ? ox / P : 0;
      ckx[j] = ox - cpx[j] * P;
    }
    for (int oy = max(0, oyc - 1); oy <= min(cut - 1, oyc + 1); ++oy) {
      const int y0 = apool_start(oy, H, cut), y1 = apool_end(oy, H, cut);
      if (y < y0 || y >= y1) continue;
      const int py = oy / P, ky = oy - py * P;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (!cin[j]) continue;
        const int x0 = cx0[j], x1 = cx1[j];
        // locate the (first) argmax of this window
        float mx = -INFINITY;
        int ay = y0, ax = x0;
        for (int yy = y0; yy < y1; ++yy)
          for (int xx = x0; xx < x1; ++xx) {
            const float v = xr[(((int64_t)b * H + yy) * W + xx) * 3 + ch];
            if (v > mx) {
              mx = v;
              ay = yy;
              ax = xx;
            }
          }
        float w = 0.5f / (float)((y1 - y0) * (x1 - x0));
        if (ay == y && ax == x) w += 0.5f;
        const int64_t prow = (int64_t)(py * gw + cpx[j]) * (3 * P * P) + (int64_t)ch * P * P + ky * P + ckx[j];
        float g = 0.f;
        for (int c = 0; c < cutn; ++c) g += ElemTraits<GT>::load(gout + ((int64_t)c * B + b) * per_img + prow);
        acc += g * w;
      }
    }
    dxr[i] = acc * istd;
  }
}

// --------------------------- spherical loss --------------------------------
// main.py:801-811: H = normalize(feats[n % B]); E = normalize(embed[n]);
// loss = coef * mean_n 2*asin(|H-E|/2)^2.  One wave per row; writes per-row terms, a second tiny
// kernel sums them deterministically.  The backward kernel emits d loss / d embed.
__global__ __launch_bounds__(256) void sph_loss_rows_kernel(const float* __restrict__ embed, const float* __restrict__ feats,
                                                            float* __restrict__ rowloss, float* __restrict__ dembed,
                                                            int N, int B, int D, float coef) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
    const float* e = embed + (int64_t)n * D;
    const float* h = feats + (int64_t)(n % B) * D;
    float ee = 0.f, hh = 0.f;
    for (int i = lane; i < D; i += 64) {
      ee += e[i] * e[i];
      hh += h[i] * h[i];
    }
    const float en = fmaxf(sqrtf(wave_sum(ee)), 1e-12f), hn = fmaxf(sqrtf(wave_sum(hh)), 1e-12f);
    float dd = 0.f, eg = 0.f;
    for (int i = lane; i < D; i += 64) {
      const float df = h[i] / hn - e[i] / en;
      dd += df * df;
    }
    const float d = sqrtf(wave_sum(dd));
    const float a = asinf(d * 0.5f);
    if (lane == 0) rowloss[n] = 2.0f * a * a;
    if (dembed) {
      // dl/dd = 2 a / sqrt(1 - d^2/4);  dd/dE = -(H-E)/d;  then through E = e/|e|
      const float dl = (d > 0.f) ? (2.0f * a / sqrtf(fmaxf(1.0f - 0.25f * d * d, 1e-20f))) / d : 0.f;
      const float k = coef / (float)N * dl;
      for (int i = lane; i < D; i += 64) {
        const float E = e[i] / en;
        const float gE = -k * (h[i] / hn - E);
        eg += gE * E;
      }
      eg = wave_sum(eg);
      for (int i = lane; i < D; i += 64) {
        const float E = e[i] / en;
        const float gE = -k * (h[i] / hn - E);
        dembed[(int64_t)n * D + i] = (gE - E * eg) / en;
      }
    }
  }
}
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ v, float* __restrict__ out, int n,
                                                        float scale) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += v[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) * scale;
}

// ------------------------------- Adam --------------------------------------
// torch.optim.Adam defaults (main.py:591) over a flat fp32 bucket; optionally refreshes the
// low-precision weight shadow the GEMMs read (same layout) in the same pass.
template <typename ST>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, ST* __restrict__ shadow,
                                                   int64_t n, float lr, float b1, float b2, float eps, float bc1,
                                                   float bc2_sqrt, float gscale, float* __restrict__ ema, float ema_w,
                                                   const float* __restrict__ dev_scale, unsigned int* __restrict__ bad_count,
                                                   const float* __restrict__ dev_hyper) {
  if (dev_hyper) {   // per-step scalars from device memory: a captured hipGraph replays this launch with fresh values
    lr = dev_hyper[0];
    bc1 = dev_hyper[1];
    bc2_sqrt = dev_hyper[2];
    gscale = dev_hyper[3];
    ema_w = dev_hyper[4];
  }
  if (dev_scale) gscale *= dev_scale[0];      // global-norm clip coefficient computed on the device (no host sync)
  // a NaN coefficient = the whole step is skipped (skip_step_on_overflow): the EMA copy keeps its state as well
  if (!(fabsf(gscale) <= 3.0e38f)) ema = nullptr;
  // Non-finite gradients (an overflowing f16 backward, or inf x the clip coefficient 0 = NaN) must never reach p / m / v / ema:
  // such an element keeps its state (no update), and the launch counts the waves that saw one in bad_count, which the host
  // reads at its next logging synchronisation to back the loss scale off (optim.FusedAdam.check_overflow).
  bool bad = false;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4_t pv = load4(p + i * 4), gv = load4(g + i * 4), mv = load4(m + i * 4), vv = load4(v + i * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gg = gv[j] * gscale;
      const bool ok = fabsf(gg) <= 3.0e38f;      // false for inf and NaN
      bad |= !ok;
      const float m2 = b1 * mv[j] + (1.0f - b1) * gg;
      const float v2 = b2 * vv[j] + (1.0f - b2) * gg * gg;
      const float denom = sqrtf(v2) / bc2_sqrt + eps;
      const float p2 = pv[j] - (lr / bc1) * (m2 / denom);
      mv[j] = ok ? m2 : mv[j];
      vv[j] = ok ? v2 : vv[j];
      pv[j] = ok ? p2 : pv[j];
    }
    store4(p + i * 4, pv);
    store4(m + i * 4, mv);
    store4(v + i * 4, vv);
    if (shadow) store4(shadow + i * 4, pv);
    if (ema) {   // torch_ema: shadow -= (1 - decay) * (shadow - param), after the optimizer step (main.py:843-844)
      f32x4_t ev = load4(ema + i * 4);
      ev -= ema_w * (ev - pv);
      store4(ema + i * 4, ev);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const float gg = g[i] * gscale;
    const bool ok = fabsf(gg) <= 3.0e38f;
    bad |= !ok;
    const float mm = ok ? b1 * m[i] + (1.0f - b1) * gg : m[i];
    const float vv = ok ? b2 * v[i] + (1.0f - b2) * gg * gg : v[i];
    const float pp = ok ? p[i] - (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps)) : p[i];
    m[i] = mm;
    v[i] = vv;
    p[i] = pp;
    if (shadow) ElemTraits<ST>::store(shadow + i, pp);
    if (ema) ema[i] -= ema_w * (ema[i] - pp);
  }
  if (bad_count && __ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(bad_count, 1u);
}

// ---- dropout (mlp_mixer_pytorch.py:20-22, vitgan.py:34-41,105,114,133: nn.Dropout inside the mapper MLPs / after attention) ----
// y[i] = (res ? res[i] : 0) + (keep(i) ? x[i] / (1 - p) : 0); keep(i) is a counter-based hash of (seed, i), so the backward
// pass regenerates the identical mask from the seed instead of storing it.
__device__ __forceinline__ bool drop_keep(uint64_t i, uint32_t seed, uint32_t thresh) {
  uint32_t h = (uint32_t)i ^ seed ^ ((uint32_t)(i >> 32) * 0x9E3779B1u);
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h >= thresh;           // P(keep) = 1 - thresh / 2^32
}
template <typename XT, typename YT>
__global__ __launch_bounds__(256) void dropout_kernel(const XT* __restrict__ x, const float* __restrict__ res, YT* __restrict__ y,
                                                      int64_t n, uint32_t seed, uint32_t thresh, float inv_keep) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = drop_keep((uint64_t)i, seed, thresh) ? ElemTraits<XT>::load(x + i) * inv_keep : 0.0f;
    if (res) v += res[i];
    ElemTraits<YT>::store(y + i, v);
  }
}

// ---- optional regularisers of the step (main.py:758-762 l2 = mean(z^2); :423-428,769-773 tv_loss) -------------------
// out[0] += scale * sum x^2  (out zeroed by the launcher)
__global__ __launch_bounds__(256) void mean_sq_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                      float scale) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}
__global__ __launch_bounds__(256) void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        float* __restrict__ y, int64_t n, float c) {
  const float f = g[0] * c;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * f;
}
// x: NHWC fp32 (B, H, W, C).  out[0] += 0.5 * (sum_h |x[h+1]-x[h]| / Nh + sum_w |x[w+1]-x[w]| / Nw)
__global__ __launch_bounds__(256) void tv_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H,
                                                     int W, int C, float inv_nh, float inv_nw) {
  __shared__ float red[4];
  const int64_t n = (int64_t)B * H * W * C, rowc = (int64_t)W * C;
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i / C;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const float v = x[i];
    if (h + 1 < H) a += fabsf(x[i + rowc] - v) * inv_nh;
    if (w + 1 < W) a += fabsf(x[i + C] - v) * inv_nw;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, 0.5f * (red[0] + red[1] + red[2] + red[3]));
}
__device__ __forceinline__ float sgn_(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
__global__ __launch_bounds__(256) void tv_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                     float* __restrict__ dx, int B, int H, int W, int C, float inv_nh,
                                                     float inv_nw) {
  const int64_t n = (int64_t)B * H * W * C, rowc = (int64_t)W * C;
  const float gg = 0.5f * g[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t pix = i / C;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const float v = x[i];
    float d = 0.f;
    if (h > 0) d += sgn_(v - x[i - rowc]) * inv_nh;
    if (h + 1 < H) d -= sgn_(x[i + rowc] - v) * inv_nh;
    if (w > 0) d += sgn_(v - x[i - C]) * inv_nw;
    if (w + 1 < W) d -= sgn_(x[i + C] - v) * inv_nw;
    dx[i] = gg * d;
  }
}

// sum of squares of a flat buffer -> out[0] (+=), for global-norm gradient clipping (main.py:833-834)
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += x[i] * x[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// y = a*x + b*y elementwise over fp32 (gradient accumulation / scaling plumbing)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float a,
                                                    float b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

// out[r % period] += sum_c x[r, c]; one wave per row
template <typename T>
__global__ __launch_bounds__(256) void rowsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                     int cols, int period) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float a = 0.f;
    if ((cols & 3) == 0) {
      for (int c = lane * 4; c < cols; c += 256) {
        const f32x4_t v = load4(x + r * cols + c);
        a += (v[0] + v[1]) + (v[2] + v[3]);
      }
    } else {
      for (int c = lane; c < cols; c += 64) a += ElemTraits<T>::load(x + r * cols + c);
    }
    a = wave_sum(a);
    if (lane == 0) atomicAdd(out + (r % period), a);
  }
}

// The same sums with one wave per (output index o, chunk of the rows o, o + period, ... that share it): the wave keeps its sum
// in registers across its rows (16-byte loads, two rows in flight) and issues ONE atomic — the row-per-wave form above issues
// rows / period same-address atomics per output (256 at the token-mixing bias gradients) and 8-byte loads.
template <typename T>
__global__ __launch_bounds__(256) void rowsum_grouped_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                             int cols, int period, int chunks, int rows_per_chunk) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int o = (int)(wv % period), ch = (int)(wv / period);
  if (ch >= chunks) return;
  const int64_t per = rows / period;                 // rows that share one output
  const int64_t b0 = (int64_t)ch * rows_per_chunk, b1 = min(per, b0 + rows_per_chunk);
  float a = 0.f;
  for (int64_t b = b0; b < b1; ++b) {
    const T* px = x + (b * period + o) * (int64_t)cols;
    for (int c = lane * 8; c < cols; c += 512) {
      const f32x8 v = load8(px + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v.v[j];
    }
  }
  a = wave_sum(a);
  if (lane == 0) atomicAdd(out + o, a);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ s, int64_t ss, float* __restrict__ d,
                                                        int64_t ds, int64_t rows, int cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    d[r * ds + c] = s[r * ss + c];
  }
}

template <typename XT, typename OT>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const XT* __restrict__ x, OT* __restrict__ out, int B, int H,
                                                        int W, int C, int Kp) {
  const int64_t n = (int64_t)B * H * W * Kp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    int64_t p = i / Kp;
    const int xx = (int)(p % W);
    p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    float v = 0.f;
    if (k < 9 * C) {
      const int tap = k / C, c = k - tap * C;
      const int iy = yy + tap / 3 - 1, ix = xx + tap % 3 - 1;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        v = ElemTraits<XT>::load(x + (((int64_t)b * H + iy) * W + ix) * C + c);
    }
    ElemTraits<OT>::store(out + i, v);
  }
}

// y = x * s[0] with the scalar read from device memory (backward of the loss w.r.t. an upstream scalar grad)
__global__ __launch_bounds__(256) void mul_dev_scalar_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                             float* __restrict__ y, int64_t n) {
  const float k = s[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = x[i] * k;
}

// dst[r, c] = c < cols ? src[r, c] : 0 for c < dst_cols, independent leading dims, dtype conversion
// (pad / unpad of per-head blocks, e.g. VitGAN's dim_head = 170 -> 176)
template <typename ST, typename DT>
__global__ __launch_bounds__(256) void copy2d_kernel(const ST* __restrict__ s, int64_t sld, DT* __restrict__ d, int64_t dld,
                                                     int64_t rows, int cols, int dst_cols) {
  const int64_t n = rows * dst_cols;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / dst_cols;
    const int c = (int)(i - r * dst_cols);
    ElemTraits<DT>::store(d + r * dld + c, c < cols ? ElemTraits<ST>::load(s + r * sld + c) : 0.f);
  }
}

// ------------------------------ augmented cutouts ----------------------------
// The reference's default MakeCutouts pipeline (main.py:164-165,170-198,222-225,797): RandomAffine(15 deg, translate
// .1, border) -> RandomPerspective(.7, zeros) -> ColorJitter(hue .1, saturation .1) -> RandomErasing -> + noise ->
// (x - mean) / std, fused into ONE resampling pass over the pooled image and written as ViT patch rows.
// Every random draw arrives as an explicit per-cutout parameter:
//   pinv [N,9]: output pixel -> coordinates in the affine-warped image (inverse perspective homography)
//   ainv [N,6]: affine-warped coordinates -> pooled-image coordinates (inverse affine)
//   cmat [N,9]: RGB colour matrix (hue rotation + saturation scale, exact for identity)
//   erase[N,4]: x0, y0, x1, y1 of the erased rectangle (x1 <= x0: none)
// value = inside(p1) * bilinear(pooled, clamp(A^-1 p1)), p1 = P^-1 p2 — one interpolation instead of kornia's two
// (kornia 0.5.10 is not restated: parity unpinned, statistically equivalent augmentation).
struct AugSample {
  int x0, y0;
  float wx, wy, m;
};
// S = side of the SOURCE image (the frame of the affine slot); the output grid may have another size (a resize / crop
// in the chain is part of pinv).
__device__ __forceinline__ AugSample aug_coords(const float* __restrict__ pinv, const float* __restrict__ ainv, int ox,
                                                int oy, int S) {
  const float x2 = (float)ox, y2 = (float)oy;
  const float w = pinv[6] * x2 + pinv[7] * y2 + pinv[8];
  const float iw = 1.0f / (fabsf(w) > 1e-8f ? w : 1e-8f);
  const float x1 = (pinv[0] * x2 + pinv[1] * y2 + pinv[2]) * iw;
  const float y1 = (pinv[3] * x2 + pinv[4] * y2 + pinv[5]) * iw;
  AugSample a;
  // zeros padding of the homography slot as grid_sample(padding_mode='zeros', align_corners=False) applies it (kornia's
  // RandomPerspective / RandomRotation): each bilinear tap outside the frame counts as zero, i.e. the image fades out linearly
  // over the pixel around its border: weight (t + 1) on [-1, 0], (S - t) on [S - 1, S], per axis
  const float rx = fminf(fmaxf(x1 + 1.0f, 0.0f), 1.0f) * fminf(fmaxf((float)S - x1, 0.0f), 1.0f);
  const float ry = fminf(fmaxf(y1 + 1.0f, 0.0f), 1.0f) * fminf(fmaxf((float)S - y1, 0.0f), 1.0f);
  // a homography slot that only scales / shifts (resize, crop, identity) has no zero-padded warp in it: coordinates are clamped
  const bool zero_pad = pinv[1] != 0.0f || pinv[3] != 0.0f || pinv[6] != 0.0f || pinv[7] != 0.0f;
  a.m = zero_pad ? rx * ry : 1.0f;
  float x0 = ainv[0] * x1 + ainv[1] * y1 + ainv[2];
  float y0 = ainv[3] * x1 + ainv[4] * y1 + ainv[5];
  x0 = fminf(fmaxf(x0, 0.0f), (float)(S - 1));                                                            // border padding
  y0 = fminf(fmaxf(y0, 0.0f), (float)(S - 1));
  a.x0 = min((int)x0, S - 2 < 0 ? 0 : S - 2);
  a.y0 = min((int)y0, S - 2 < 0 ? 0 : S - 2);
  a.wx = x0 - (float)a.x0;
  a.wy = y0 - (float)a.y0;
  return a;
}

// SEQUENTIAL form (round 5; kornia's nn.Sequential, main.py:199,219: RandomAffine and RandomPerspective are TWO bilinear resamples,
// the second one reading the first one's output): the value at an output pixel is the homography slot's interpolation (zeros /
// fade, exactly as above with an identity affine) of the INTERMEDIATE image I, and every one of its (up to) four taps I(p), p an
// integer pixel, is itself the border-padded affine interpolation of the source: I(p) = bilinear(src, clamp(A^-1 p)).  Evaluated
// lazily — 16 source taps per output pixel — the intermediate image never exists in memory, and the result is what the two-launch
// form (one launch per warp) produces up to fp32 summation order.
struct AugTaps {
  AugSample s[4];      // source-image samples of the (up to) four intermediate pixels
  float w[4];          // their weights (homography-slot bilinear weight x fade); 0 = unused
};
template <bool SEQ>
__device__ __forceinline__ void aug_taps(const float* __restrict__ pinv, const float* __restrict__ ainv, int ox, int oy, int S,
                                         AugTaps& t) {
  if constexpr (!SEQ) {
    t.s[0] = aug_coords(pinv, ainv, ox, oy, S);
    t.w[0] = t.s[0].m;
    t.s[0].m = 1.0f;
    t.w[1] = t.w[2] = t.w[3] = 0.0f;
    t.s[1] = t.s[2] = t.s[3] = t.s[0];
  } else {
    const float ident[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    const AugSample q = aug_coords(pinv, ident, ox, oy, S);          // position in the intermediate image (side S as well)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int px = q.x0 + (k & 1), py = q.y0 + (k >> 1);
      t.w[k] = q.m * ((k & 1) ? q.wx : 1.0f - q.wx) * ((k >> 1) ? q.wy : 1.0f - q.wy);
      float x0 = ainv[0] * (float)px + ainv[1] * (float)py + ainv[2];
      float y0 = ainv[3] * (float)px + ainv[4] * (float)py + ainv[5];
      x0 = fminf(fmaxf(x0, 0.0f), (float)(S - 1));                   // border padding of the affine warp
      y0 = fminf(fmaxf(y0, 0.0f), (float)(S - 1));
      AugSample a;
      a.x0 = min((int)x0, S - 2 < 0 ? 0 : S - 2);
      a.y0 = min((int)y0, S - 2 < 0 ? 0 : S - 2);
      a.wx = x0 - (float)a.x0;
      a.wy = y0 - (float)a.y0;
      a.m = 1.0f;
      t.s[k] = a;
    }
  }
}
// value of the resampled (pre colour) image at one output pixel, all three channels
template <bool SEQ>
__device__ __forceinline__ void aug_gather(const float* __restrict__ img3, int Ss, const AugTaps& t, float (&rgb)[3]) {
  rgb[0] = rgb[1] = rgb[2] = 0.0f;
#pragma unroll
  for (int k = 0; k < (SEQ ? 4 : 1); ++k) {
    if (t.w[k] == 0.0f) continue;
    const AugSample& a = t.s[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* src = img3 + (int64_t)c * Ss * Ss + (int64_t)a.y0 * Ss + a.x0;
      const float v00 = src[0], v01 = src[1], v10 = src[Ss], v11 = src[Ss + 1];
      rgb[c] += t.w[k] * ((1.f - a.wy) * ((1.f - a.wx) * v00 + a.wx * v01) + a.wy * ((1.f - a.wx) * v10 + a.wx * v11));
    }
  }
}

template <typename OT, bool SEQ = false>
__global__ __launch_bounds__(256) void augment_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ pinv,
                                                          const float* __restrict__ ainv, const float* __restrict__ cmat,
                                                          const int* __restrict__ erase, const float* __restrict__ noise,
                                                          const float* __restrict__ facs, const float* __restrict__ coff,
                                                          const float* __restrict__ cj, OT* __restrict__ out, int B, int S,
                                                          int Ss, int cutn, int P, float m0, float m1, float m2, float s0,
                                                          float s1, float s2) {
  const int gw = S / P;
  const int64_t per_img = (int64_t)3 * S * S;
  const float mean[3] = {m0, m1, m2}, istd[3] = {1.0f / s0, 1.0f / s1, 1.0f / s2};
  // one cutout per blockIdx.y: its 39 parameters (two warps, colour matrix + offset, jitter, erase box) are wave-uniform, so they
  // travel in scalar registers and the jitter's operator order is a scalar branch (r6: 1.19 -> 0.86 ms isolated at cfg2's 512 cutouts)
  const int n = blockIdx.y;
  const int b = n % B;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < S * S; i += gridDim.x * 256) {
    const int oy = i / S;
    const int ox = i - oy * S;
    AugTaps taps;
    aug_taps<SEQ>(pinv + n * 9, ainv + n * 6, ox, oy, Ss, taps);
    const int* er = erase + n * 4;
    const bool erased = ox >= er[0] && ox < er[2] && oy >= er[1] && oy < er[3];
    float rgb[3];
#if defined(FFVC_AUGF_EXP) && FFVC_AUGF_EXP == 6      // timing experiment (wrong results): no source gathers
    rgb[0] = taps.w[0] + taps.s[0].wx; rgb[1] = taps.w[1] + taps.s[1].wy; rgb[2] = taps.w[2] + taps.w[3] + taps.s[3].wx;
#else
    aug_gather<SEQ>(pooled + (int64_t)b * 3 * Ss * Ss, Ss, taps, rgb);
#endif
    const float* cm = cmat + n * 9;
    const int py = oy / P, ky = oy - py * P, px = ox / P, kx = ox - px * P;
    float col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) col[c] = cm[c * 3] * rgb[0] + cm[c * 3 + 1] * rgb[1] + cm[c * 3 + 2] * rgb[2] + (coff ? coff[n * 3 + c] : 0.0f);
#if defined(FFVC_AUGF_EXP) && FFVC_AUGF_EXP == 5      // timing experiment (wrong results): no colour jitter
    if (cj && cj[n * 8] == 12345.0f && !erased) {
#else
    if (cj && cj[n * 8] != 0.0f && !erased) {      // kornia ColorJitter (hsv round trips, clamps, random order): augment_cj.h
#endif
      float J[3][3], o[3];
      ffvc_cj::color_jitter(cj + n * 8, col, o, J);
      col[0] = o[0];
      col[1] = o[1];
      col[2] = o[2];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = erased ? 0.0f : col[c];
#if !(defined(FFVC_AUGF_EXP) && FFVC_AUGF_EXP == 7)   // 7: timing experiment (wrong results): no noise read
      if (noise) v += facs[n] * noise[(int64_t)n * per_img + ((int64_t)c * S + oy) * S + ox];
#endif
      const int64_t prow = (int64_t)(py * gw + px) * (3 * P * P) + (int64_t)c * P * P + ky * P + kx;
#if defined(FFVC_AUGF_EXP) && FFVC_AUGF_EXP == 8      // timing experiment (wrong results): no store
      if (v == 12345.678f)
#endif
      ElemTraits<OT>::store(out + (int64_t)n * per_img + prow, (v - mean[c]) * istd[c]);
    }
  }
}

// dpooled (pre-zeroed) += scatter of the bilinear taps (fp32 atomics: ~4 per output pixel and channel)
template <typename GT, bool SEQ = false>
__global__ __launch_bounds__(256) void augment_bwd_kernel(const GT* __restrict__ gout, const float* __restrict__ pinv,
                                                          const float* __restrict__ ainv, const float* __restrict__ cmat,
                                                          const int* __restrict__ erase, const float* __restrict__ pooled,
                                                          const float* __restrict__ coff, const float* __restrict__ cj,
                                                          float* __restrict__ dpooled, int B,
                                                          int S, int Ss, int cutn, int P, float s0, float s1, float s2) {
  const int gw = S / P;
  const int64_t n_px = (int64_t)cutn * B * S * S;
  const int64_t per_img = (int64_t)3 * S * S;
  const float istd[3] = {1.0f / s0, 1.0f / s1, 1.0f / s2};
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_px; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % S);
    int64_t t = i / S;
    const int oy = (int)(t % S);
    const int n = (int)(t / S);
    const int b = n % B;
    const int* er = erase + n * 4;
    if (ox >= er[0] && ox < er[2] && oy >= er[1] && oy < er[3]) continue;
    AugTaps taps;
    aug_taps<SEQ>(pinv + n * 9, ainv + n * 6, ox, oy, Ss, taps);
    if (taps.w[0] == 0.0f && taps.w[1] == 0.0f && taps.w[2] == 0.0f && taps.w[3] == 0.0f) continue;
    const int py = oy / P, ky = oy - py * P, px = ox / P, kx = ox - px * P;
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t prow = (int64_t)(py * gw + px) * (3 * P * P) + (int64_t)c * P * P + ky * P + kx;
      g[c] = ElemTraits<GT>::load(gout + (int64_t)n * per_img + prow) * istd[c];
    }
    const float* cm = cmat + n * 9;
    if (cj && cj[n * 8] != 0.0f) {
      // the jitter is not linear: recompute this pixel's forward value up to the jitter's input, then g <- J^T g
      float rgb[3], col[3];
      aug_gather<SEQ>(pooled + (int64_t)b * 3 * Ss * Ss, Ss, taps, rgb);
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = cm[c * 3] * rgb[0] + cm[c * 3 + 1] * rgb[1] + cm[c * 3 + 2] * rgb[2] + (coff ? coff[n * 3 + c] : 0.0f);
      float J[3][3], o[3];
      ffvc_cj::color_jitter(cj + n * 8, col, o, J);
      const float g0 = g[0], g1 = g[1], g2 = g[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = J[0][c] * g0 + J[1][c] * g1 + J[2][c] * g2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float gc = cm[c] * g[0] + cm[3 + c] * g[1] + cm[6 + c] * g[2];          // transpose of the colour matrix
#pragma unroll
      for (int k = 0; k < (SEQ ? 4 : 1); ++k) {
        if (taps.w[k] == 0.0f) continue;
        const AugSample& a = taps.s[k];
        const float gk = gc * taps.w[k];
        float* dst = dpooled + ((int64_t)b * 3 + c) * Ss * Ss + (int64_t)a.y0 * Ss + a.x0;
        atomicAdd(dst, gk * (1.f - a.wy) * (1.f - a.wx));
        atomicAdd(dst + 1, gk * (1.f - a.wy) * a.wx);
        atomicAdd(dst + Ss, gk * a.wy * (1.f - a.wx));
        atomicAdd(dst + Ss + 1, gk * a.wy * a.wx);
      }
    }
  }
}

// Tiled form of augment_bwd_kernel (round 4): one workgroup = one 16 x 16 output tile of one cutout.  The four bilinear taps of
// neighbouring output pixels land on the same source pixels, so the tile first accumulates into an LDS image of its source
// footprint (bounding box of the taps, ds_add_f32) and then issues ONE global atomic per touched source pixel and channel instead
// of four per output pixel and channel (~3x fewer L2 atomics; 308 M per step at cfg2 before).  Tiles whose footprint does not
// fit the LDS image (strong perspective) fall back to direct global atomics.
// Tile shape: 16 x 16 output pixels.  (32 x 8 — a half-wave per output row, consecutive LDS words — measured 1 % faster at cfg2, r6:
// the 48 accumulations per thread are bound by their VALU / exec-mask work, not by bank conflicts; -DFFVC_AUGT_X=32 rebuilds it)
#ifndef FFVC_AUGT_X
#define FFVC_AUGT_X 16
#endif
constexpr int AUGTX = FFVC_AUGT_X, AUGTY = 256 / AUGTX, AUG_CAP = 2048;      // source pixels of the LDS image (x 3 channels x 4 B = 24 KiB)
template <typename GT, bool SEQ = false>
__global__ __launch_bounds__(256) void augment_bwd_tiled_kernel(const GT* __restrict__ gout, const float* __restrict__ pinv,
                                                                const float* __restrict__ ainv, const float* __restrict__ cmat,
                                                                const int* __restrict__ erase, const float* __restrict__ pooled,
                                                                const float* __restrict__ coff, const float* __restrict__ cj,
                                                                float* __restrict__ dpooled, int B, int S, int Ss, int P,
                                                                float s0, float s1, float s2) {
  __shared__ float img[3 * AUG_CAP];
  __shared__ int bb[4];
  constexpr int NK = SEQ ? 4 : 1;
  const int n = blockIdx.y, b = n % B;
  const int tiles_x = (S + AUGTX - 1) / AUGTX;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int ox = tx * AUGTX + (threadIdx.x & (AUGTX - 1)), oy = ty * AUGTY + (threadIdx.x / AUGTX);
  const int gw = S / P;
  const int64_t per_img = (int64_t)3 * S * S;
  const float istd[3] = {1.0f / s0, 1.0f / s1, 1.0f / s2};
  if (threadIdx.x == 0) {
    bb[0] = bb[1] = 1 << 30;
    bb[2] = bb[3] = -1;
  }
  for (int i = threadIdx.x; i < 3 * AUG_CAP; i += 256) img[i] = 0.0f;
  __syncthreads();
  bool live = ox < S && oy < S;
  AugTaps taps;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    taps.w[k] = 0.0f;
    taps.s[k].x0 = taps.s[k].y0 = 0;
    taps.s[k].wx = taps.s[k].wy = taps.s[k].m = 0.0f;
  }
  float gc[3] = {0.f, 0.f, 0.f};
  if (live) {
    const int* er = erase + n * 4;
    if (ox >= er[0] && ox < er[2] && oy >= er[1] && oy < er[3]) live = false;
  }
  if (live) {
    aug_taps<SEQ>(pinv + n * 9, ainv + n * 6, ox, oy, Ss, taps);
    if (taps.w[0] == 0.0f && taps.w[1] == 0.0f && taps.w[2] == 0.0f && taps.w[3] == 0.0f) live = false;
  }
  if (live) {
    const int py = oy / P, ky = oy - py * P, px = ox / P, kx = ox - px * P;
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int64_t prow = (int64_t)(py * gw + px) * (3 * P * P) + (int64_t)c * P * P + ky * P + kx;
      g[c] = ElemTraits<GT>::load(gout + (int64_t)n * per_img + prow) * istd[c];
    }
    const float* cm = cmat + n * 9;
#if defined(FFVC_AUGB_EXP) && FFVC_AUGB_EXP == 3      // timing experiment (wrong results): no colour-jitter recompute
    if (cj && cj[n * 8] == 12345.0f) {
#else
    if (cj && cj[n * 8] != 0.0f) {
#endif
      float rgb[3], col[3];
      aug_gather<SEQ>(pooled + (int64_t)b * 3 * Ss * Ss, Ss, taps, rgb);
#pragma unroll
      for (int c = 0; c < 3; ++c) col[c] = cm[c * 3] * rgb[0] + cm[c * 3 + 1] * rgb[1] + cm[c * 3 + 2] * rgb[2] + (coff ? coff[n * 3 + c] : 0.0f);
      float J[3][3], o[3];
      ffvc_cj::color_jitter(cj + n * 8, col, o, J);
      const float g0 = g[0], g1 = g[1], g2 = g[2];
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = J[0][c] * g0 + J[1][c] * g1 + J[2][c] * g2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gc[c] = cm[c] * g[0] + cm[3 + c] * g[1] + cm[6 + c] * g[2];          // transpose of the colour matrix
  }
  {
    // bounding box of the tile's source footprint: reduced inside each wave first, ONE LDS atomic per wave and bound.  (Per-lane
    // atomics on the four shared words serialise 64 lanes x 4 taps each: 0.6 of the kernel's 1.9 ms at cfg2, r6)
    int mnx = 1 << 30, mny = 1 << 30, mxx = -1, mxy = -1;
    if (live) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (taps.w[k] == 0.0f) continue;
        mnx = min(mnx, taps.s[k].x0);
        mny = min(mny, taps.s[k].y0);
        mxx = max(mxx, taps.s[k].x0 + 1);
        mxy = max(mxy, taps.s[k].y0 + 1);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, o));
      mny = min(mny, __shfl_xor(mny, o));
      mxx = max(mxx, __shfl_xor(mxx, o));
      mxy = max(mxy, __shfl_xor(mxy, o));
    }
    if ((threadIdx.x & 63) == 0 && mxx >= 0) {
      atomicMin(&bb[0], mnx);
      atomicMin(&bb[1], mny);
      atomicMax(&bb[2], mxx);
      atomicMax(&bb[3], mxy);
    }
  }
  __syncthreads();
  const int bx0 = bb[0], by0 = bb[1], bw = bb[2] - bb[0] + 1, bh = bb[3] - bb[1] + 1;
  if (bb[2] < 0) return;                                   // nothing live in this tile
  float* dimg = dpooled + (int64_t)b * 3 * Ss * Ss;
  if (bw * bh <= AUG_CAP) {
#if defined(FFVC_AUGB_EXP) && FFVC_AUGB_EXP == 4      // timing experiment (wrong results): no LDS accumulation
    if (live && gc[0] == 12345.0f) {
#else
    if (live) {
#endif
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        if (taps.w[k] == 0.0f) continue;
        const AugSample& a = taps.s[k];
        const int o = (a.y0 - by0) * bw + (a.x0 - bx0);
        const float w00 = taps.w[k] * (1.f - a.wy) * (1.f - a.wx), w01 = taps.w[k] * (1.f - a.wy) * a.wx,
                    w10 = taps.w[k] * a.wy * (1.f - a.wx), w11 = taps.w[k] * a.wy * a.wx;
        // (an operator that is off for this cutout is an identity map: its interpolation has ONE non-zero corner — skip the
        // other three LDS atomics; with p = 0.7 per operator that is half of all cutouts for one of the two warps)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float* t = img + c * AUG_CAP + o;
          if (w00 != 0.0f) atomicAdd(t, gc[c] * w00);
          if (w01 != 0.0f) atomicAdd(t + 1, gc[c] * w01);
          if (w10 != 0.0f) atomicAdd(t + bw, gc[c] * w10);
          if (w11 != 0.0f) atomicAdd(t + bw + 1, gc[c] * w11);
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bw * bh; i += 256) {
      const int yy = i / bw, xx = i - yy * bw;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = img[c * AUG_CAP + i];
#if defined(FFVC_AUGB_EXP) && FFVC_AUGB_EXP == 1      // timing experiment (wrong results): plain stores instead of the global atomics
        if (v != 0.0f) dimg[(int64_t)c * Ss * Ss + (int64_t)(by0 + yy) * Ss + bx0 + xx] = v;
#elif defined(FFVC_AUGB_EXP) && FFVC_AUGB_EXP == 2    // timing experiment (wrong results): no flush at all
        if (v == 12345.678f) dimg[(int64_t)c * Ss * Ss + (int64_t)(by0 + yy) * Ss + bx0 + xx] = v;
#else
        if (v != 0.0f) atomicAdd(dimg + (int64_t)c * Ss * Ss + (int64_t)(by0 + yy) * Ss + bx0 + xx, v);
#endif
      }
    }
  } else if (live) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (taps.w[k] == 0.0f) continue;
      const AugSample& a = taps.s[k];
      const float w00 = taps.w[k] * (1.f - a.wy) * (1.f - a.wx), w01 = taps.w[k] * (1.f - a.wy) * a.wx,
                  w10 = taps.w[k] * a.wy * (1.f - a.wx), w11 = taps.w[k] * a.wy * a.wx;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float* dst = dimg + (int64_t)c * Ss * Ss + (int64_t)a.y0 * Ss + a.x0;
        atomicAdd(dst, gc[c] * w00);
        atomicAdd(dst + 1, gc[c] * w01);
        atomicAdd(dst + Ss, gc[c] * w10);
        atomicAdd(dst + Ss + 1, gc[c] * w11);
      }
    }
  }
}

// MakeCutouts(interpolate=True) (main.py:226-228): adaptive average pooling of the augmented batch x [N,3,S,S] fp32 to So x So,
// then mean/std and the ViT patch layout.  The backward spreads g/std/|window| over each window.
template <typename OT>
__global__ __launch_bounds__(256) void avgpool_patches_fwd_kernel(const float* __restrict__ x, OT* __restrict__ out, int N, int S,
                                                                  int So, int P, float m0, float m1, float m2, float s0,
                                                                  float s1, float s2) {
  const int gw = So / P;
  const int64_t n = (int64_t)N * 3 * So * So;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % So);
    int64_t t = i / So;
    const int oy = (int)(t % So);
    t /= So;
    const int ch = (int)(t % 3);
    const int img = (int)(t / 3);
    const int y0 = apool_start(oy, S, So), y1 = apool_end(oy, S, So);
    const int x0 = apool_start(ox, S, So), x1 = apool_end(ox, S, So);
    const float* src = x + ((int64_t)img * 3 + ch) * S * S;
    float sum = 0.f;
    for (int y = y0; y < y1; ++y)
      for (int xx = x0; xx < x1; ++xx) sum += src[(int64_t)y * S + xx];
    const float v = sum / (float)((y1 - y0) * (x1 - x0));
    const float mean = ch == 0 ? m0 : (ch == 1 ? m1 : m2);
    const float istd = 1.0f / (ch == 0 ? s0 : (ch == 1 ? s1 : s2));
    const int py = oy / P, ky = oy - py * P, px = ox / P, kx = ox - px * P;
    const int64_t prow = (int64_t)(py * gw + px) * (3 * P * P) + (int64_t)ch * P * P + ky * P + kx;
    ElemTraits<OT>::store(out + (int64_t)img * 3 * So * So + prow, (v - mean) * istd);
  }
}

template <typename GT>
__global__ __launch_bounds__(256) void avgpool_patches_bwd_kernel(const GT* __restrict__ gout, float* __restrict__ dx, int N, int S,
                                                                  int So, int P, float s0, float s1, float s2) {
  const int gw = So / P;
  const int64_t n = (int64_t)N * 3 * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % S);
    int64_t t = i / S;
    const int y = (int)(t % S);
    t /= S;
    const int ch = (int)(t % 3);
    const int img = (int)(t / 3);
    const float istd = 1.0f / (ch == 0 ? s0 : (ch == 1 ? s1 : s2));
    // output cells whose window [start(o), end(o)) holds this pixel: a contiguous range around floor(y*So/S)
    float acc = 0.f;
    const int oyc = (int)(((int64_t)y * So) / S), oxc = (int)(((int64_t)x * So) / S);
    const int r = So > S ? (So + S - 1) / S + 1 : 1;
    for (int oy = max(0, oyc - r); oy <= min(So - 1, oyc + r); ++oy) {
      const int y0 = apool_start(oy, S, So), y1 = apool_end(oy, S, So);
      if (y < y0 || y >= y1) continue;
      for (int ox = max(0, oxc - r); ox <= min(So - 1, oxc + r); ++ox) {
        const int x0 = apool_start(ox, S, So), x1 = apool_end(ox, S, So);
        if (x < x0 || x >= x1) continue;
        const int py = oy / P, ky = oy - py * P, px = ox / P, kx = ox - px * P;
        const int64_t prow = (int64_t)(py * gw + px) * (3 * P * P) + (int64_t)ch * P * P + ky * P + kx;
        acc += ElemTraits<GT>::load(gout + (int64_t)img * 3 * So * So + prow) / (float)((y1 - y0) * (x1 - x0));
      }
    }
    dx[i] = acc * istd;
  }
}

}  // namespace

extern "C" int ffvc_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  FFVC_CHECK_ARG(src && dst && n > 0, "ffvc_cast: bad args");
  FFVC_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "ffvc_cast: pointers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(src_dtype, ST, DISPATCH_DT(dst_dtype, DT,
              hipLaunchKernelGGL((cast_kernel<ST, DT>), dim3(ew_grid(n / 4 + 1, 256)), dim3(256), 0, st,
                                 (const ST*)src, (DT*)dst, n)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_split3(const float* src, void* dst, int dst_dtype, int64_t rows, int K, int64_t ld_src, int weight_order,
                           void* stream) {
  FFVC_CHECK_ARG(src && dst && rows > 0 && K > 0 && (K % 4) == 0 && ld_src >= K && (ld_src % 4) == 0,
                 "ffvc_split3: bad args (K and ld_src must be multiples of 4)");
  FFVC_CHECK_ARG(dst_dtype == FFVC_F16 || dst_dtype == FFVC_BF16, "ffvc_split3: dst must be a 16-bit dtype");
  FFVC_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 8) == 0, "ffvc_split3: misaligned pointers");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n4 = rows * (K / 4);
  if (dst_dtype == FFVC_F16)
    hipLaunchKernelGGL((split3_kernel<f16_t>), dim3(ew_grid(n4, 256)), dim3(256), 0, st, src, (f16_t*)dst, rows, K, ld_src, weight_order);
  else
    hipLaunchKernelGGL((split3_kernel<uint16_t>), dim3(ew_grid(n4, 256)), dim3(256), 0, st, src, (uint16_t*)dst, rows, K, ld_src, weight_order);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_transpose(const void* src, int src_dtype, void* dst, int dst_dtype, int batch, int rows, int cols,
                              int64_t src_batch_stride, int64_t dst_batch_stride, int dst_ld, void* stream) {
  FFVC_CHECK_ARG(src && dst && batch > 0 && rows > 0 && cols > 0, "ffvc_transpose: bad args");
  if (dst_ld <= 0) dst_ld = rows;
  FFVC_CHECK_ARG(dst_ld >= rows, "ffvc_transpose: dst_ld < rows");
  hipStream_t st = (hipStream_t)stream;
  if (batch > 65535) {   // split huge batches (gridDim.z limit)
    for (int b0 = 0; b0 < batch; b0 += 65535) {
      const int nb = batch - b0 < 65535 ? batch - b0 : 65535;
      const size_t ss = ffvc_dtype_size(src_dtype), ds = ffvc_dtype_size(dst_dtype);
      int e = ffvc_transpose((const char*)src + (size_t)b0 * src_batch_stride * ss, src_dtype,
                             (char*)dst + (size_t)b0 * dst_batch_stride * ds, dst_dtype, nb, rows, cols, src_batch_stride,
                             dst_batch_stride, dst_ld, stream);
      if (e) return e;
    }
    return 0;
  }
  dim3 grid(ceil_div(cols, 64), ceil_div(dst_ld, 64), batch);
  FFVC_CHECK_ARG(grid.y <= 65535, "ffvc_transpose: too many rows");
  DISPATCH_DT(src_dtype, ST, DISPATCH_DT(dst_dtype, DT,
              hipLaunchKernelGGL((transpose_kernel<ST, DT>), grid, dim3(256), 0, st, (const ST*)src, (DT*)dst, rows,
                                 cols, src_batch_stride, dst_batch_stride, dst_ld)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_transpose_multi(const ffvc_tr_item* items, const int* tile_prefix, int n_items, int total_tiles,
                                    int dtype, void* stream) {
  FFVC_CHECK_ARG(items && tile_prefix && n_items > 0 && total_tiles > 0, "ffvc_transpose_multi: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype != FFVC_F32)     // a transpose only moves bits: both 16-bit formats share the instantiation
    hipLaunchKernelGGL((transpose_multi_kernel<uint16_t>), dim3(total_tiles), dim3(256), 0, st, items, tile_prefix, n_items);
  else
    hipLaunchKernelGGL((transpose_multi_kernel<float>), dim3(total_tiles), dim3(256), 0, st, items, tile_prefix, n_items);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_colsum(const void* x, int dtype, float* out, int64_t rows, int cols, int64_t ld, int accumulate,
                           void* stream) {
  FFVC_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "ffvc_colsum: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)cols * sizeof(float), st);
    if (e != hipSuccess) {
      ffvc_set_error("ffvc_colsum: memset failed: %s", hipGetErrorString(e));
      return (int)e;
    }
  }
  const int es = ffvc_dtype_size(dtype);
  const int vec = (cols % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)x) % (4 * es) == 0);
  // row strips: column-blocks x strips workgroups.  Every strip ends in one fp32 atomic per column, and atomics on one address serialise
  // (~75 ns): 1024 workgroups over 256 columns = 1024 atomics per column = 77 us for an 8 MB tensor (the x-transformer's bias gradients,
  // rocprofv3 r4: 98 us per call).  With eight loads in flight per lane ~256 workgroups are enough (FFVC_COLSUM_WGS overrides).
  static int target = -1;
  if (target < 0) {
    const char* e = getenv("FFVC_COLSUM_WGS");
    target = e ? atoi(e) : 256;
    if (target < 1) target = 1;
  }
  const int colblocks = ceil_div(cols, vec ? 256 : 64);
  int64_t strips = (target + colblocks - 1) / colblocks;
  if (strips > 128) strips = 128;                  // <= 128 atomics per column (16384 x 256 f16: 102 us at 1024 strips, 29 at 256, 17 at 128)
  if (strips > rows / 8) strips = rows / 8;
  if (strips < 1) strips = 1;
  const int rpb = (int)((rows + strips - 1) / strips);
  dim3 grid(colblocks, ceil_div(rows, rpb));
  DISPATCH_DT(dtype, T, hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, st, (const T*)x, out, rows, cols,
                                           ld, rpb, vec));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_clamp_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, float mul, float add, float lo,
                              float hi, void* stream) {
  FFVC_CHECK_ARG(x && y && n > 0, "ffvc_clamp_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(x_dtype, XT, DISPATCH_DT(y_dtype, YT,
              hipLaunchKernelGGL((clamp_fwd_kernel<XT, YT>), dim3(ew_grid(n, 1024)), dim3(256), 0, st, (const XT*)x,
                                 (YT*)y, n, mul, add, lo, hi)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_clamp_bwd(const void* x, int x_dtype, const void* g, int g_dtype, void* dx, int64_t n, float mul,
                              float add, float lo, float hi, void* stream) {
  FFVC_CHECK_ARG(x && g && dx && n > 0, "ffvc_clamp_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(x_dtype, XT, DISPATCH_DT(g_dtype, GT,
              hipLaunchKernelGGL((clamp_bwd_kernel<XT, GT>), dim3(ew_grid(n, 1024)), dim3(256), 0, st, (const XT*)x,
                                 (const GT*)g, (XT*)dx, n, mul, add, lo, hi)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_sumpool2x2(const void* src, void* dst, int dtype, int B, int H, int W, int C, void* stream) {
  FFVC_CHECK_ARG(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && (C % 4) == 0, "ffvc_sumpool2x2: bad args (C=%d)", C);
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * H * W * (C / 4);
  DISPATCH_DT(dtype, T, hipLaunchKernelGGL((sumpool2_kernel<T>), dim3(ew_grid(n, 256)), dim3(256), 0, st,
                                           (const T*)src, (T*)dst, B, H, W, C));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_rownorm_sq(const float* x, float* out, int64_t rows, int dim, void* stream) {
  FFVC_CHECK_ARG(x && out && rows > 0 && dim > 0, "ffvc_rownorm_sq: bad args");
  hipLaunchKernelGGL(rownorm_kernel, dim3(ew_grid(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, dim);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_vq_argmin(const float* dot, const float* xnorm, const float* cnorm, int64_t* idx, int64_t rows,
                              int ncodes, int64_t ld, void* stream) {
  FFVC_CHECK_ARG(dot && xnorm && cnorm && idx && rows > 0 && ncodes > 0 && ld >= ncodes, "ffvc_vq_argmin: bad args");
  hipLaunchKernelGGL(vq_argmin_kernel, dim3(ew_grid(rows, 4)), dim3(256), 0, (hipStream_t)stream, dot, xnorm, cnorm,
                     idx, rows, ncodes, ld);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_gather_rows(const float* table, const int64_t* idx, const float* pos, int period, void* out,
                                int out_dtype, int64_t rows, int dim, void* stream) {
  FFVC_CHECK_ARG(table && idx && out && rows > 0 && dim > 0, "ffvc_gather_rows: bad args");
  FFVC_CHECK_ARG(!pos || period > 0, "ffvc_gather_rows: pos needs period");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(out_dtype, DT, hipLaunchKernelGGL((gather_rows_kernel<DT>), dim3(ew_grid(rows * dim, 1024)), dim3(256), 0,
                                                st, table, idx, pos, period > 0 ? period : 1, (DT*)out, rows, dim));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_eot_gather(const void* x, int x_dtype, const int64_t* tokens, float* out, int B, int L, int dim,
                               void* stream) {
  FFVC_CHECK_ARG(x && tokens && out && B > 0 && L > 0 && dim > 0, "ffvc_eot_gather: bad args");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(x_dtype, XT, hipLaunchKernelGGL((eot_gather_kernel<XT>), dim3(B), dim3(64), 0, st, (const XT*)x, tokens,
                                              out, L, dim));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_cutouts_fwd(const float* xr, const float* noise, const float* facs, void* out, int out_dtype, int B,
                                int H, int W, int cut, int cutn, int patch, float mean_r, float mean_g, float mean_b,
                                float std_r, float std_g, float std_b, void* stream) {
  FFVC_CHECK_ARG(xr && out, "ffvc_cutouts_fwd: null pointer");
  FFVC_CHECK_ARG(B > 0 && H > 0 && W > 0 && cut > 0 && cutn > 0 && patch > 0 && cut % patch == 0,
                 "ffvc_cutouts_fwd: bad geometry cut=%d patch=%d", cut, patch);
  FFVC_CHECK_ARG(H <= APOOL_MAX && W <= APOOL_MAX && cut <= APOOL_MAX, "ffvc_cutouts_fwd: sides above 32768 are not supported");
  FFVC_CHECK_ARG((noise == nullptr) == (facs == nullptr), "ffvc_cutouts_fwd: noise and facs go together");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * 3 * cut * cut;
  DISPATCH_DT(out_dtype, OT, hipLaunchKernelGGL((cutouts_fwd_kernel<OT>), dim3(ew_grid(n, 256)), dim3(256), 0, st, xr,
                                                noise, facs, (OT*)out, B, H, W, cut, cutn, patch, mean_r, mean_g,
                                                mean_b, std_r, std_g, std_b));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_cutouts_bwd(const float* xr, const void* gout, int g_dtype, float* dxr, int B, int H, int W, int cut,
                                int cutn, int patch, float std_r, float std_g, float std_b, void* stream) {
  FFVC_CHECK_ARG(xr && gout && dxr, "ffvc_cutouts_bwd: null pointer");
  FFVC_CHECK_ARG(B > 0 && H > 0 && W > 0 && cut > 0 && cutn > 0 && patch > 0 && cut % patch == 0 && H >= cut && W >= cut,
                 "ffvc_cutouts_bwd: bad geometry");
  FFVC_CHECK_ARG(H <= APOOL_MAX && W <= APOOL_MAX && cut <= APOOL_MAX, "ffvc_cutouts_bwd: sides above 32768 are not supported");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * H * W * 3;
  DISPATCH_DT(g_dtype, GT, hipLaunchKernelGGL((cutouts_bwd_kernel<GT>), dim3(ew_grid(n, 256)), dim3(256), 0, st, xr,
                                              (const GT*)gout, dxr, B, H, W, cut, cutn, patch, std_r, std_g, std_b));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_spherical_loss(const float* embed, const float* feats, float* rowloss, float* loss, float* dembed,
                                   int N, int B, int D, float coef, void* stream) {
  FFVC_CHECK_ARG(embed && feats && rowloss && loss && N > 0 && B > 0 && D > 0 && N % B == 0,
                 "ffvc_spherical_loss: bad args N=%d B=%d", N, B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sph_loss_rows_kernel, dim3(ew_grid(N, 4)), dim3(256), 0, st, embed, feats, rowloss, dembed, N, B, D,
                     coef);
  hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, rowloss, loss, N, coef / (float)N);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_adam(float* p, const float* g, float* m, float* v, void* shadow, int shadow_dtype, int64_t n,
                         float lr, float beta1, float beta2, float eps, int step, float grad_scale, float* ema,
                         float ema_weight, const float* dev_scale, uint32_t* nonfinite_count, const float* dev_hyper,
                         void* stream) {
  FFVC_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "ffvc_adam: bad args");
  FFVC_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                     ((uintptr_t)v % 16) == 0 && ((uintptr_t)shadow % 16) == 0 && ((uintptr_t)ema % 16) == 0,
                 "ffvc_adam: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  const int grid = ew_grid(n / 4 + 1, 256);
  if (shadow && shadow_dtype == FFVC_BF16)
    hipLaunchKernelGGL((adam_kernel<uint16_t>), dim3(grid), dim3(256), 0, st, p, g, m, v, (uint16_t*)shadow, n, lr, beta1,
                       beta2, eps, bc1, bc2s, grad_scale, ema, ema_weight, dev_scale, nonfinite_count, dev_hyper);
  else if (shadow && shadow_dtype == FFVC_F16)
    hipLaunchKernelGGL((adam_kernel<f16_t>), dim3(grid), dim3(256), 0, st, p, g, m, v, (f16_t*)shadow, n, lr, beta1,
                       beta2, eps, bc1, bc2s, grad_scale, ema, ema_weight, dev_scale, nonfinite_count, dev_hyper);
  else
    hipLaunchKernelGGL((adam_kernel<float>), dim3(grid), dim3(256), 0, st, p, g, m, v, (float*)shadow, n, lr, beta1, beta2,
                       eps, bc1, bc2s, grad_scale, ema, ema_weight, dev_scale, nonfinite_count, dev_hyper);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_dropout(const void* x, int x_dtype, const float* residual, void* y, int y_dtype, int64_t n, float p,
                            uint32_t seed, void* stream) {
  FFVC_CHECK_ARG(x && y && n > 0 && p >= 0.0f && p < 1.0f, "ffvc_dropout: bad args (p=%f)", (double)p);
  FFVC_CHECK_ARG(ffvc_dtype_ok(x_dtype) && ffvc_dtype_ok(y_dtype), "ffvc_dropout: bad dtype");
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const float inv_keep = 1.0f / (1.0f - p);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(x_dtype, XT, DISPATCH_DT(y_dtype, YT,
              hipLaunchKernelGGL((dropout_kernel<XT, YT>), dim3(ew_grid(n, 1024)), dim3(256), 0, st, (const XT*)x, residual,
                                 (YT*)y, n, seed, thresh, inv_keep)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_mean_sq(const float* x, float* out, int64_t n, void* stream) {
  FFVC_CHECK_ARG(x && out && n > 0, "ffvc_mean_sq: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return FFVC_E_BADARG;
  hipLaunchKernelGGL(mean_sq_kernel, dim3(ew_grid(n, 2048)), dim3(256), 0, st, x, out, n, 1.0f / (float)n);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_mean_sq_bwd(const float* x, const float* g, float* dx, int64_t n, void* stream) {
  FFVC_CHECK_ARG(x && g && dx && n > 0, "ffvc_mean_sq_bwd: bad args");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, g, dx, n,
                     2.0f / (float)n);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_tv_loss_fwd(const float* x, float* out, int B, int H, int W, int C, void* stream) {
  FFVC_CHECK_ARG(x && out && B > 0 && H > 1 && W > 1 && C > 0, "ffvc_tv_loss_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(float), st) != hipSuccess) return FFVC_E_BADARG;
  const int64_t n = (int64_t)B * H * W * C;
  const float inh = 1.0f / ((float)B * C * (H - 1) * W), inw = 1.0f / ((float)B * C * H * (W - 1));
  hipLaunchKernelGGL(tv_fwd_kernel, dim3(ew_grid(n, 2048)), dim3(256), 0, st, x, out, B, H, W, C, inh, inw);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_tv_loss_bwd(const float* x, const float* g, float* dx, int B, int H, int W, int C, void* stream) {
  FFVC_CHECK_ARG(x && g && dx && B > 0 && H > 1 && W > 1 && C > 0, "ffvc_tv_loss_bwd: bad args");
  const int64_t n = (int64_t)B * H * W * C;
  const float inh = 1.0f / ((float)B * C * (H - 1) * W), inw = 1.0f / ((float)B * C * H * (W - 1));
  hipLaunchKernelGGL(tv_bwd_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, g, dx, B, H, W, C, inh,
                     inw);
  FFVC_LAUNCH_CHECK();
  return 0;
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float gscale, float* __restrict__ out) {
  const float total = sqrtf(sumsq[0]) * fabsf(gscale);        // norm of the gradients as the optimizer will see them
  // torch.nn.utils.clip_grad_norm_ (main.py:833-834).  A non-finite norm (an inf / NaN gradient somewhere: f16 overflow) makes
  // the coefficient NaN ON PURPOSE: every scaled gradient then fails the Adam kernel's finite test and the whole step is
  // skipped — fminf alone would drop the NaN and return 1, an inf norm would give 0 and a half-applied step.
  const bool finite = total == total && total < 3.0e38f;
  out[0] = finite ? fminf(1.0f, max_norm / (total + 1e-6f)) : __builtin_nanf("");
  out[1] = total;
}

extern "C" int ffvc_clip_coef(const float* sumsq, float max_norm, float grad_scale, float* out, void* stream) {
  FFVC_CHECK_ARG(sumsq && out && max_norm > 0.f, "ffvc_clip_coef: bad args");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, grad_scale, out);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_sumsq(const float* x, float* out, int64_t n, void* stream) {
  FFVC_CHECK_ARG(x && out && n > 0, "ffvc_sumsq: bad args");
  hipLaunchKernelGGL(sumsq_kernel, dim3(ew_grid(n, 2048)), dim3(256), 0, (hipStream_t)stream, x, out, n);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_axpby(const float* x, float* y, int64_t n, float a, float b, void* stream) {
  FFVC_CHECK_ARG(x && y && n > 0, "ffvc_axpby: bad args");
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, y, n, a, b);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_rowsum(const void* x, int dtype, float* out, int64_t rows, int cols, int period, int accumulate,
                           void* stream) {
  FFVC_CHECK_ARG(x && out && rows > 0 && cols > 0 && period > 0, "ffvc_rowsum: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)period * sizeof(float), st);
    if (e != hipSuccess) {
      ffvc_set_error("ffvc_rowsum: memset failed: %s", hipGetErrorString(e));
      return (int)e;
    }
  }
  if (dtype != FFVC_F32 && (cols % 8) == 0 && (rows % period) == 0 && rows / period >= 8 && ((uintptr_t)x % 16) == 0) {
    const int64_t per = rows / period;
    int chunks = (int)((4096 + period - 1) / period);            // ~4096 waves in flight
    if (chunks > per / 4) chunks = (int)(per / 4);
    if (chunks < 1) chunks = 1;
    const int rpc = (int)((per + chunks - 1) / chunks);
    chunks = (int)((per + rpc - 1) / rpc);
    const int64_t waves = (int64_t)period * chunks;
    DISPATCH_DT(dtype, T, {
      if constexpr (sizeof(T) == 2)
        hipLaunchKernelGGL((rowsum_grouped_kernel<T>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const T*)x, out, rows, cols,
                           period, chunks, rpc);
    });
    FFVC_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_DT(dtype, T, hipLaunchKernelGGL((rowsum_kernel<T>), dim3(ew_grid(rows, 4)), dim3(256), 0, st, (const T*)x,
                                           out, rows, cols, period));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_copy_rows(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows,
                              int cols, void* stream) {
  FFVC_CHECK_ARG(src && dst && rows > 0 && cols > 0, "ffvc_copy_rows: bad args");
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ew_grid(rows * cols, 1024)), dim3(256), 0, (hipStream_t)stream, src,
                     src_stride, dst, dst_stride, rows, cols);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_im2col3x3(const void* x, int x_dtype, void* out, int out_dtype, int B, int H, int W, int C, int Kp,
                              void* stream) {
  FFVC_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C > 0 && Kp >= 9 * C, "ffvc_im2col3x3: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * H * W * Kp;
  DISPATCH_DT(x_dtype, XT, DISPATCH_DT(out_dtype, OT,
              hipLaunchKernelGGL((im2col3x3_kernel<XT, OT>), dim3(ew_grid(n, 1024)), dim3(256), 0, st, (const XT*)x,
                                 (OT*)out, B, H, W, C, Kp)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_mul_dev_scalar(const float* x, const float* s, float* y, int64_t n, void* stream) {
  FFVC_CHECK_ARG(x && s && y && n > 0, "ffvc_mul_dev_scalar: bad args");
  hipLaunchKernelGGL(mul_dev_scalar_kernel, dim3(ew_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, x, s, y, n);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_copy2d(const void* src, int src_dtype, int64_t src_ld, void* dst, int dst_dtype, int64_t dst_ld,
                           int64_t rows, int cols, int dst_cols, void* stream) {
  FFVC_CHECK_ARG(src && dst && rows > 0 && cols > 0 && dst_cols > 0 && src_ld >= cols && dst_ld >= dst_cols,
                 "ffvc_copy2d: bad args");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(src_dtype, ST, DISPATCH_DT(dst_dtype, DT,
              hipLaunchKernelGGL((copy2d_kernel<ST, DT>), dim3(ew_grid(rows * dst_cols, 1024)), dim3(256), 0, st,
                                 (const ST*)src, src_ld, (DT*)dst, dst_ld, rows, cols, dst_cols)));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_slab_reduce(const float* slabs, float* y, int64_t n, int nslab, int accumulate, void* stream) {
  FFVC_CHECK_ARG(slabs && y && n > 0 && nslab > 0, "ffvc_slab_reduce: bad args");
  FFVC_CHECK_ARG(((uintptr_t)slabs % 16) == 0 && ((uintptr_t)y % 16) == 0 && (n % 4) == 0,
                 "ffvc_slab_reduce: needs 16-byte aligned buffers and n %% 4 == 0");
  const dim3 grid(ew_grid(n / 4, 256));
  hipStream_t st = (hipStream_t)stream;
  switch (nslab) {
    case 2: hipLaunchKernelGGL(slab_reduce_kernel<2>, grid, dim3(256), 0, st, slabs, y, n, nslab, accumulate); break;
    case 3: hipLaunchKernelGGL(slab_reduce_kernel<3>, grid, dim3(256), 0, st, slabs, y, n, nslab, accumulate); break;
    case 4: hipLaunchKernelGGL(slab_reduce_kernel<4>, grid, dim3(256), 0, st, slabs, y, n, nslab, accumulate); break;
    case 8: hipLaunchKernelGGL(slab_reduce_kernel<8>, grid, dim3(256), 0, st, slabs, y, n, nslab, accumulate); break;
    default: hipLaunchKernelGGL(slab_reduce_kernel<0>, grid, dim3(256), 0, st, slabs, y, n, nslab, accumulate); break;
  }
  FFVC_LAUNCH_CHECK();
  return 0;
}

static int augment_fwd_impl(bool seq, const float* pooled, const float* pinv, const float* ainv, const float* cmat, const float* coff,
                            const float* cj, const int32_t* erase, const float* noise, const float* facs, void* out, int out_dtype, int B,
                            int S, int S_src, int cutn, int patch, float mean_r, float mean_g, float mean_b, float std_r,
                            float std_g, float std_b, void* stream) {
  FFVC_CHECK_ARG(pooled && pinv && ainv && cmat && erase && out, "ffvc_augment_fwd: null pointer");
  FFVC_CHECK_ARG(B > 0 && S > 1 && S_src > 1 && cutn > 0 && patch > 0 && S % patch == 0, "ffvc_augment_fwd: bad geometry");
  FFVC_CHECK_ARG((noise == nullptr) == (facs == nullptr), "ffvc_augment_fwd: noise and facs go together");
  hipStream_t st = (hipStream_t)stream;
  FFVC_CHECK_ARG((int64_t)cutn * B <= 65535, "ffvc_augment_fwd: more than 65535 cutouts per launch");
  const dim3 grid((S * S + 255) / 256, cutn * B);
  if (seq) {
    DISPATCH_DT(out_dtype, OT, hipLaunchKernelGGL((augment_fwd_kernel<OT, true>), grid, dim3(256), 0, st, pooled,
                                                  pinv, ainv, cmat, erase, noise, facs, coff, cj, (OT*)out, B, S, S_src, cutn, patch, mean_r,
                                                  mean_g, mean_b, std_r, std_g, std_b));
  } else {
    DISPATCH_DT(out_dtype, OT, hipLaunchKernelGGL((augment_fwd_kernel<OT, false>), grid, dim3(256), 0, st, pooled,
                                                  pinv, ainv, cmat, erase, noise, facs, coff, cj, (OT*)out, B, S, S_src, cutn, patch, mean_r,
                                                  mean_g, mean_b, std_r, std_g, std_b));
  }
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_augment_fwd(const float* pooled, const float* pinv, const float* ainv, const float* cmat, const float* coff,
                                const float* cj, const int32_t* erase, const float* noise, const float* facs, void* out, int out_dtype, int B,
                                int S, int S_src, int cutn, int patch, float mean_r, float mean_g, float mean_b, float std_r,
                                float std_g, float std_b, void* stream) {
  return augment_fwd_impl(false, pooled, pinv, ainv, cmat, coff, cj, erase, noise, facs, out, out_dtype, B, S, S_src, cutn, patch, mean_r, mean_g,
                          mean_b, std_r, std_g, std_b, stream);
}

extern "C" int ffvc_augment_seq_fwd(const float* pooled, const float* pinv, const float* ainv, const float* cmat, const float* coff,
                                    const float* cj, const int32_t* erase, const float* noise, const float* facs, void* out, int out_dtype,
                                    int B, int S, int S_src, int cutn, int patch, float mean_r, float mean_g, float mean_b, float std_r,
                                    float std_g, float std_b, void* stream) {
  return augment_fwd_impl(true, pooled, pinv, ainv, cmat, coff, cj, erase, noise, facs, out, out_dtype, B, S, S_src, cutn, patch, mean_r, mean_g,
                          mean_b, std_r, std_g, std_b, stream);
}

static int augment_bwd_impl(bool seq, const void* gout, int g_dtype, const float* pinv, const float* ainv, const float* cmat,
                            const int32_t* erase, const float* pooled, const float* coff, const float* cj, float* dpooled,
                            int B, int S, int S_src, int cutn, int patch, float std_r, float std_g, float std_b,
                            void* stream) {
  FFVC_CHECK_ARG(gout && pinv && ainv && cmat && erase && dpooled, "ffvc_augment_bwd: null pointer");
  FFVC_CHECK_ARG(cj == nullptr || pooled != nullptr, "ffvc_augment_bwd: the colour jitter's backward needs the forward's source image");
  FFVC_CHECK_ARG(B > 0 && S > 1 && S_src > 1 && cutn > 0 && patch > 0 && S % patch == 0, "ffvc_augment_bwd: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dpooled, 0, (size_t)B * 3 * S_src * S_src * sizeof(float), st);
  if (e != hipSuccess) {
    ffvc_set_error("ffvc_augment_bwd: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  const int64_t n = (int64_t)cutn * B * S * S;
  static int tiled = -1;
  if (tiled < 0) {
    const char* e = getenv("FFVC_AUG_BWD_TILED");
    tiled = e ? atoi(e) : 1;
  }
  const int tiles = ((S + AUGTX - 1) / AUGTX) * ((S + AUGTY - 1) / AUGTY);
  if (tiled && (int64_t)cutn * B <= 65535) {
    if (seq) {
      DISPATCH_DT(g_dtype, GT, hipLaunchKernelGGL((augment_bwd_tiled_kernel<GT, true>), dim3(tiles, cutn * B), dim3(256), 0, st, (const GT*)gout, pinv,
                                                  ainv, cmat, erase, pooled, coff, cj, dpooled, B, S, S_src, patch, std_r, std_g, std_b));
    } else {
      DISPATCH_DT(g_dtype, GT, hipLaunchKernelGGL((augment_bwd_tiled_kernel<GT, false>), dim3(tiles, cutn * B), dim3(256), 0, st, (const GT*)gout, pinv,
                                                  ainv, cmat, erase, pooled, coff, cj, dpooled, B, S, S_src, patch, std_r, std_g, std_b));
    }
  } else if (seq) {
    DISPATCH_DT(g_dtype, GT, hipLaunchKernelGGL((augment_bwd_kernel<GT, true>), dim3(ew_grid(n, 256)), dim3(256), 0, st,
                                                (const GT*)gout, pinv, ainv, cmat, erase, pooled, coff, cj, dpooled, B, S, S_src, cutn, patch, std_r,
                                                std_g, std_b));
  } else {
    DISPATCH_DT(g_dtype, GT, hipLaunchKernelGGL((augment_bwd_kernel<GT, false>), dim3(ew_grid(n, 256)), dim3(256), 0, st,
                                                (const GT*)gout, pinv, ainv, cmat, erase, pooled, coff, cj, dpooled, B, S, S_src, cutn, patch, std_r,
                                                std_g, std_b));
  }
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_augment_bwd(const void* gout, int g_dtype, const float* pinv, const float* ainv, const float* cmat,
                                const int32_t* erase, const float* pooled, const float* coff, const float* cj, float* dpooled,
                                int B, int S, int S_src, int cutn, int patch, float std_r, float std_g, float std_b, void* stream) {
  return augment_bwd_impl(false, gout, g_dtype, pinv, ainv, cmat, erase, pooled, coff, cj, dpooled, B, S, S_src, cutn, patch, std_r, std_g, std_b,
                          stream);
}

extern "C" int ffvc_augment_seq_bwd(const void* gout, int g_dtype, const float* pinv, const float* ainv, const float* cmat,
                                    const int32_t* erase, const float* pooled, const float* coff, const float* cj, float* dpooled,
                                    int B, int S, int S_src, int cutn, int patch, float std_r, float std_g, float std_b, void* stream) {
  FFVC_CHECK_ARG(S == S_src, "ffvc_augment_seq_bwd: the sequential form needs source, intermediate and output images of one size");
  return augment_bwd_impl(true, gout, g_dtype, pinv, ainv, cmat, erase, pooled, coff, cj, dpooled, B, S, S_src, cutn, patch, std_r, std_g, std_b,
                          stream);
}

extern "C" int ffvc_avgpool_patches_fwd(const float* x, void* out, int out_dtype, int N, int S, int So, int patch, float mean_r,
                                        float mean_g, float mean_b, float std_r, float std_g, float std_b, void* stream) {
  FFVC_CHECK_ARG(x && out && N > 0 && S > 0 && So > 0 && patch > 0 && So % patch == 0, "ffvc_avgpool_patches_fwd: bad args");
  FFVC_CHECK_ARG(S <= APOOL_MAX && So <= APOOL_MAX, "ffvc_avgpool_patches_fwd: sides above 32768 are not supported");
  const int64_t n = (int64_t)N * 3 * So * So;
  DISPATCH_DT(out_dtype, OT, hipLaunchKernelGGL((avgpool_patches_fwd_kernel<OT>), dim3(ew_grid(n, 256)), dim3(256), 0,
                                                (hipStream_t)stream, x, (OT*)out, N, S, So, patch, mean_r, mean_g, mean_b,
                                                std_r, std_g, std_b));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_avgpool_patches_bwd(const void* gout, int g_dtype, float* dx, int N, int S, int So, int patch, float std_r,
                                        float std_g, float std_b, void* stream) {
  FFVC_CHECK_ARG(gout && dx && N > 0 && S > 0 && So > 0 && patch > 0 && So % patch == 0, "ffvc_avgpool_patches_bwd: bad args");
  FFVC_CHECK_ARG(S <= APOOL_MAX && So <= APOOL_MAX, "ffvc_avgpool_patches_bwd: sides above 32768 are not supported");
  const int64_t n = (int64_t)N * 3 * S * S;
  DISPATCH_DT(g_dtype, GT, hipLaunchKernelGGL((avgpool_patches_bwd_kernel<GT>), dim3(ew_grid(n, 256)), dim3(256), 0,
                                              (hipStream_t)stream, (const GT*)gout, dx, N, S, So, patch, std_r, std_g, std_b));
  FFVC_LAUNCH_CHECK();
  return 0;
}
