// norm.hip — HBM-bound normalisation kernels: LayerNorm, GroupNorm(+swish) on NHWC, row softmax.
// All statistics, reductions and transcendental math are fp32 (fp64 for the GroupNorm moment
// combine); storage is bf16 or fp32 per tensor.  One wavefront (64 lanes) owns one row /
// pixel-strip, 16-byte vector accesses, no LDS round trip for row reductions (wave shuffles).
#include <stdlib.h>

// The big operands of these kernels are streamed once per pass: `nt` loads / stores keep them from evicting the GEMM
// operands of concurrently running kernels out of L2 / MALL (step: -0.9 ms; GroupNorm at 256^2: +3-5 %).
#ifndef FFVC_STREAM_NT
#define FFVC_STREAM_NT 1
#endif
#include "common.h"
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace {

constexpr int LN_MAXE = 32;  // elements cached per lane -> dim <= 2048

template <int VEC, typename T>
__device__ __forceinline__ void ld_vec(const T* p, float* out) {
  if constexpr (VEC == 4) {
    f32x4_t v = load4s(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = v[j];
  } else {
    out[0] = ElemTraits<T>::load(p);
  }
}
template <int VEC, typename T>
__device__ __forceinline__ void st_vec(T* p, const float* in) {
  if constexpr (VEC == 4) {
    f32x4_t v = {in[0], in[1], in[2], in[3]};
    store4s(p, v);
  } else {
    ElemTraits<T>::store(p, in[0]);
  }
}

// ------------------------------- LayerNorm ---------------------------------
// y = (x - mean) * rstd * gamma + beta over the last dim (mlp_mixer_pytorch.py:11,14,37;
// cloob.py:170-176; vitgan.py:14,21).  mean/rstd are saved for the backward.
template <int VEC, typename XT, typename YT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const XT* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, YT* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     int64_t rows, int dim, float eps, uint8_t* __restrict__ y8 = nullptr,
                                                     float* __restrict__ f8_state = nullptr, int f8_fmt = 0) {
  // y8 (VEC == 4, 16-bit YT): the row also leaves as fp8 bytes = saturate(round_YT(y) * f8_state[0]) and f8_state[1] collects
  // max |round_YT(y)| — exactly what ffvc_fp8_quant would produce from y; y itself may then be NULL
  constexpr int NIT = LN_MAXE / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float f8s = y8 ? f8_state[0] : 0.0f;
  float f8m = 0.0f;
  // Round 6: software-pipelined over the rows of a wave.  One wave per row with everything in lock step made the launch a sequence of
  // chip-wide phases — load burst, two dependent wave reductions, store burst — with HBM idle in between (32 us for 100 MB = 3.1 TB/s,
  // profiles/r05_kernel_trace_bench_cfg2.txt).  Now a wave owns several rows (the launcher caps the grid) and requests row i + 1
  // before it reduces and stores row i: the next row's loads are in flight under the current row's arithmetic and stores.
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t row = (int64_t)blockIdx.x * 4 + wave;
  float vn[LN_MAXE];
  auto fetch = [&](int64_t r) {
    const XT* xr = x + r * dim;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) ld_vec<VEC>(xr + idx, &vn[k * VEC]);
    }
  };
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    float v[LN_MAXE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          v[k * VEC + j] = vn[k * VEC + j];
          s += v[k * VEC + j];
        }
      }
    }
    if (row + stride < rows) fetch(row + stride);
    const float mu = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = v[k * VEC + j] - mu;
          q += d * d;
        }
      }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / dim + eps);
    YT* yr = y + row * dim;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        float o[VEC], g[VEC], b[VEC];
        ld_vec<VEC>(gamma + idx, g);
        ld_vec<VEC>(beta + idx, b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = (v[k * VEC + j] - mu) * rs * g[j] + b[j];
        if (y) st_vec<VEC>(yr + idx, o);
        if constexpr (VEC == 4 && sizeof(YT) == 2) {
          if (y8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              o[j] = lo_round<YT>(o[j]);
              f8m = fmaxf(f8m, fabsf(o[j]));
            }
            *(uint32_t*)(y8 + row * dim + idx) = f8_pack4(f8_fmt, o[0] * f8s, o[1] * f8s, o[2] * f8s, o[3] * f8s);
          }
        }
      }
    }
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
  if (y8) f8_amax_block(f8m, f8_state);
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) [+ dres],  g = dy * gamma.
// Each workgroup owns a strip of rows and emits one partial row of dgamma / dbeta
// (part_g/part_b: [gridDim.x, dim], reduced afterwards by ffvc_colsum); NULL skips them.
// MAXE = elements cached per lane (16: dim <= 1024, 32: dim <= 2048): the kernel is latency-bound on its row loads, so
// register count (= waves in flight per SIMD) is what sets its bandwidth.
// NWB = waves per workgroup.  The parameter gradients leave the workgroup as one fp32 atomic per column, and same-address
// atomics serialise in L2 (~75 ns each): their cost is proportional to the NUMBER OF WORKGROUPS, so large launches use 16
// waves per workgroup (a quarter of the workgroups at the same number of waves in flight) — an experiment (FFVC_LN_WIDE=1):
// the 16-way LDS combine costs more than the atomics it saves, so the 4-wave form stays the default.
template <int VEC, typename DYT, typename XT, int MAXE = LN_MAXE, int NWB = 4>
__global__ __launch_bounds__(64 * NWB, (MAXE <= 16 ? 4 : 2)) void ln_bwd_kernel(const DYT* __restrict__ dy, const XT* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const XT* __restrict__ dres,
                                                     XT* __restrict__ dx, float* __restrict__ part_g,
                                                     float* __restrict__ part_b, int64_t rows, int dim,
                                                     int rows_per_block, int acc_mode, DYT* __restrict__ dx_lo,
                                                     int64_t pstride, int ln_plain_combine) {
  constexpr int NIT = MAXE / VEC;
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];  // [2][dim] when partials requested
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool want_p = part_g != nullptr;
  float ag[MAXE], ab[MAXE];
#pragma unroll
  for (int i = 0; i < MAXE; ++i) ag[i] = ab[i] = 0.f;
  if (want_p && !ln_plain_combine) {
    for (int i = threadIdx.x; i < 2 * dim; i += 64 * NWB) ln_smem[i] = 0.f;
    __syncthreads();
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  for (int64_t row = r0 + wave; row < r1; row += NWB) {
    const float mu = mean[row], rs = rstd[row];
    float g[MAXE], xh[MAXE], rsd[MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        float d[VEC], xv[VEC], gm[VEC];
        ld_vec<VEC>(dy + row * dim + idx, d);
        ld_vec<VEC>(x + row * dim + idx, xv);
        ld_vec<VEC>(gamma + idx, gm);
        if (dres) ld_vec<VEC>(dres + row * dim + idx, &rsd[k * VEC]);   // issued with the other row loads, used after the reductions
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float h = (xv[j] - mu) * rs;
          xh[k * VEC + j] = h;
          g[k * VEC + j] = d[j] * gm[j];
          s1 += g[k * VEC + j];
          s2 += g[k * VEC + j] * h;
          ag[k * VEC + j] += d[j] * h;
          ab[k * VEC + j] += d[j];
        }
      }
    }
    s1 = wave_sum(s1) / dim;
    s2 = wave_sum(s2) / dim;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = rs * (g[k * VEC + j] - s1 - xh[k * VEC + j] * s2);
        if (dres) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] += rsd[k * VEC + j];
        }
        st_vec<VEC>(dx + row * dim + idx, o);
        if (dx_lo) st_vec<VEC>(dx_lo + row * dim + idx, o);   // bf16 copy for the GEMM that consumes this gradient
      }
    }
  }
  if (want_p && ln_plain_combine) {
    // every wave parks its column sums in its own LDS rows with plain vector stores ([wave][2][dim]); the column loop below adds
    // the NWB rows — the LDS float atomics this replaces serialise per lane (32 ds_add_f32 per lane and wave)
    float* mine = ln_smem + (size_t)wave * 2 * dim;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        st_vec<VEC>(mine + idx, &ag[k * VEC]);
        st_vec<VEC>(mine + dim + idx, &ab[k * VEC]);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * dim; i += 64 * NWB) {
      float t = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NWB; ++w2) t += ln_smem[(size_t)w2 * 2 * dim + i];
      if (acc_mode) atomicAdd((i < dim ? part_g : part_b - dim) + i, t);
      else (i < dim ? part_g : part_b - dim)[(int64_t)blockIdx.x * pstride + i] = t;
    }
    return;
  }
  if (want_p) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          atomicAdd(&ln_smem[idx + j], ag[k * VEC + j]);
          atomicAdd(&ln_smem[dim + idx + j], ab[k * VEC + j]);
        }
      }
    }
    __syncthreads();
    if (acc_mode) {   // part_g / part_b are the [dim] gradients themselves: one fp32 atomic per column and workgroup
      for (int i = threadIdx.x; i < dim; i += 64 * NWB) {
        atomicAdd(part_g + i, ln_smem[i]);
        atomicAdd(part_b + i, ln_smem[dim + i]);
      }
    } else {
      for (int i = threadIdx.x; i < dim; i += 64 * NWB) {
        part_g[(int64_t)blockIdx.x * pstride + i] = ln_smem[i];
        part_b[(int64_t)blockIdx.x * pstride + i] = ln_smem[dim + i];
      }
    }
  }
}

// LayerNorm backward of a FROZEN layer on the fp32 residual stream (the CLIP towers) whose result is the operand of an fp8 dgrad next:
// dx = LN'(dy) (+ dres) in fp32 and the same values, rounded to dy's 16-bit type first, as fp8 bytes in the consumer's scale — what
// ffvc_fp8_quant makes of the 16-bit copy (`dx_lo`) the plain kernel writes for that consumer.  No parameter gradients: the kernel keeps
// the row in 3 x MAXE registers and nothing else.
template <typename DYT, int MAXE>
__global__ __launch_bounds__(256) void ln_bwd_f8_kernel(const DYT* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ dres, float* __restrict__ dx, uint8_t* __restrict__ dx8,
                                                        float* __restrict__ f8_state, int f8_fmt, int64_t rows, int dim) {
  constexpr int NIT = MAXE / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float f8s = f8_state[0];
  float f8m = 0.0f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float g[MAXE], xh[MAXE], rsd[MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * 4;
      if (idx < dim) {
        float d[4], xv[4], gm[4];
        ld_vec<4>(dy + row * dim + idx, d);
        ld_vec<4>(x + row * dim + idx, xv);
        ld_vec<4>(gamma + idx, gm);
        if (dres) ld_vec<4>(dres + row * dim + idx, &rsd[k * 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float h = (xv[j] - mu) * rs;
          xh[k * 4 + j] = h;
          g[k * 4 + j] = d[j] * gm[j];
          s1 += g[k * 4 + j];
          s2 += g[k * 4 + j] * h;
        }
      }
    }
    s1 = wave_sum(s1) / dim;
    s2 = wave_sum(s2) / dim;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * 4;
      if (idx < dim) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = rs * (g[k * 4 + j] - s1 - xh[k * 4 + j] * s2);
        if (dres) {
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] += rsd[k * 4 + j];
        }
        st_vec<4>(dx + row * dim + idx, o);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = lo_round<DYT>(o[j]);
          f8m = fmaxf(f8m, fabsf(o[j]));
        }
        *(uint32_t*)(dx8 + row * dim + idx) = f8_pack4(f8_fmt, o[0] * f8s, o[1] * f8s, o[2] * f8s, o[3] * f8s);
      }
    }
  }
  f8_amax_block(f8m, f8_state);
}

// ------------------------- self-modulated LayerNorm ------------------------
// vitgan.py:8-21 (SLN): out = gamma_s * w * LN(hl) + beta_s * w, gamma_s/beta_s scalar parameters, w the
// per-token modulation tensor.  hl, w fp32 [rows, dim]; out in the compute dtype.
template <int VEC, typename YT>
__global__ __launch_bounds__(256) void sln_fwd_kernel(const float* __restrict__ hl, const float* __restrict__ w,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ gs, const float* __restrict__ bs,
                                                      YT* __restrict__ y, float* __restrict__ mean,
                                                      float* __restrict__ rstd, int64_t rows, int dim, float eps) {
  constexpr int NIT = LN_MAXE / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float g_s = gs[0], b_s = bs[0];
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float v[LN_MAXE];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        ld_vec<VEC>(hl + row * dim + idx, &v[k * VEC]);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += v[k * VEC + j];
      }
    }
    const float mu = wave_sum(s) / dim;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float d = v[k * VEC + j] - mu;
          q += d * d;
        }
      }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / dim + eps);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        float o[VEC], g[VEC], b[VEC], wv[VEC];
        ld_vec<VEC>(gamma + idx, g);
        ld_vec<VEC>(beta + idx, b);
        ld_vec<VEC>(w + row * dim + idx, wv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = (g_s * ((v[k * VEC + j] - mu) * rs * g[j] + b[j]) + b_s) * wv[j];
        st_vec<VEC>(y + row * dim + idx, o);
      }
    }
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
  }
}

// Backward of SLN: dhl (+ dres), dw, LayerNorm dgamma/dbeta partial rows, and partial sums of the two scalars
// (part_s[block][2] = {sum dy*w*ln, sum dy*w}).
// X2 (ffvc_sln_bwd_acc2, opt-in): the second scalar gradient has its own address and dw may be a running sum.  A separate instantiation:
// the default one must stay the code of rounds 3-5 — with the two extra operands compiled into it, 16 of 100 backward passes of a 9-block
// generator came out with one sample's gradients changed at f16-rounding level, with the old kernel in the same library 0 of 100
// (tools/r6/vitgan_determinism_old.py; mechanism not found: the kernel alone is bit-reproducible).
template <int VEC, typename DYT, bool X2 = false>
__global__ __launch_bounds__(256) void sln_bwd_kernel(const DYT* __restrict__ dy, const float* __restrict__ hl,
                                                      const float* __restrict__ w, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ gs,
                                                      const float* __restrict__ bs, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ dres,
                                                      float* __restrict__ dhl, float* __restrict__ dw,
                                                      float* __restrict__ part_g, float* __restrict__ part_b,
                                                      float* __restrict__ part_s, int64_t rows, int dim,
                                                      int rows_per_block, int acc_mode, float* __restrict__ part_s1,
                                                      int dw_acc) {
  constexpr int NIT = LN_MAXE / VEC;
  extern __shared__ __attribute__((aligned(16))) float ln_smem[];  // [2][dim] + [2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float g_s = gs[0], b_s = bs[0];
  float ag[LN_MAXE], ab[LN_MAXE];
#pragma unroll
  for (int i = 0; i < LN_MAXE; ++i) ag[i] = ab[i] = 0.f;
  float sg = 0.f, sb = 0.f;
  for (int i = threadIdx.x; i < 2 * dim + 2; i += 256) ln_smem[i] = 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  for (int64_t row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    float g[LN_MAXE], xh[LN_MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        float d[VEC], xv[VEC], gm[VEC], bt[VEC], wv[VEC], dwv[VEC];
        ld_vec<VEC>(dy + row * dim + idx, d);
        ld_vec<VEC>(hl + row * dim + idx, xv);
        ld_vec<VEC>(gamma + idx, gm);
        ld_vec<VEC>(beta + idx, bt);
        ld_vec<VEC>(w + row * dim + idx, wv);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float h = (xv[j] - mu) * rs;
          const float ln = h * gm[j] + bt[j];
          dwv[j] = d[j] * (g_s * ln + b_s);
          sg += d[j] * wv[j] * ln;
          sb += d[j] * wv[j];
          const float dln = d[j] * g_s * wv[j];
          xh[k * VEC + j] = h;
          g[k * VEC + j] = dln * gm[j];
          s1 += g[k * VEC + j];
          s2 += g[k * VEC + j] * h;
          ag[k * VEC + j] += dln * h;
          ab[k * VEC + j] += dln;
        }
        if constexpr (X2) {
          if (dw_acc) {     // every SLN of the network modulates with the same w: its gradient is one running sum (ffvc_sln_bwd_acc2)
            float old[VEC];
            ld_vec<VEC>(dw + row * dim + idx, old);
#pragma unroll
            for (int j = 0; j < VEC; ++j) dwv[j] += old[j];
          }
        }
        st_vec<VEC>(dw + row * dim + idx, dwv);
      }
    }
    s1 = wave_sum(s1) / dim;
    s2 = wave_sum(s2) / dim;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int idx = (k * 64 + lane) * VEC;
      if (idx < dim) {
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = rs * (g[k * VEC + j] - s1 - xh[k * VEC + j] * s2);
        if (dres) {
          float r[VEC];
          ld_vec<VEC>(dres + row * dim + idx, r);
#pragma unroll
          for (int j = 0; j < VEC; ++j) o[j] += r[j];
        }
        st_vec<VEC>(dhl + row * dim + idx, o);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int idx = (k * 64 + lane) * VEC;
    if (idx < dim) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        atomicAdd(&ln_smem[idx + j], ag[k * VEC + j]);
        atomicAdd(&ln_smem[dim + idx + j], ab[k * VEC + j]);
      }
    }
  }
  sg = wave_sum(sg);
  sb = wave_sum(sb);
  if (lane == 0) {
    atomicAdd(&ln_smem[2 * dim], sg);
    atomicAdd(&ln_smem[2 * dim + 1], sb);
  }
  __syncthreads();
  if (acc_mode) {   // part_g / part_b / part_s[0], part_s[1] are the gradients themselves: accumulate with fp32 atomics
    for (int i = threadIdx.x; i < dim; i += 256) {
      atomicAdd(part_g + i, ln_smem[i]);
      atomicAdd(part_b + i, ln_smem[dim + i]);
    }
    if constexpr (X2) {
      if (threadIdx.x < 2) atomicAdd(threadIdx.x ? part_s1 : part_s, ln_smem[2 * dim + threadIdx.x]);
    } else {
      if (threadIdx.x < 2) atomicAdd(part_s + threadIdx.x, ln_smem[2 * dim + threadIdx.x]);
    }
    return;
  }
  for (int i = threadIdx.x; i < dim; i += 256) {
    part_g[(int64_t)blockIdx.x * dim + i] = ln_smem[i];
    part_b[(int64_t)blockIdx.x * dim + i] = ln_smem[dim + i];
  }
  if (threadIdx.x < 2) part_s[(int64_t)blockIdx.x * 2 + threadIdx.x] = ln_smem[2 * dim + threadIdx.x];
}

// ------------------------------ GroupNorm ----------------------------------
// NHWC tensor [B, HW, C], G groups of C/G consecutive channels, eps 1e-6, affine, optional
// fused swish (taming Normalize + nonlinearity).
// Pass 1: per (chunk, image) partial sum / sum-of-squares per group in fp64 -> ws[B][NCH][G][2].
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, double* __restrict__ ws, int HW,
                                                       int C, int G, int rows_per_chunk) {
  constexpr int EPC = ElemTraits<T>::kPerChunk;  // 8 bf16 / 4 f32 channels per thread-load
  // round 5: the per-thread partial sums meet in a FIXED order (one LDS slot per thread, then a serial sum per channel in
  // fp64) instead of through LDS float atomics — the atomics' arrival order changed the fp32 sums in their last bits from
  // run to run, and a deep 16-bit decoder amplifies any such difference to the size of its rounding noise (run-to-run 7e-4
  // rel-rms in the decoded image, 4-6e-2 in the gradients behind the max-pool's argmax: tools/r5/grad_spread.py)
  __shared__ float s_sum[1024], s_sq[1024];
  __shared__ float s_p1[256 * EPC], s_p2[256 * EPC];
  const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  const int cpr = C / EPC;           // 16-B columns per pixel
  const int rpp = 256 / cpr;         // pixels per pass
  const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
  float a1[EPC], a2[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) a1[j] = a2[j] = 0.f;
  const int p0 = chunk * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
  if (rr < rpp) {
    for (int p = p0 + rr; p < p1; p += rpp) {
      const T* px = x + ((int64_t)b * HW + p) * C + cc * EPC;
      if constexpr (EPC == 8) {
        f32x8 v = load8s(px);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          a1[j] += v.v[j];
          a2[j] += v.v[j] * v.v[j];
        }
      } else {
        f32x4_t v = load4(px);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a1[j] += v[j];
          a2[j] += v[j] * v[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      s_p1[threadIdx.x * EPC + j] = a1[j];
      s_p2[threadIdx.x * EPC + j] = a2[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {       // channel c: its rpp row-threads, in row order
    const int ccx = c / EPC, j = c - ccx * EPC;
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < rpp; ++r) {
      t1 += (double)s_p1[(r * cpr + ccx) * EPC + j];
      t2 += (double)s_p2[(r * cpr + ccx) * EPC + j];
    }
    s_sum[c] = (float)t1;
    s_sq[c] = (float)t2;
  }
  __syncthreads();
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += 256) {
    double t1 = 0.0, t2 = 0.0;
    for (int c = 0; c < cpg; ++c) {
      t1 += (double)s_sum[g * cpg + c];
      t2 += (double)s_sq[g * cpg + c];
    }
    double* o = ws + (((int64_t)b * nch + chunk) * G + g) * 2;
    o[0] = t1;
    o[1] = t2;
  }
}

// Combine the chunk partials of image b (every block does it redundantly: NCH <= 64, G = 32).
__device__ __forceinline__ void gn_finalize(const double* ws, int b, int nch, int G, int64_t n, float eps,
                                            float* s_mean, float* s_rstd) {
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    double t1 = 0.0, t2 = 0.0;
    for (int c = 0; c < nch; ++c) {
      const double* o = ws + (((int64_t)b * nch + c) * G + g) * 2;
      t1 += o[0];
      t2 += o[1];
    }
    const double mu = t1 / (double)n;
    double var = t2 / (double)n - mu * mu;
    if (var < 0.0) var = 0.0;
    s_mean[g] = (float)mu;
    s_rstd[g] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const double* __restrict__ ws,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int HW,
                                                       int C, int G, int nch, float eps, int swish,
                                                       int rows_per_block, uint8_t* __restrict__ y8 = nullptr,
                                                       float* __restrict__ f8_state = nullptr, int f8_fmt = 0) {
  // y8 (16-bit T): the output also leaves as fp8 bytes (see ln_fwd_kernel); y may then be NULL — the operand of an fp8 convolution
  // straight from the normalisation pass, 3 bytes per element over HBM instead of 4 + 3
  constexpr int EPC = ElemTraits<T>::kPerChunk;
  __shared__ float s_mean[64], s_rstd[64];
  const int b = blockIdx.y;
  const int cpg = C / G;
  gn_finalize(ws, b, nch, G, (int64_t)HW * cpg, eps, s_mean, s_rstd);
  __syncthreads();
  if (blockIdx.x == 0)
    for (int g = threadIdx.x; g < G; g += 256) {
      mean[b * G + g] = s_mean[g];
      rstd[b * G + g] = s_rstd[g];
    }
  const int cpr = C / EPC, rpp = 256 / cpr;
  const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
  float sc[EPC], sh[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    const int c = (cc * EPC + j) % C, g = c / cpg;
    sc[j] = gamma[c] * s_rstd[g];
    sh[j] = beta[c] - s_mean[g] * sc[j];
  }
  const float f8s = y8 ? f8_state[0] : 0.0f;
  float f8m = 0.0f;
  const int p0 = blockIdx.x * rows_per_block, p1 = rr < rpp ? min(HW, p0 + rows_per_block) : 0;
  for (int p = p0 + rr; p < p1; p += rpp) {
    const int64_t off = ((int64_t)b * HW + p) * C + cc * EPC;
    if constexpr (EPC == 8) {
      f32x8 v = load8s(x + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float u = v.v[j] * sc[j] + sh[j];
        v.v[j] = swish ? act_swish_t<T>(u) : u;
      }
      if (y) store8s(y + off, v);
      if (y8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v.v[j] = lo_round<T>(v.v[j]);
          f8m = fmaxf(f8m, fabsf(v.v[j]));
        }
        u32x2_t o;
        o[0] = f8_pack4(f8_fmt, v.v[0] * f8s, v.v[1] * f8s, v.v[2] * f8s, v.v[3] * f8s);
        o[1] = f8_pack4(f8_fmt, v.v[4] * f8s, v.v[5] * f8s, v.v[6] * f8s, v.v[7] * f8s);
        __builtin_nontemporal_store(o, (u32x2_t*)(y8 + off));
      }
    } else {
      f32x4_t v = load4(x + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = v[j] * sc[j] + sh[j];
        v[j] = swish ? act_swish_t<T>(u) : u;
      }
      store4(y + off, v);
    }
  }
  if (y8) f8_amax_block(f8m, f8_state);
}

// Backward pass 1: per group S1 = sum dxh, S2 = sum dxh * xh, with dxh = dy * swish'(u) * gamma.
// ACC > 0: more workgroups than partial slots (image groups of a few images, see ffvc_groupnorm_bwd): workgroup `chunk` ADDS its totals
// into slot chunk % ACC of a zeroed ws (fp64 atomics) instead of owning a slot.
template <typename T, bool ACCUM = false>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, double* __restrict__ ws,
                                                           int HW, int C, int G, int swish, int rows_per_chunk, int nslots = 0) {
  constexpr int EPC = ElemTraits<T>::kPerChunk;
  __shared__ float s_1[1024], s_2[1024];
  __shared__ float s_p1[256 * EPC], s_p2[256 * EPC];      // fixed-order combine of the row-threads (see gn_stats_kernel)
  const int b = blockIdx.y, chunk = blockIdx.x, nch = ACCUM ? nslots : gridDim.x;
  const int cpg = C / G;
  const int cpr = C / EPC, rpp = 256 / cpr;
  const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
  if (rr < rpp) {
    float gm[EPC], bt[EPC], mu[EPC], rs[EPC], a1[EPC], a2[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      const int c = cc * EPC + j, g = c / cpg;
      gm[j] = gamma[c];
      bt[j] = beta[c];
      mu[j] = mean[b * G + g];
      rs[j] = rstd[b * G + g];
      a1[j] = a2[j] = 0.f;
    }
    const int p0 = chunk * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
    for (int p = p0 + rr; p < p1; p += rpp) {
      const int64_t off = ((int64_t)b * HW + p) * C + cc * EPC;
      float dv[EPC], xv[EPC];
      if constexpr (EPC == 8) {
        f32x8 a = load8s(dy + off), c2 = load8s(x + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          dv[j] = a.v[j];
          xv[j] = c2.v[j];
        }
      } else {
        f32x4_t a = load4(dy + off), c2 = load4(x + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dv[j] = a[j];
          xv[j] = c2[j];
        }
      }
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        const float xh = (xv[j] - mu[j]) * rs[j];
        float d = dv[j];
        if (swish) d *= act_swish_grad_t<T>(xh * gm[j] + bt[j]);
        d *= gm[j];
        a1[j] += d;
        a2[j] += d * xh;
      }
    }
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      s_p1[threadIdx.x * EPC + j] = a1[j];
      s_p2[threadIdx.x * EPC + j] = a2[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int ccx = c / EPC, j = c - ccx * EPC;
    double t1 = 0.0, t2 = 0.0;
    for (int r = 0; r < rpp; ++r) {
      t1 += (double)s_p1[(r * cpr + ccx) * EPC + j];
      t2 += (double)s_p2[(r * cpr + ccx) * EPC + j];
    }
    s_1[c] = (float)t1;
    s_2[c] = (float)t2;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    double t1 = 0.0, t2 = 0.0;
    for (int c = 0; c < cpg; ++c) {
      t1 += (double)s_1[g * cpg + c];
      t2 += (double)s_2[g * cpg + c];
    }
    if constexpr (ACCUM) {
      double* o = ws + (((int64_t)b * nch + (chunk % nch)) * G + g) * 2;
      atomicAdd(o, t1);
      atomicAdd(o + 1, t2);
    } else {
      double* o = ws + (((int64_t)b * nch + chunk) * G + g) * 2;
      o[0] = t1;
      o[1] = t2;
    }
  }
}

// Backward pass 2: dx = rstd * (dxh - S1/n - xh * S2/n) (+ dres).
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const double* __restrict__ ws, const T* __restrict__ dres,
                                                           T* __restrict__ dx, int HW, int C, int G, int nch,
                                                           int swish, int rows_per_block, uint8_t* __restrict__ dx8 = nullptr,
                                                           float* __restrict__ f8_state = nullptr, int f8_fmt = 1) {
  // dx8 (16-bit T): the gradient also leaves as fp8 bytes for the producing convolution's dgrad (see gn_apply_kernel); dx may be NULL
  constexpr int EPC = ElemTraits<T>::kPerChunk;
  __shared__ float s_1[64], s_2[64];
  const int b = blockIdx.y;
  const int cpg = C / G;
  const double n = (double)HW * cpg;
  for (int g = threadIdx.x; g < G; g += 256) {
    double t1 = 0.0, t2 = 0.0;
    for (int c = 0; c < nch; ++c) {
      const double* o = ws + (((int64_t)b * nch + c) * G + g) * 2;
      t1 += o[0];
      t2 += o[1];
    }
    s_1[g] = (float)(t1 / n);
    s_2[g] = (float)(t2 / n);
  }
  __syncthreads();
  const int cpr = C / EPC, rpp = 256 / cpr;
  const int cc = threadIdx.x % cpr, rr = threadIdx.x / cpr;
  float gm[EPC], bt[EPC], mu[EPC], rs[EPC], m1[EPC], m2[EPC];
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    const int c = (cc * EPC + j) % C, g = c / cpg;
    gm[j] = gamma[c];
    bt[j] = beta[c];
    mu[j] = mean[b * G + g];
    rs[j] = rstd[b * G + g];
    m1[j] = s_1[g];
    m2[j] = s_2[g];
  }
  const float f8s = dx8 ? f8_state[0] : 0.0f;
  float f8m = 0.0f;
  const int p0 = blockIdx.x * rows_per_block, p1 = rr < rpp ? min(HW, p0 + rows_per_block) : 0;
  for (int p = p0 + rr; p < p1; p += rpp) {
    const int64_t off = ((int64_t)b * HW + p) * C + cc * EPC;
    float dv[EPC], xv[EPC], rv[EPC];
    if constexpr (EPC == 8) {
      f32x8 a = load8s(dy + off), c2 = load8s(x + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dv[j] = a.v[j];
        xv[j] = c2.v[j];
        rv[j] = 0.f;
      }
      if (dres) {
        f32x8 r = load8s(dres + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) rv[j] = r.v[j];
      }
    } else {
      f32x4_t a = load4(dy + off), c2 = load4(x + off);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        dv[j] = a[j];
        xv[j] = c2[j];
        rv[j] = 0.f;
      }
      if (dres) {
        f32x4_t r = load4(dres + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) rv[j] = r[j];
      }
    }
    float o[EPC];
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      const float xh = (xv[j] - mu[j]) * rs[j];
      float d = dv[j];
      if (swish) d *= act_swish_grad_t<T>(xh * gm[j] + bt[j]);
      d *= gm[j];
      o[j] = rs[j] * (d - m1[j] - xh * m2[j]) + rv[j];
    }
    if constexpr (EPC == 8) {
      f32x8 w;
#pragma unroll
      for (int j = 0; j < 8; ++j) w.v[j] = o[j];
      if (dx) store8s(dx + off, w);
      if (dx8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          o[j] = lo_round<T>(o[j]);
          f8m = fmaxf(f8m, fabsf(o[j]));
        }
        u32x2_t q;
        q[0] = f8_pack4(f8_fmt, o[0] * f8s, o[1] * f8s, o[2] * f8s, o[3] * f8s);
        q[1] = f8_pack4(f8_fmt, o[4] * f8s, o[5] * f8s, o[6] * f8s, o[7] * f8s);
        __builtin_nontemporal_store(q, (u32x2_t*)(dx8 + off));
      }
    } else {
      f32x4_t w = {o[0], o[1], o[2], o[3]};
      store4(dx + off, w);
    }
  }
  if (dx8) f8_amax_block(f8m, f8_state);
}

// Backward in ONE pass over HBM (round 4): 2 reads (dy, x) + 1 write (dx) (+ dres) instead of the 4 + 1 of the two kernels above.
// A group of `wg_per_img` workgroups owns one image at a time; every workgroup keeps its share of dy and x — NL 16-byte chunks of
// each per thread, 128 KiB per workgroup, two workgroups per CU: the 33.5 MB of a 256 x 256 x 128 image fit the register files of
// the chip — while the group meets at a counter to learn the image's sums:
//   load (dy, x) -> dxh = dy * swish'(u) * gamma and xh, partial sums (LDS, then one fp64 atomic pair per group and workgroup)
//   -> arrive / poll on cnt[image] -> dx = rstd * (dxh - S1/n - xh * S2/n) (+ dres) from the registers.
// dxh and xh are parked in the registers of dy / x in the 16-bit storage type (dx is stored in that type anyway).
// Inter-workgroup hand-off (MI355X_MICROARCH.md, "8-byte agent atomics both sides"): the sums travel as fp64 atomic adds and are
// read back with 8-byte agent-scope atomic loads, the counter is an agent-scope atomic; `s_waitcnt vmcnt(0)` + barrier between the
// sums and the counter.  All workgroups of the launch are co-resident (the launcher sizes the grid from the occupancy query), so
// the poll cannot deadlock; with several groups in flight (and two workgroups per CU) one group's wait is another's streaming.
constexpr int GNF_NL = 16;
constexpr int GNF_REP = 32, GNF_CNT = 16;      // replicas of the per-image sums / sub-counters of the arrival count
template <typename T>
__global__ __launch_bounds__(256, 2) void gn_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const T* __restrict__ dres, T* __restrict__ dx,
                                                              double* __restrict__ sums, unsigned* __restrict__ cnt, int B,
                                                              int HW, int C, int G, int swish, int wg_per_img, int ngroups) {
  static_assert(sizeof(T) == 2, "16-bit storage types only");
  __shared__ float s_1[1024], s_2[1024];
  const int grp = blockIdx.x / wg_per_img, idx = blockIdx.x - grp * wg_per_img;
  if (grp >= ngroups) return;
  const int tid = threadIdx.x;
  const int cpg = C / G, cpr = C / 8, rpp = 256 / cpr;
  const int cc = tid % cpr, rr = tid / cpr;
  const int px_per_wg = GNF_NL * rpp;
  const int p0 = idx * px_per_wg;
  const double inv_n = 1.0 / ((double)HW * cpg);
  float gm[8], bt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    gm[j] = gamma[cc * 8 + j];
    bt[j] = beta[cc * 8 + j];
  }
  for (int b = grp; b < B; b += ngroups) {
    float rs[8], murs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (cc * 8 + j) / cpg;
      rs[j] = rstd[b * G + g];
      murs[j] = mean[b * G + g] * rs[j];
    }
    u32x4_t xr[GNF_NL], dr[GNF_NL];
    const int64_t base = ((int64_t)b * HW + p0 + rr) * C + cc * 8;
#pragma unroll
    for (int k = 0; k < GNF_NL; ++k) {
      const bool ok = p0 + rr + k * rpp < HW;
      const int64_t off = base + (int64_t)k * rpp * C;
      xr[k] = ok ? __builtin_nontemporal_load((const u32x4_t*)(x + off)) : u32x4_t{0, 0, 0, 0};
      dr[k] = ok ? __builtin_nontemporal_load((const u32x4_t*)(dy + off)) : u32x4_t{0, 0, 0, 0};
    }
    for (int i = tid; i < C; i += 256) s_1[i] = s_2[i] = 0.f;
    float a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll
    for (int k = 0; k < GNF_NL; ++k) {
      if (p0 + rr + k * rpp < HW) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float xv[2], dv[2];
          if constexpr (std::is_same<T, f16_t>::value) {
            const f32pair_t a = unpack_f16x2(xr[k][i]), c2 = unpack_f16x2(dr[k][i]);
            xv[0] = a[0]; xv[1] = a[1]; dv[0] = c2[0]; dv[1] = c2[1];
          } else {
            xv[0] = __uint_as_float(xr[k][i] << 16); xv[1] = __uint_as_float(xr[k][i] & 0xffff0000u);
            dv[0] = __uint_as_float(dr[k][i] << 16); dv[1] = __uint_as_float(dr[k][i] & 0xffff0000u);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = 2 * i + e;
            const float xh = xv[e] * rs[j] - murs[j];
            float d = dv[e];
            if (swish) d *= act_swish_grad_t<T>(xh * gm[j] + bt[j]);
            d *= gm[j];
            a1[j] += d;
            a2[j] += d * xh;
            xv[e] = xh;
            dv[e] = d;
          }
          xr[k][i] = lo_pack2<T>(xv[0], xv[1]);
          dr[k][i] = lo_pack2<T>(dv[0], dv[1]);
        }
      }
    }
    __syncthreads();                      // s_1 / s_2 zeroed
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      atomicAdd(&s_1[cc * 8 + j], a1[j]);
      atomicAdd(&s_2[cc * 8 + j], a2[j]);
    }
    __syncthreads();
    // same-address atomics from hundreds of workgroups serialise at the memory side (measured: the first version, one slot per
    // group and one counter per image, ran 1564 us against the two-pass form's 1165): the sums are spread over GNF_REP replicas
    // (workgroup idx adds into replica idx % GNF_REP), the arrivals over GNF_CNT sub-counters
    if (tid < G) {
      double t1 = 0.0, t2 = 0.0;
      for (int c = 0; c < cpg; ++c) {
        t1 += (double)s_1[tid * cpg + c];
        t2 += (double)s_2[tid * cpg + c];
      }
      double* sp = sums + (((int64_t)b * GNF_REP + (idx % GNF_REP)) * G + tid) * 2;
      atomicAdd(sp, t1);
      atomicAdd(sp + 1, t2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's contributions have left the CU ...
    __syncthreads();
    if (tid < 64) {                                          // ... before it is counted
      unsigned* cb = cnt + (int64_t)b * GNF_CNT;
      if (tid == 0) atomicAdd(cb + (idx % GNF_CNT), 1u);
      for (;;) {
        unsigned tot = tid < GNF_CNT ? __hip_atomic_load(cb + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int o = 1; o < GNF_CNT; o <<= 1) tot += __shfl_xor(tot, o, 64);
        if (__shfl(tot, 0, 64) >= (unsigned)wg_per_img) break;
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();
    // totals: thread t < 2 G folds the replicas of one sum into LDS (s_1 is free again)
    if (tid < 2 * G) {
      const unsigned long long* sp = (const unsigned long long*)(sums + (int64_t)b * GNF_REP * G * 2) + tid;
      double t = 0.0;
      const int nrep = wg_per_img < GNF_REP ? wg_per_img : GNF_REP;
      for (int r = 0; r < nrep; ++r)
        t += __longlong_as_double((long long)__hip_atomic_load(sp + (int64_t)r * G * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      s_1[tid] = (float)(t * inv_n);
    }
    __syncthreads();
    float m1[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (cc * 8 + j) / cpg;
      m1[j] = s_1[2 * g];
      m2[j] = s_1[2 * g + 1];
    }
#pragma unroll
    for (int k = 0; k < GNF_NL; ++k) {
      if (p0 + rr + k * rpp < HW) {
        const int64_t off = base + (int64_t)k * rpp * C;
        u32x4_t rv = {0, 0, 0, 0};
        if (dres) rv = __builtin_nontemporal_load((const u32x4_t*)(dres + off));
        u32x4_t o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float xh[2], d[2], r[2];
          if constexpr (std::is_same<T, f16_t>::value) {
            const f32pair_t a = unpack_f16x2(xr[k][i]), c2 = unpack_f16x2(dr[k][i]), c3 = unpack_f16x2(rv[i]);
            xh[0] = a[0]; xh[1] = a[1]; d[0] = c2[0]; d[1] = c2[1]; r[0] = c3[0]; r[1] = c3[1];
          } else {
            xh[0] = __uint_as_float(xr[k][i] << 16); xh[1] = __uint_as_float(xr[k][i] & 0xffff0000u);
            d[0] = __uint_as_float(dr[k][i] << 16); d[1] = __uint_as_float(dr[k][i] & 0xffff0000u);
            r[0] = __uint_as_float(rv[i] << 16); r[1] = __uint_as_float(rv[i] & 0xffff0000u);
          }
          float w[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int j = 2 * i + e;
            w[e] = rs[j] * (d[e] - m1[j] - xh[e] * m2[j]) + r[e];
          }
          o[i] = lo_pack2<T>(w[0], w[1]);
        }
        __builtin_nontemporal_store(o, (u32x4_t*)(dx + off));
      }
    }
    __syncthreads();                      // s_1 / s_2 are re-zeroed by the next image
  }
}

// ------------------------------- softmax -----------------------------------
// p[r, :cols] = softmax(scale * s[r, :cols] (+ causal mask)), columns [cols, ldp) zero-filled.
// Causal: key j is visible to query i = r % q_len iff j <= i (cloob.py:510-516; x-transformers Decoder).
constexpr int SM_MAXE = 16;  // cols <= 1024
template <typename PT>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ s, PT* __restrict__ p,
                                                          int64_t rows, int cols, int lds, int ldp, float scale,
                                                          int causal, int q_len) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const int limit = causal ? min(cols, (int)(row % q_len) + 1) : cols;
    float v[SM_MAXE];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_MAXE; ++k) {
      const int j = k * 64 + lane;
      v[k] = (j < limit) ? s[row * lds + j] * scale : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAXE; ++k) {
      const int j = k * 64 + lane;
      v[k] = (j < limit) ? __expf(v[k] - mx) : 0.f;
      sum += v[k];
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int k = 0; k < SM_MAXE; ++k) {
      const int j = k * 64 + lane;
      if (j < ldp) ElemTraits<PT>::store(p + row * ldp + j, v[k] * inv);
    }
  }
}

// ds = scale * p * (dp - sum_j p_j dp_j); columns [cols, ldp) zero-filled.
template <typename PT>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const PT* __restrict__ p, const float* __restrict__ dp,
                                                          PT* __restrict__ ds, int64_t rows, int cols, int ldp,
                                                          int lddp, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    float pv[SM_MAXE], dv[SM_MAXE];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAXE; ++k) {
      const int j = k * 64 + lane;
      pv[k] = (j < cols) ? ElemTraits<PT>::load(p + row * ldp + j) : 0.f;
      dv[k] = (j < cols) ? dp[row * lddp + j] : 0.f;
      dot += pv[k] * dv[k];
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < SM_MAXE; ++k) {
      const int j = k * 64 + lane;
      if (j < ldp) ElemTraits<PT>::store(ds + row * ldp + j, scale * pv[k] * (dv[k] - dot));
    }
  }
}

inline int grid_rows(int64_t rows) {
  int64_t g = (rows + 3) / 4;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
// LayerNorm forward: the kernel can pipeline several rows per wave (FFVC_LN_FWD_GRID = workgroups of 4 waves).  Measured (round 6,
// profiles/r06_ln_fwd.txt, 16384 x 1024 fp32 -> f16): 24.6 us with one row per wave (4096 workgroups), 23.7 / 25.9 / 27.1 / 44.9 us at
// 2048 / 1024 / 512 / 256, and the step 0.8 ms SLOWER at 1024 than at 4096 — at ~4 TB/s the kernel is not waiting on its own phases,
// so the default stays one row per wave.
inline int grid_rows_fwd(int64_t rows) {
  static int cap = -1;
  if (cap < 0) {
    const char* e = getenv("FFVC_LN_FWD_GRID");
    cap = e ? atoi(e) : 4096;
    if (cap < 1) cap = 1;
  }
  int64_t g = (rows + 3) / 4;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int ffvc_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y,
                                  int y_dtype, float* mean, float* rstd, int64_t rows, int dim, float eps,
                                  void* stream) {
  FFVC_CHECK_ARG(x && gamma && beta && y && mean && rstd, "ffvc_layernorm_fwd: null pointer");
  FFVC_CHECK_ARG(rows > 0 && dim > 0 && dim <= 64 * LN_MAXE, "ffvc_layernorm_fwd: dim=%d unsupported (max %d)", dim,
                 64 * LN_MAXE);
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_rows_fwd(rows);
  const bool v4 = (dim % 4) == 0;
  DISPATCH_DT(x_dtype, XT, DISPATCH_DT(y_dtype, YT, {
                if (v4)
                  hipLaunchKernelGGL((ln_fwd_kernel<4, XT, YT>), dim3(grid), dim3(256), 0, st, (const XT*)x, gamma,
                                     beta, (YT*)y, mean, rstd, rows, dim, eps);
                else
                  hipLaunchKernelGGL((ln_fwd_kernel<1, XT, YT>), dim3(grid), dim3(256), 0, st, (const XT*)x, gamma,
                                     beta, (YT*)y, mean, rstd, rows, dim, eps);
              }));
  FFVC_LAUNCH_CHECK();
  return 0;
}

// LayerNorm whose output (also) leaves as fp8 bytes for the fp8 GEMM that consumes it: y8 = saturate(round_16(y) * f8_state[0]), f8_state[1]
// collects max |round_16(y)| — byte for byte what ffvc_fp8_quant makes of y (delayed per-tensor scaling).  y may be NULL (fp8 only).
extern "C" int ffvc_layernorm_fwd_f8(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype, void* y8,
                                     float* f8_state, int f8_fmt, float* mean, float* rstd, int64_t rows, int dim, float eps,
                                     void* stream) {
  FFVC_CHECK_ARG(x && gamma && beta && y8 && f8_state && mean && rstd, "ffvc_layernorm_fwd_f8: null pointer");
  FFVC_CHECK_ARG(rows > 0 && dim > 0 && dim <= 64 * LN_MAXE && dim % 4 == 0, "ffvc_layernorm_fwd_f8: dim=%d unsupported (multiple of 4, max %d)",
                 dim, 64 * LN_MAXE);
  FFVC_CHECK_ARG(y_dtype == FFVC_F16 || y_dtype == FFVC_BF16, "ffvc_layernorm_fwd_f8: the rounding type of y must be f16 or bf16");
  FFVC_CHECK_ARG(f8_fmt == 0 || f8_fmt == 1, "ffvc_layernorm_fwd_f8: f8_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(((uintptr_t)y8 % 4) == 0, "ffvc_layernorm_fwd_f8: misaligned y8");
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_rows_fwd(rows);
  DISPATCH_DT(x_dtype, XT, {
    if (y_dtype == FFVC_F16)
      hipLaunchKernelGGL((ln_fwd_kernel<4, XT, f16_t>), dim3(grid), dim3(256), 0, st, (const XT*)x, gamma, beta, (f16_t*)y, mean, rstd, rows,
                         dim, eps, (uint8_t*)y8, f8_state, f8_fmt);
    else
      hipLaunchKernelGGL((ln_fwd_kernel<4, XT, uint16_t>), dim3(grid), dim3(256), 0, st, (const XT*)x, gamma, beta, (uint16_t*)y, mean, rstd,
                         rows, dim, eps, (uint8_t*)y8, f8_state, f8_fmt);
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

// Rows per workgroup (4 waves, one row per wave and trip) of the LayerNorm / SLN backward kernels.  32 suits the 16384-row
// streams of the Mixer; with a few hundred rows (VitGAN at 32 samples per GPU: 512) that leaves 16 workgroups on 256 CUs walking
// 8 rows per wave one after the other (55 us for a 2 MB tensor), so small problems get as few rows per workgroup as it takes to
// put ~256 of them on the chip (never fewer than 4: one per wave).  FFVC_LN_RPB overrides the 32.
static int ln_rows_per_block(int64_t rows) {
  static int rpb = -1;
  if (rpb < 0) {
    const char* e = getenv("FFVC_LN_RPB");
    rpb = e ? atoi(e) : 32;
    if (rpb < 4) rpb = 4;
  }
  int r = rpb;
  while (r > 4 && (rows + r - 1) / r < 256) r >>= 1;
  return r < 4 ? 4 : r;
}

extern "C" int ffvc_layernorm_bwd_blocks(int64_t rows) {
  // number of partial rows ffvc_layernorm_bwd writes into part_g / part_b
  const int rpb = ln_rows_per_block(rows);
  int64_t nb = (rows + rpb - 1) / rpb;
  return (int)(nb < 1 ? 1 : nb);
}

static int ln_bwd_launch(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                         const float* rstd, const void* dres, void* dx, float* part_g, float* part_b, int64_t rows,
                         int dim, void* stream, int acc_mode, void* dx_lo);

extern "C" int ffvc_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                                  const float* mean, const float* rstd, const void* dres, void* dx, float* part_g,
                                  float* part_b, void* dx_lo, int64_t rows, int dim, void* stream) {
  return ln_bwd_launch(dy, dy_dtype, x, x_dtype, gamma, mean, rstd, dres, dx, part_g, part_b, rows, dim, stream, 0, dx_lo);
}

// Frozen-layer LayerNorm backward on the fp32 residual stream with producer-side quantisation (see ln_bwd_f8_kernel): dx fp32 [rows, dim]
// (+ dres) and dx8 = the fp8 bytes ffvc_fp8_quant would make of round_{dy_dtype}(dx).  dim % 4 == 0, dim <= 2048, dy 16-bit.
extern "C" int ffvc_layernorm_bwd_f8(const void* dy, int dy_dtype, const float* x, const float* gamma, const float* mean, const float* rstd,
                                     const float* dres, float* dx, void* dx8, float* f8_state, int f8_fmt, int64_t rows, int dim,
                                     void* stream) {
  FFVC_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dx8 && f8_state, "ffvc_layernorm_bwd_f8: null pointer");
  FFVC_CHECK_ARG(dy_dtype == FFVC_F16 || dy_dtype == FFVC_BF16, "ffvc_layernorm_bwd_f8: dy must be f16 or bf16");
  FFVC_CHECK_ARG(rows > 0 && dim > 0 && dim % 4 == 0 && dim <= 64 * LN_MAXE, "ffvc_layernorm_bwd_f8: dim=%d unsupported", dim);
  FFVC_CHECK_ARG(f8_fmt == 0 || f8_fmt == 1, "ffvc_layernorm_bwd_f8: f8_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(((uintptr_t)dx8 % 4) == 0, "ffvc_layernorm_bwd_f8: misaligned dx8");
  hipStream_t st = (hipStream_t)stream;
  int64_t gq = (rows + 3) / 4;
  const int grid = (int)(gq < 1 ? 1 : (gq > 2048 ? 2048 : gq));
  if (dy_dtype == FFVC_F16) {
    if (dim <= 1024)
      hipLaunchKernelGGL((ln_bwd_f8_kernel<f16_t, 16>), dim3(grid), dim3(256), 0, st, (const f16_t*)dy, x, gamma, mean, rstd, dres, dx,
                         (uint8_t*)dx8, f8_state, f8_fmt, rows, dim);
    else
      hipLaunchKernelGGL((ln_bwd_f8_kernel<f16_t, 32>), dim3(grid), dim3(256), 0, st, (const f16_t*)dy, x, gamma, mean, rstd, dres, dx,
                         (uint8_t*)dx8, f8_state, f8_fmt, rows, dim);
  } else {
    if (dim <= 1024)
      hipLaunchKernelGGL((ln_bwd_f8_kernel<uint16_t, 16>), dim3(grid), dim3(256), 0, st, (const uint16_t*)dy, x, gamma, mean, rstd, dres, dx,
                         (uint8_t*)dx8, f8_state, f8_fmt, rows, dim);
    else
      hipLaunchKernelGGL((ln_bwd_f8_kernel<uint16_t, 32>), dim3(grid), dim3(256), 0, st, (const uint16_t*)dy, x, gamma, mean, rstd, dres, dx,
                         (uint8_t*)dx8, f8_state, f8_fmt, rows, dim);
  }
  FFVC_LAUNCH_CHECK();
  return 0;
}

// Per-stream scratch for the parameter-gradient partials of the accumulate form: [workgroups][2 * dim] fp32.  Launches on one
// stream run in order, so one buffer per stream is enough; it grows on demand (the old one is freed after a stream sync).
static float* ln_scratch(hipStream_t st, size_t bytes) {
  static std::mutex mu;
  static std::map<hipStream_t, std::pair<float*, size_t>> pool;
  std::lock_guard<std::mutex> lk(mu);
  auto& slot = pool[st];
  if (slot.second < bytes) {
    if (slot.first) {
      (void)hipStreamSynchronize(st);
      (void)hipFree(slot.first);
      slot = {nullptr, 0};
    }
    const size_t want = bytes < (8u << 20) ? (8u << 20) : bytes;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    slot = {(float*)p, want};
  }
  return slot.first;
}

extern "C" int ffvc_colsum(const void* x, int dtype, float* out, int64_t rows, int cols, int64_t ld, int accumulate, void* stream);

extern "C" int ffvc_layernorm_bwd_acc(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma,
                                      const float* mean, const float* rstd, const void* dres, void* dx, float* dgamma,
                                      float* dbeta, void* dx_lo, int64_t rows, int dim, void* stream) {
  FFVC_CHECK_ARG(dgamma && dbeta, "ffvc_layernorm_bwd_acc: null gradient pointer");
  // Same-address fp32 atomics serialise in L2 (~75 ns each): with hundreds of workgroups adding into the same [dim] gradient
  // the tail of the kernel is that queue.  The alternative built in round 3 — one partial row per workgroup with plain stores
  // and a column-sum launch that folds them (FFVC_LN_ATOMIC=0) — measured WORSE: 82 vs 75 us isolated at 16384 x 1024, and
  // 17.6 vs 13.7 ms per cfg2 step for the 65 launches (profiles/r03_layernorm_bwd_ab.txt): the parameter-gradient cost is the
  // in-kernel accumulation (32 more live registers, an LDS combine per workgroup), not the global atomics, and the extra launch
  // boundary costs more than the queue.  Default stays the in-kernel atomics.
  static int atomic_mode = -1;
  if (atomic_mode < 0) {
    const char* e = getenv("FFVC_LN_ATOMIC");
    atomic_mode = e ? atoi(e) : 1;
  }
  const int nblk = ffvc_layernorm_bwd_blocks(rows);
  if (!atomic_mode && nblk >= 64) {
    float* scratch = ln_scratch((hipStream_t)stream, (size_t)nblk * 2 * dim * sizeof(float));
    if (scratch) {
      int e = ln_bwd_launch(dy, dy_dtype, x, x_dtype, gamma, mean, rstd, dres, dx, scratch, scratch + dim, rows, dim, stream,
                            2, dx_lo);
      if (e) return e;
      if (dbeta == dgamma + dim) return ffvc_colsum(scratch, FFVC_F32, dgamma, nblk, 2 * dim, 2 * (int64_t)dim, 1, stream);
      e = ffvc_colsum(scratch, FFVC_F32, dgamma, nblk, dim, 2 * (int64_t)dim, 1, stream);
      if (e) return e;
      return ffvc_colsum(scratch + dim, FFVC_F32, dbeta, nblk, dim, 2 * (int64_t)dim, 1, stream);
    }
  }
  return ln_bwd_launch(dy, dy_dtype, x, x_dtype, gamma, mean, rstd, dres, dx, dgamma, dbeta, rows, dim, stream, 1, dx_lo);
}

static int ln_bwd_launch(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* gamma, const float* mean,
                         const float* rstd, const void* dres, void* dx, float* part_g, float* part_b, int64_t rows,
                         int dim, void* stream, int acc_mode, void* dx_lo) {
  FFVC_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "ffvc_layernorm_bwd: null pointer");
  FFVC_CHECK_ARG(!dx_lo || (x_dtype == FFVC_F32 && dy_dtype != FFVC_F32),
                 "ffvc_layernorm_bwd: dx_lo (copy in dy's 16-bit dtype) only next to an fp32 dx");
  FFVC_CHECK_ARG(rows > 0 && dim > 0 && dim <= 64 * LN_MAXE, "ffvc_layernorm_bwd: dim=%d unsupported", dim);
  FFVC_CHECK_ARG((part_g == nullptr) == (part_b == nullptr), "ffvc_layernorm_bwd: need both partial buffers or none");
  hipStream_t st = (hipStream_t)stream;
  const int rpb = ln_rows_per_block(rows);
  const int grid = ffvc_layernorm_bwd_blocks(rows);
  static int plain_opt = -1;
  if (plain_opt < 0) {
    const char* e = getenv("FFVC_LN_PLAIN");
    plain_opt = e ? atoi(e) : 1;
  }
  const int plain = (plain_opt && part_g) ? 1 : 0;           // per-wave LDS rows + plain stores instead of LDS float atomics
  const size_t smem = part_g ? (plain ? 4 : 1) * 2 * (size_t)dim * sizeof(float) : 0;
  const bool v4 = (dim % 4) == 0;
  const int64_t pstride = acc_mode == 2 ? 2 * (int64_t)dim : (int64_t)dim;   // 2: [blocks][dgamma row | dbeta row]
  if (acc_mode == 2) acc_mode = 0;
  static int wide = -1;
  if (wide < 0) {
    const char* e = getenv("FFVC_LN_WIDE");
    wide = e ? atoi(e) : 0;     // measured: 16384x1024 74.8 us (4 waves) vs 89.9 us (16 waves) isolated, step 141.6 vs 142-143 ms: off
  }
  if (wide && acc_mode && part_g && v4 && dim <= 1024 && rows >= 8192) {
    // accumulate-into-the-bucket mode on a large launch: 16-wave workgroups, 64 rows each (see ln_bwd_kernel)
    const int rpb16 = 64;
    const int grid16 = (int)((rows + rpb16 - 1) / rpb16);
    DISPATCH_DT(dy_dtype, DYT, DISPATCH_DT(x_dtype, XT, {
                  hipLaunchKernelGGL((ln_bwd_kernel<4, DYT, XT, 16, 16>), dim3(grid16), dim3(1024), smem, st, (const DYT*)dy,
                                     (const XT*)x, gamma, mean, rstd, (const XT*)dres, (XT*)dx, part_g, part_b, rows, dim,
                                     rpb16, acc_mode, (DYT*)dx_lo, (int64_t)dim, 0);
                }));
    FFVC_LAUNCH_CHECK();
    return 0;
  }
  DISPATCH_DT(dy_dtype, DYT, DISPATCH_DT(x_dtype, XT, {
                if (v4 && dim <= 1024)
                  hipLaunchKernelGGL((ln_bwd_kernel<4, DYT, XT, 16>), dim3(grid), dim3(256), smem, st, (const DYT*)dy,
                                     (const XT*)x, gamma, mean, rstd, (const XT*)dres, (XT*)dx, part_g, part_b, rows,
                                     dim, rpb, acc_mode, (DYT*)dx_lo, pstride, plain);
                else if (v4)
                  hipLaunchKernelGGL((ln_bwd_kernel<4, DYT, XT>), dim3(grid), dim3(256), smem, st, (const DYT*)dy,
                                     (const XT*)x, gamma, mean, rstd, (const XT*)dres, (XT*)dx, part_g, part_b, rows,
                                     dim, rpb, acc_mode, (DYT*)dx_lo, pstride, plain);
                else
                  hipLaunchKernelGGL((ln_bwd_kernel<1, DYT, XT>), dim3(grid), dim3(256), smem, st, (const DYT*)dy,
                                     (const XT*)x, gamma, mean, rstd, (const XT*)dres, (XT*)dx, part_g, part_b, rows,
                                     dim, rpb, acc_mode, (DYT*)dx_lo, pstride, plain);
              }));
  FFVC_LAUNCH_CHECK();
  return 0;
}

static int gn_chunks(int B, int HW, int* rows_per_chunk) {
  int nch = 2048 / (B < 1 ? 1 : B);
  if (nch > 64) nch = 64;
  if (nch < 1) nch = 1;
  int maxc = (HW + 63) / 64;
  if (nch > maxc) nch = maxc;
  if (nch < 1) nch = 1;
  *rows_per_chunk = (HW + nch - 1) / nch;
  nch = (HW + *rows_per_chunk - 1) / *rows_per_chunk;
  return nch;
}

extern "C" int64_t ffvc_groupnorm_ws_bytes(int B, int HW, int G) {
  int rpc;
  const int nch = gn_chunks(B, HW, &rpc);
  // + the arrival counters of the one-pass backward ([B][G][2] sums, then B counters)
  const int64_t two_pass = (int64_t)B * nch * G * 2 * (int64_t)sizeof(double);
  const int64_t one_pass = (int64_t)B * GNF_REP * G * 2 * (int64_t)sizeof(double) + (int64_t)B * GNF_CNT * (int64_t)sizeof(unsigned);
  return (two_pass > one_pass ? two_pass : one_pass) + 8;
}

static int gn_check(int B, int HW, int C, int G, int dtype, const char* who) {
  const int epc = dtype == FFVC_F32 ? 4 : 8;
  FFVC_CHECK_ARG(B > 0 && B <= 65535 && HW > 0 && C > 0 && G > 0 && G <= 64, "%s: bad dims B=%d HW=%d C=%d G=%d", who,
                 B, HW, C, G);
  FFVC_CHECK_ARG(C % G == 0 && C % epc == 0 && C <= 1024 && (C / epc) <= 256, "%s: C=%d unsupported (G=%d)", who, C, G);
  return 0;
}

extern "C" int ffvc_groupnorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean,
                                  float* rstd, void* ws, int B, int HW, int C, int G, float eps, int swish,
                                  int dtype, void* stream) {
  FFVC_CHECK_ARG(x && y && gamma && beta && mean && rstd && ws, "ffvc_groupnorm_fwd: null pointer");
  if (int e = gn_check(B, HW, C, G, dtype, "ffvc_groupnorm_fwd")) return e;
  hipStream_t st = (hipStream_t)stream;
  int rpc;
  const int nch = gn_chunks(B, HW, &rpc);
  // ~4096 workgroups in total (each one re-derives the group statistics from the chunk partials, so tiny
  // workgroups waste time), at least 64 pixels per workgroup
  int bpi = 4096 / (B < 1 ? 1 : B);
  if (bpi > HW / 64) bpi = HW / 64;
  if (bpi < 1) bpi = 1;
  const int rpb = (HW + bpi - 1) / bpi;
  const int nblk = (HW + rpb - 1) / rpb;
  DISPATCH_DT(dtype, T, {
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)x, (double*)ws, HW, C, G, rpc);
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, (T*)y, gamma, beta,
                       (const double*)ws, mean, rstd, HW, C, G, nch, eps, swish, rpb);
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_groupnorm_fwd_sums(const void* x, void* y, const float* gamma, const float* beta, float* mean,
                                       float* rstd, const double* sums, int B, int HW, int C, int G, float eps, int swish,
                                       int dtype, void* stream) {
  FFVC_CHECK_ARG(x && y && gamma && beta && mean && rstd && sums, "ffvc_groupnorm_fwd_sums: null pointer");
  if (int e = gn_check(B, HW, C, G, dtype, "ffvc_groupnorm_fwd_sums")) return e;
  hipStream_t st = (hipStream_t)stream;
  int bpi = 4096 / (B < 1 ? 1 : B);
  if (bpi > HW / 64) bpi = HW / 64;
  if (bpi < 1) bpi = 1;
  const int rpb = (HW + bpi - 1) / bpi;
  const int nblk = (HW + rpb - 1) / rpb;
  // the producer's sums have the layout of ONE chunk of partial moments: ws[B][1][G][2]
  DISPATCH_DT(dtype, T, {
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, (T*)y, gamma, beta, sums, mean,
                       rstd, HW, C, G, 1, eps, swish, rpb);
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                                  const float* mean, const float* rstd, const void* dres, void* dx, void* ws, int B,
                                  int HW, int C, int G, int swish, int dtype, void* stream) {
  FFVC_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && ws, "ffvc_groupnorm_bwd: null pointer");
  if (int e = gn_check(B, HW, C, G, dtype, "ffvc_groupnorm_bwd")) return e;
  hipStream_t st = (hipStream_t)stream;
  // one-pass form (gn_bwd_fused_kernel): 16-bit tensors whose image fits the registers of the co-resident workgroups
  {
    static int fused_opt = -1, cap_f16 = 0, cap_bf16 = 0;
    if (fused_opt < 0) {
      const char* e = getenv("FFVC_GN_BWD_FUSED");
      fused_opt = e ? atoi(e) : 0;   // measured (profiles/r04_gn_bwd_onepass.txt): slower than the two-pass form at every level -> opt-in
      int dev = 0, ncu = 0, occ = 0;
      (void)hipGetDevice(&dev);
      if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gn_bwd_fused_kernel<f16_t>, 256, 0) == hipSuccess && occ > 0)
        cap_f16 = ncu * (occ > 2 ? 2 : occ);
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gn_bwd_fused_kernel<uint16_t>, 256, 0) == hipSuccess && occ > 0)
        cap_bf16 = ncu * (occ > 2 ? 2 : occ);
    }
    const int cap = dtype == FFVC_F16 ? cap_f16 : (dtype == FFVC_BF16 ? cap_bf16 : 0);
    const int cpr = C / 8;
    if (fused_opt && cap > 0 && dtype != FFVC_F32 && C % 8 == 0 && cpr <= 256 && 256 % cpr == 0 && C <= 1024) {
      const int px_per_wg = GNF_NL * (256 / cpr);
      const int wg_per_img = (HW + px_per_wg - 1) / px_per_wg;
      if (wg_per_img <= cap) {
        int ngroups = cap / wg_per_img;
        if (ngroups > B) ngroups = B;
        double* sums = (double*)ws;
        unsigned* cnt = (unsigned*)(sums + (size_t)B * GNF_REP * G * 2);
        hipError_t me = hipMemsetAsync(ws, 0, (size_t)B * GNF_REP * G * 2 * sizeof(double) + (size_t)B * GNF_CNT * sizeof(unsigned), st);
        if (me != hipSuccess) {
          ffvc_set_error("ffvc_groupnorm_bwd: memset failed: %s", hipGetErrorString(me));
          return (int)me;
        }
        if (dtype == FFVC_F16)
          hipLaunchKernelGGL((gn_bwd_fused_kernel<f16_t>), dim3(ngroups * wg_per_img), dim3(256), 0, st, (const f16_t*)dy, (const f16_t*)x,
                             gamma, beta, mean, rstd, (const f16_t*)dres, (f16_t*)dx, sums, cnt, B, HW, C, G, swish, wg_per_img, ngroups);
        else
          hipLaunchKernelGGL((gn_bwd_fused_kernel<uint16_t>), dim3(ngroups * wg_per_img), dim3(256), 0, st, (const uint16_t*)dy,
                             (const uint16_t*)x, gamma, beta, mean, rstd, (const uint16_t*)dres, (uint16_t*)dx, sums, cnt, B, HW, C, G,
                             swish, wg_per_img, ngroups);
        FFVC_LAUNCH_CHECK();
        return 0;
      }
    }
  }
  // Experiment (FFVC_GN_BWD_CHUNK_MB=n, default 0 = off): image groups whose dy + x are at most n MiB, statistics + apply per group, so
  // that the apply pass finds in the 256 MiB Infinity Cache what the statistics pass just streamed.
  static int chunk_mb = -1;
  if (chunk_mb < 0) {
    const char* e = getenv("FFVC_GN_BWD_CHUNK_MB");
    chunk_mb = e ? atoi(e) : 0;
  }
  const int64_t img = (int64_t)HW * C * (dtype == FFVC_F32 ? 4 : 2);
  int nb = B;
  if (chunk_mb > 0) {
    int64_t k = ((int64_t)chunk_mb << 20) / (2 * img);
    nb = (int)(k < 1 ? 1 : (k > B ? B : k));
    if ((int64_t)nb * HW < 16384) nb = B;
  }
  if (nb < B) {
    // few images per launch: 256-pixel workgroups adding into NS replicated slots per (image, group) of a zeroed workspace
    constexpr int NS = 8;
    hipError_t me = hipMemsetAsync(ws, 0, (size_t)B * NS * G * 2 * sizeof(double), st);
    if (me != hipSuccess) {
      ffvc_set_error("ffvc_groupnorm_bwd: memset failed: %s", hipGetErrorString(me));
      return (int)me;
    }
    const int rpc = 256, nblk_s = (HW + rpc - 1) / rpc;
    for (int i0 = 0; i0 < B; i0 += nb) {
      const int Bc = B - i0 < nb ? B - i0 : nb;
      const char* dyc = (const char*)dy + i0 * img;
      const char* xc = (const char*)x + i0 * img;
      const char* drc = dres ? (const char*)dres + i0 * img : nullptr;
      char* dxc = (char*)dx + i0 * img;
      const float* mc = mean + (int64_t)i0 * G;
      const float* rc = rstd + (int64_t)i0 * G;
      double* wsc = (double*)ws + (int64_t)i0 * NS * G * 2;
      int bpi = 4096 / Bc;
      if (bpi > HW / 64) bpi = HW / 64;
      if (bpi < 1) bpi = 1;
      const int rpb = (HW + bpi - 1) / bpi;
      const int nblk = (HW + rpb - 1) / rpb;
      DISPATCH_DT(dtype, T, {
        hipLaunchKernelGGL((gn_bwd_stats_kernel<T, true>), dim3(nblk_s, Bc), dim3(256), 0, st, (const T*)dyc, (const T*)xc, gamma, beta, mc,
                           rc, wsc, HW, C, G, swish, rpc, NS);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(nblk, Bc), dim3(256), 0, st, (const T*)dyc, (const T*)xc, gamma, beta, mc, rc,
                           (const double*)wsc, (const T*)drc, (T*)dxc, HW, C, G, NS, swish, rpb);
      });
    }
    FFVC_LAUNCH_CHECK();
    return 0;
  }
  {
    int rpc;
    const int nch = gn_chunks(B, HW, &rpc);
    // ~4096 workgroups in total (each one re-derives the group statistics from the chunk partials, so tiny
    // workgroups waste time), at least 64 pixels per workgroup
    int bpi = 4096 / (B < 1 ? 1 : B);
    if (bpi > HW / 64) bpi = HW / 64;
    if (bpi < 1) bpi = 1;
    const int rpb = (HW + bpi - 1) / bpi;
    const int nblk = (HW + rpb - 1) / rpb;
    DISPATCH_DT(dtype, T, {
      hipLaunchKernelGGL((gn_bwd_stats_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma,
                         beta, mean, rstd, (double*)ws, HW, C, G, swish, rpc);
      hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma,
                         beta, mean, rstd, (const double*)ws, (const T*)dres, (T*)dx, HW, C, G, nch, swish, rpb);
    });
  }
  FFVC_LAUNCH_CHECK();
  return 0;
}

// GroupNorm backward with the statistics already accumulated by the dgrad convolution that produced dy (FFVC_F_GNB_SUMS,
// gnb_sums[B][G][2] fp64): the apply pass only — 3 reads + 1 write instead of 5 + 1.
extern "C" int ffvc_groupnorm_bwd_sums(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                       const float* rstd, const void* dres, void* dx, const double* sums, int B, int HW, int C, int G,
                                       int swish, int dtype, void* stream) {
  FFVC_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && sums, "ffvc_groupnorm_bwd_sums: null pointer");
  if (int e = gn_check(B, HW, C, G, dtype, "ffvc_groupnorm_bwd_sums")) return e;
  hipStream_t st = (hipStream_t)stream;
  int bpi = 4096 / (B < 1 ? 1 : B);
  if (bpi > HW / 64) bpi = HW / 64;
  if (bpi < 1) bpi = 1;
  const int rpb = (HW + bpi - 1) / bpi;
  const int nblk = (HW + rpb - 1) / rpb;
  DISPATCH_DT(dtype, T, {
    hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, beta, mean, rstd, sums,
                       (const T*)dres, (T*)dx, HW, C, G, 1, swish, rpb);
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

// GroupNorm (+ swish) whose output (also) leaves as fp8 bytes — the operand of the fp8 3x3 convolution that follows (see
// ffvc_layernorm_fwd_f8 for the byte contract).  sums != NULL: the moments come from the producing GEMM (ffvc_groupnorm_fwd_sums), ws
// is not used.  y may be NULL.  16-bit tensors only.
extern "C" int ffvc_groupnorm_fwd_f8(const void* x, void* y, void* y8, float* f8_state, int f8_fmt, const float* gamma, const float* beta,
                                     float* mean, float* rstd, void* ws, const double* sums, int B, int HW, int C, int G, float eps,
                                     int swish, int dtype, void* stream) {
  FFVC_CHECK_ARG(x && y8 && f8_state && gamma && beta && mean && rstd && (ws || sums), "ffvc_groupnorm_fwd_f8: null pointer");
  FFVC_CHECK_ARG(dtype == FFVC_F16 || dtype == FFVC_BF16, "ffvc_groupnorm_fwd_f8: 16-bit tensors only");
  FFVC_CHECK_ARG(f8_fmt == 0 || f8_fmt == 1, "ffvc_groupnorm_fwd_f8: f8_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(((uintptr_t)y8 % 8) == 0, "ffvc_groupnorm_fwd_f8: misaligned y8");
  if (int e = gn_check(B, HW, C, G, dtype, "ffvc_groupnorm_fwd_f8")) return e;
  hipStream_t st = (hipStream_t)stream;
  int rpc = 0;
  const int nch = sums ? 1 : gn_chunks(B, HW, &rpc);
  int bpi = 4096 / (B < 1 ? 1 : B);
  if (bpi > HW / 64) bpi = HW / 64;
  if (bpi < 1) bpi = 1;
  const int rpb = (HW + bpi - 1) / bpi;
  const int nblk = (HW + rpb - 1) / rpb;
  DISPATCH_DT(dtype, T, {
    if constexpr (sizeof(T) == 2) {
      if (!sums) hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)x, (double*)ws, HW, C, G, rpc);
      hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)x, (T*)y, gamma, beta,
                         sums ? sums : (const double*)ws, mean, rstd, HW, C, G, nch, eps, swish, rpb, (uint8_t*)y8, f8_state, f8_fmt);
    }
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

// GroupNorm backward (two-pass form) whose dx (also) leaves as fp8 bytes (e5m2 on the path) for the dgrad of the convolution that
// produced the normalised tensor.  dx may be NULL.  16-bit tensors only.
extern "C" int ffvc_groupnorm_bwd_f8(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                     const float* rstd, const void* dres, void* dx, void* dx8, float* f8_state, int f8_fmt, void* ws,
                                     int B, int HW, int C, int G, int swish, int dtype, void* stream) {
  FFVC_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx8 && f8_state && ws, "ffvc_groupnorm_bwd_f8: null pointer");
  FFVC_CHECK_ARG(dtype == FFVC_F16 || dtype == FFVC_BF16, "ffvc_groupnorm_bwd_f8: 16-bit tensors only");
  FFVC_CHECK_ARG(f8_fmt == 0 || f8_fmt == 1, "ffvc_groupnorm_bwd_f8: f8_fmt must be 0 (e4m3) or 1 (e5m2)");
  FFVC_CHECK_ARG(((uintptr_t)dx8 % 8) == 0, "ffvc_groupnorm_bwd_f8: misaligned dx8");
  if (int e = gn_check(B, HW, C, G, dtype, "ffvc_groupnorm_bwd_f8")) return e;
  hipStream_t st = (hipStream_t)stream;
  int rpc;
  const int nch = gn_chunks(B, HW, &rpc);
  int bpi = 4096 / (B < 1 ? 1 : B);
  if (bpi > HW / 64) bpi = HW / 64;
  if (bpi < 1) bpi = 1;
  const int rpb = (HW + bpi - 1) / bpi;
  const int nblk = (HW + rpb - 1) / rpb;
  DISPATCH_DT(dtype, T, {
    if constexpr (sizeof(T) == 2) {
      hipLaunchKernelGGL((gn_bwd_stats_kernel<T>), dim3(nch, B), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, beta, mean, rstd,
                         (double*)ws, HW, C, G, swish, rpc);
      hipLaunchKernelGGL((gn_bwd_apply_kernel<T>), dim3(nblk, B), dim3(256), 0, st, (const T*)dy, (const T*)x, gamma, beta, mean, rstd,
                         (const double*)ws, (const T*)dres, (T*)dx, HW, C, G, nch, swish, rpb, (uint8_t*)dx8, f8_state, f8_fmt);
    }
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_softmax_fwd(const float* s, void* p, int p_dtype, int64_t rows, int cols, int lds, int ldp,
                                float scale, int causal, int q_len, void* stream) {
  FFVC_CHECK_ARG(s && p, "ffvc_softmax_fwd: null pointer");
  FFVC_CHECK_ARG(rows > 0 && cols > 0 && cols <= 64 * SM_MAXE && ldp >= cols && ldp <= 64 * SM_MAXE && lds >= cols,
                 "ffvc_softmax_fwd: cols=%d ldp=%d lds=%d unsupported (max %d)", cols, ldp, lds, 64 * SM_MAXE);
  FFVC_CHECK_ARG(!causal || q_len > 0, "ffvc_softmax_fwd: causal needs q_len");
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(p_dtype, PT,
              hipLaunchKernelGGL((softmax_fwd_kernel<PT>), dim3(grid_rows(rows)), dim3(256), 0, st, s, (PT*)p, rows,
                                 cols, lds, ldp, scale, causal, q_len > 0 ? q_len : 1));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_softmax_bwd(const void* p, const float* dp, void* ds, int p_dtype, int64_t rows, int cols, int ldp,
                                int lddp, float scale, void* stream) {
  FFVC_CHECK_ARG(p && dp && ds, "ffvc_softmax_bwd: null pointer");
  FFVC_CHECK_ARG(rows > 0 && cols > 0 && ldp >= cols && ldp <= 64 * SM_MAXE && lddp >= cols,
                 "ffvc_softmax_bwd: cols=%d ldp=%d unsupported", cols, ldp);
  hipStream_t st = (hipStream_t)stream;
  DISPATCH_DT(p_dtype, PT,
              hipLaunchKernelGGL((softmax_bwd_kernel<PT>), dim3(grid_rows(rows)), dim3(256), 0, st, (const PT*)p, dp,
                                 (PT*)ds, rows, cols, ldp, lddp, scale));
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_sln_fwd(const float* hl, const float* w, const float* gamma, const float* beta, const float* gamma_s,
                            const float* beta_s, void* y, int y_dtype, float* mean, float* rstd, int64_t rows, int dim,
                            float eps, void* stream) {
  FFVC_CHECK_ARG(hl && w && gamma && beta && gamma_s && beta_s && y && mean && rstd, "ffvc_sln_fwd: null pointer");
  FFVC_CHECK_ARG(rows > 0 && dim > 0 && dim <= 64 * LN_MAXE, "ffvc_sln_fwd: dim=%d unsupported", dim);
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_rows(rows);
  DISPATCH_DT(y_dtype, YT, {
    if (dim % 4 == 0)
      hipLaunchKernelGGL((sln_fwd_kernel<4, YT>), dim3(grid), dim3(256), 0, st, hl, w, gamma, beta, gamma_s, beta_s,
                         (YT*)y, mean, rstd, rows, dim, eps);
    else
      hipLaunchKernelGGL((sln_fwd_kernel<1, YT>), dim3(grid), dim3(256), 0, st, hl, w, gamma, beta, gamma_s, beta_s,
                         (YT*)y, mean, rstd, rows, dim, eps);
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

static int sln_bwd_launch(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma,
                          const float* beta, const float* gamma_s, const float* beta_s, const float* mean,
                          const float* rstd, const float* dres, float* dhl, float* dw, float* part_g, float* part_b,
                          float* part_s, int64_t rows, int dim, void* stream, int acc_mode, float* part_s1 = nullptr,
                          int dw_acc = 0) {
  if (!part_s1 && part_s) part_s1 = part_s + 1;
  FFVC_CHECK_ARG(dy && hl && w && gamma && beta && gamma_s && beta_s && mean && rstd && dhl && dw && part_g && part_b &&
                     part_s, "ffvc_sln_bwd: null pointer");
  FFVC_CHECK_ARG(rows > 0 && dim > 0 && dim <= 64 * LN_MAXE, "ffvc_sln_bwd: dim=%d unsupported", dim);
  hipStream_t st = (hipStream_t)stream;
  const int rpb = ln_rows_per_block(rows);
  const int grid = ffvc_layernorm_bwd_blocks(rows);
  const size_t smem = (2 * (size_t)dim + 2) * sizeof(float);
  const bool x2 = dw_acc != 0 || part_s1 != part_s + 1;
  DISPATCH_DT(dy_dtype, DYT, {
    if (x2) {
      if (dim % 4 == 0)
        hipLaunchKernelGGL((sln_bwd_kernel<4, DYT, true>), dim3(grid), dim3(256), smem, st, (const DYT*)dy, hl, w, gamma, beta,
                           gamma_s, beta_s, mean, rstd, dres, dhl, dw, part_g, part_b, part_s, rows, dim, rpb, acc_mode, part_s1, dw_acc);
      else
        hipLaunchKernelGGL((sln_bwd_kernel<1, DYT, true>), dim3(grid), dim3(256), smem, st, (const DYT*)dy, hl, w, gamma, beta,
                           gamma_s, beta_s, mean, rstd, dres, dhl, dw, part_g, part_b, part_s, rows, dim, rpb, acc_mode, part_s1, dw_acc);
    } else {
      if (dim % 4 == 0)
        hipLaunchKernelGGL((sln_bwd_kernel<4, DYT, false>), dim3(grid), dim3(256), smem, st, (const DYT*)dy, hl, w, gamma, beta,
                           gamma_s, beta_s, mean, rstd, dres, dhl, dw, part_g, part_b, part_s, rows, dim, rpb, acc_mode, part_s1, dw_acc);
      else
        hipLaunchKernelGGL((sln_bwd_kernel<1, DYT, false>), dim3(grid), dim3(256), smem, st, (const DYT*)dy, hl, w, gamma, beta,
                           gamma_s, beta_s, mean, rstd, dres, dhl, dw, part_g, part_b, part_s, rows, dim, rpb, acc_mode, part_s1, dw_acc);
    }
  });
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_sln_bwd(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma,
                            const float* beta, const float* gamma_s, const float* beta_s, const float* mean,
                            const float* rstd, const float* dres, float* dhl, float* dw, float* part_g, float* part_b,
                            float* part_s, int64_t rows, int dim, void* stream) {
  return sln_bwd_launch(dy, dy_dtype, hl, w, gamma, beta, gamma_s, beta_s, mean, rstd, dres, dhl, dw, part_g, part_b,
                        part_s, rows, dim, stream, 0);
}

extern "C" int ffvc_sln_bwd_acc(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma,
                                const float* beta, const float* gamma_s, const float* beta_s, const float* mean,
                                const float* rstd, const float* dres, float* dhl, float* dw, float* dgamma, float* dbeta,
                                float* dscalars, int64_t rows, int dim, void* stream) {
  return sln_bwd_launch(dy, dy_dtype, hl, w, gamma, beta, gamma_s, beta_s, mean, rstd, dres, dhl, dw, dgamma, dbeta,
                        dscalars, rows, dim, stream, 1);
}

extern "C" int ffvc_sln_bwd_acc2(const void* dy, int dy_dtype, const float* hl, const float* w, const float* gamma,
                                 const float* beta, const float* gamma_s, const float* beta_s, const float* mean,
                                 const float* rstd, const float* dres, float* dhl, float* dw, int dw_accumulate, float* dgamma,
                                 float* dbeta, float* dgamma_s, float* dbeta_s, int64_t rows, int dim, void* stream) {
  FFVC_CHECK_ARG(dgamma_s && dbeta_s, "ffvc_sln_bwd_acc2: null pointer");
  return sln_bwd_launch(dy, dy_dtype, hl, w, gamma, beta, gamma_s, beta_s, mean, rstd, dres, dhl, dw, dgamma, dbeta,
                        dgamma_s, rows, dim, stream, 1, dbeta_s, dw_accumulate ? 1 : 0);
}
