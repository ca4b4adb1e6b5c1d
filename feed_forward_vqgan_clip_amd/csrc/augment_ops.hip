// augment_ops.hip — the MakeCutouts augmentations that do not compose into one homography (reference main.py:169,179,181:
// K.RandomSharpness 'Sh', K.RandomElasticTransform 'Et', K.RandomThinPlateSpline 'Ts'), as image -> image kernels on the
// cutout batch x [N,3,S,S] fp32 (HBM-bound: one read + one write per pass).  kornia 0.5.10 semantics as restated in
// oracle/kornia_aug.py:
//   sharpness        out = blur + (x - blur) * f, blur = clamp01(3x3 conv [[1,1,1],[1,5,1],[1,1,1]]/13) inside, x on the border;
//                    clamped to [0,1] unless 0 <= f <= 1                                   (kornia/enhance/adjust.py)
//   warp by a grid   bilinear grid_sample(align_corners=False, padding_mode='zeros') at NORMALISED coordinates
//   thin-plate grid  warped(p) = sum_k w_k U(|p - c_k|^2) + a_0 + A p, U(d2) = 0.5 d2 log(d2 + 1e-6), p on linspace(-1,1)^2
//   elastic grid     identity + alpha * Gaussian(63, sigma 32, reflect) * noise, clamped to [-1,1]
// `on[n] == 0` passes sample n through untouched (kornia applies an operator to the samples its Bernoulli draw selects).
#include "common.h"

namespace {

inline int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8 * 256 ? 8 * 256 : g));
}

__device__ __forceinline__ float blur3(const float* __restrict__ p, int S) {
  return (p[-S - 1] + p[-S] + p[-S + 1] + p[-1] + 5.0f * p[0] + p[1] + p[S - 1] + p[S] + p[S + 1]) * (1.0f / 13.0f);
}

__global__ __launch_bounds__(256) void sharpness_fwd_kernel(const float* __restrict__ x, const float* __restrict__ factor,
                                                            const float* __restrict__ on, float* __restrict__ y, int64_t n,
                                                            int S) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S);
    const int img = (int)(i / ((int64_t)3 * S * S));
    const float v = x[i];
    if (on[img] == 0.0f) {
      y[i] = v;
      continue;
    }
    const bool inner = ox > 0 && oy > 0 && ox < S - 1 && oy < S - 1;
    const float res = inner ? fminf(fmaxf(blur3(x + i, S), 0.0f), 1.0f) : v;
    const float f = factor[img];
    float o = res + (v - res) * f;
    if (!(f >= 0.0f && f <= 1.0f)) o = fminf(fmaxf(o, 0.0f), 1.0f);
    y[i] = o;
  }
}

// dx[p] = f * gate(p) * g[p] + sum over inner pixels q in the 3x3 neighbourhood of p:  K[q - p] * bgate(q) * (1 - f) * gate(q) * g[q]
// gate(q) = output clamp not active at q, bgate(q) = blur clamp not active at q (both recomputed from x)
__global__ __launch_bounds__(256) void sharpness_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ factor, const float* __restrict__ on,
                                                            float* __restrict__ dx, int64_t n, int S) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S);
    const int img = (int)(i / ((int64_t)3 * S * S));
    if (on[img] == 0.0f) {
      dx[i] = g[i];
      continue;
    }
    const float f = factor[img];
    const bool clampy = !(f >= 0.0f && f <= 1.0f);
    float acc = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dxx = -1; dxx <= 1; ++dxx) {
        const int qx = ox + dxx, qy = oy + dy;
        if (qx < 0 || qy < 0 || qx >= S || qy >= S) continue;
        const int64_t q = i + (int64_t)dy * S + dxx;
        const bool inner = qx > 0 && qy > 0 && qx < S - 1 && qy < S - 1;
        const float xv = x[q];
        float res = xv;
        bool bgate = false;
        if (inner) {
          const float bl = blur3(x + q, S);
          bgate = bl >= 0.0f && bl <= 1.0f;
          res = fminf(fmaxf(bl, 0.0f), 1.0f);
        }
        const float o = res + (xv - res) * f;
        const bool gate = !clampy || (o >= 0.0f && o <= 1.0f);
        if (!gate) continue;
        const float gq = g[q];
        if (dy == 0 && dxx == 0) acc += f * gq + ((inner && bgate) ? (1.0f - f) * gq * (5.0f / 13.0f) : 0.0f) + (inner ? 0.0f : (1.0f - f) * gq);
        else if (inner && bgate) acc += (1.0f - f) * gq * (1.0f / 13.0f);
      }
    dx[i] = acc;
  }
}

// normalised coordinate -> pixel coordinate of grid_sample(align_corners=False)
__device__ __forceinline__ float unnorm(float c, int S) { return ((c + 1.0f) * (float)S - 1.0f) * 0.5f; }

__global__ __launch_bounds__(256) void warp_grid_fwd_kernel(const float* __restrict__ x, const float* __restrict__ grid,
                                                            const float* __restrict__ on, float* __restrict__ y, int N, int S) {
  const int64_t n = (int64_t)N * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int img = (int)(i / ((int64_t)S * S));
    const int64_t pix = i - (int64_t)img * S * S;
    const float* src = x + (int64_t)img * 3 * S * S;
    float* dst = y + (int64_t)img * 3 * S * S + pix;
    if (on[img] == 0.0f) {
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[(int64_t)c * S * S] = src[(int64_t)c * S * S + pix];
      continue;
    }
    const float fx = unnorm(grid[2 * i], S), fy = unnorm(grid[2 * i + 1], S);
    const float flx = floorf(fx), fly = floorf(fy);
    const int x0 = (int)flx, y0 = (int)fly;
    const float wx = fx - flx, wy = fy - fly;
    const bool vx0 = x0 >= 0 && x0 < S, vx1 = x0 + 1 >= 0 && x0 + 1 < S, vy0 = y0 >= 0 && y0 < S, vy1 = y0 + 1 >= 0 && y0 + 1 < S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* sc = src + (int64_t)c * S * S;
      const float v00 = (vx0 && vy0) ? sc[(int64_t)y0 * S + x0] : 0.0f, v01 = (vx1 && vy0) ? sc[(int64_t)y0 * S + x0 + 1] : 0.0f;
      const float v10 = (vx0 && vy1) ? sc[(int64_t)(y0 + 1) * S + x0] : 0.0f, v11 = (vx1 && vy1) ? sc[(int64_t)(y0 + 1) * S + x0 + 1] : 0.0f;
      dst[(int64_t)c * S * S] = (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
    }
  }
}

__global__ __launch_bounds__(256) void warp_grid_bwd_kernel(const float* __restrict__ g, const float* __restrict__ grid,
                                                            const float* __restrict__ on, float* __restrict__ dx, int N, int S) {
  const int64_t n = (int64_t)N * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int img = (int)(i / ((int64_t)S * S));
    const int64_t pix = i - (int64_t)img * S * S;
    const float* gs = g + (int64_t)img * 3 * S * S + pix;
    float* dst = dx + (int64_t)img * 3 * S * S;
    if (on[img] == 0.0f) {
#pragma unroll
      for (int c = 0; c < 3; ++c) atomicAdd(dst + (int64_t)c * S * S + pix, gs[(int64_t)c * S * S]);
      continue;
    }
    const float fx = unnorm(grid[2 * i], S), fy = unnorm(grid[2 * i + 1], S);
    const float flx = floorf(fx), fly = floorf(fy);
    const int x0 = (int)flx, y0 = (int)fly;
    const float wx = fx - flx, wy = fy - fly;
    const bool vx0 = x0 >= 0 && x0 < S, vx1 = x0 + 1 >= 0 && x0 + 1 < S, vy0 = y0 >= 0 && y0 < S, vy1 = y0 + 1 >= 0 && y0 + 1 < S;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float* dc = dst + (int64_t)c * S * S;
      const float gv = gs[(int64_t)c * S * S];
      if (vx0 && vy0) atomicAdd(dc + (int64_t)y0 * S + x0, gv * (1.f - wy) * (1.f - wx));
      if (vx1 && vy0) atomicAdd(dc + (int64_t)y0 * S + x0 + 1, gv * (1.f - wy) * wx);
      if (vx0 && vy1) atomicAdd(dc + (int64_t)(y0 + 1) * S + x0, gv * wy * (1.f - wx));
      if (vx1 && vy1) atomicAdd(dc + (int64_t)(y0 + 1) * S + x0 + 1, gv * wy * wx);
    }
  }
}

// tps[n]: 5 centres (x, y), 5 kernel weights (wx, wy), affine a0 (2), A rows for x (2) and y (2)  = 10 + 10 + 6 = 26 floats:
//   [0..9] centres, [10..19] kernel weights (k-major, (x,y) pairs), [20,21] a0, [22,23] coefficients of p.x, [24,25] of p.y
__global__ __launch_bounds__(256) void tps_grid_kernel(const float* __restrict__ tps, float* __restrict__ grid, int N, int S) {
  const int64_t n = (int64_t)N * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int img = (int)(i / ((int64_t)S * S));
    const int64_t pix = i - (int64_t)img * S * S;
    const int oy = (int)(pix / S), ox = (int)(pix - (int64_t)oy * S);
    const float px = S > 1 ? -1.0f + 2.0f * (float)ox / (float)(S - 1) : 0.0f;
    const float py = S > 1 ? -1.0f + 2.0f * (float)oy / (float)(S - 1) : 0.0f;
    const float* t = tps + (int64_t)img * 26;
    float gx = t[20] + px * t[22] + py * t[24], gy = t[21] + px * t[23] + py * t[25];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float ddx = px - t[2 * k], ddy = py - t[2 * k + 1];
      const float d2 = ddx * ddx + ddy * ddy;
      const float u = 0.5f * d2 * logf(d2 + 1e-6f);
      gx += u * t[10 + 2 * k];
      gy += u * t[11 + 2 * k];
    }
    grid[2 * i] = gx;
    grid[2 * i + 1] = gy;
  }
}

// one separable Gaussian pass (reflect border, as F.pad(mode='reflect')) over planes [P][S][S]: along x (axis 0) or y (axis 1)
__global__ __launch_bounds__(256) void gauss1d_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int S,
                                                      int ksize, float sigma, int axis) {
  extern __shared__ float wts[];
  float sum = 0.0f;
  for (int k = 0; k < ksize; ++k) {
    const float xx = (float)(k - ksize / 2) + ((ksize & 1) ? 0.0f : 0.5f);
    sum += expf(-xx * xx / (2.0f * sigma * sigma));
  }
  for (int k = threadIdx.x; k < ksize; k += 256) {
    const float xx = (float)(k - ksize / 2) + ((ksize & 1) ? 0.0f : 0.5f);
    wts[k] = expf(-xx * xx / (2.0f * sigma * sigma)) / sum;
  }
  __syncthreads();
  const int h = ksize / 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S);
    const int64_t plane = i - (int64_t)oy * S - ox;
    float acc = 0.0f;
    for (int k = 0; k < ksize; ++k) {
      int c = (axis == 0 ? ox : oy) + k - h;
      // reflect without repeating the border sample; a 63-tap kernel on a small image may bounce more than once
      while (c < 0 || c >= S) c = c < 0 ? -c : 2 * (S - 1) - c;
      acc += wts[k] * (axis == 0 ? src[plane + (int64_t)oy * S + c] : src[plane + (int64_t)c * S + ox]);
    }
    dst[i] = acc;
  }
}

// disp [N,2,S,S] (blurred noise) -> grid [N,S,S,2] = clamp(identity + alpha * disp, -1, 1)
__global__ __launch_bounds__(256) void elastic_grid_kernel(const float* __restrict__ disp, float* __restrict__ grid, int N, int S,
                                                           float ax, float ay) {
  const int64_t n = (int64_t)N * S * S;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int img = (int)(i / ((int64_t)S * S));
    const int64_t pix = i - (int64_t)img * S * S;
    const int oy = (int)(pix / S), ox = (int)(pix - (int64_t)oy * S);
    const float px = S > 1 ? -1.0f + 2.0f * (float)ox / (float)(S - 1) : 0.0f;
    const float py = S > 1 ? -1.0f + 2.0f * (float)oy / (float)(S - 1) : 0.0f;
    const float* d = disp + (int64_t)img * 2 * S * S + pix;
    grid[2 * i] = fminf(fmaxf(px + ax * d[0], -1.0f), 1.0f);
    grid[2 * i + 1] = fminf(fmaxf(py + ay * d[(int64_t)S * S], -1.0f), 1.0f);
  }
}

}  // namespace

extern "C" int ffvc_sharpness_fwd(const float* x, const float* factor, const float* on, float* y, int N, int S, void* stream) {
  FFVC_CHECK_ARG(x && factor && on && y && N > 0 && S > 2, "ffvc_sharpness_fwd: bad args");
  const int64_t n = (int64_t)N * 3 * S * S;
  hipLaunchKernelGGL(sharpness_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, factor, on, y, n, S);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_sharpness_bwd(const float* g, const float* x, const float* factor, const float* on, float* dx, int N, int S,
                                  void* stream) {
  FFVC_CHECK_ARG(g && x && factor && on && dx && N > 0 && S > 2, "ffvc_sharpness_bwd: bad args");
  const int64_t n = (int64_t)N * 3 * S * S;
  hipLaunchKernelGGL(sharpness_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, x, factor, on, dx, n, S);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_warp_grid_fwd(const float* x, const float* grid, const float* on, float* y, int N, int S, void* stream) {
  FFVC_CHECK_ARG(x && grid && on && y && N > 0 && S > 1, "ffvc_warp_grid_fwd: bad args");
  hipLaunchKernelGGL(warp_grid_fwd_kernel, dim3(grid_for((int64_t)N * S * S)), dim3(256), 0, (hipStream_t)stream, x, grid, on, y, N, S);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_warp_grid_bwd(const float* g, const float* grid, const float* on, float* dx, int N, int S, void* stream) {
  FFVC_CHECK_ARG(g && grid && on && dx && N > 0 && S > 1, "ffvc_warp_grid_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dx, 0, (size_t)N * 3 * S * S * sizeof(float), st);
  if (e != hipSuccess) {
    ffvc_set_error("ffvc_warp_grid_bwd: memset failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  hipLaunchKernelGGL(warp_grid_bwd_kernel, dim3(grid_for((int64_t)N * S * S)), dim3(256), 0, st, g, grid, on, dx, N, S);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_tps_grid(const float* tps, float* grid, int N, int S, void* stream) {
  FFVC_CHECK_ARG(tps && grid && N > 0 && S > 1, "ffvc_tps_grid: bad args");
  hipLaunchKernelGGL(tps_grid_kernel, dim3(grid_for((int64_t)N * S * S)), dim3(256), 0, (hipStream_t)stream, tps, grid, N, S);
  FFVC_LAUNCH_CHECK();
  return 0;
}

extern "C" int ffvc_elastic_grid(const float* noise, float* tmp, float* disp, float* grid, int N, int S, int ksize, float sigma,
                                 float alpha_x, float alpha_y, void* stream) {
  FFVC_CHECK_ARG(noise && tmp && disp && grid && N > 0 && S > 1 && ksize > 0 && ksize <= 255 && sigma > 0.0f, "ffvc_elastic_grid: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)N * 2 * S * S;
  hipLaunchKernelGGL(gauss1d_kernel, dim3(grid_for(n)), dim3(256), ksize * sizeof(float), st, noise, tmp, n, S, ksize, sigma, 0);
  hipLaunchKernelGGL(gauss1d_kernel, dim3(grid_for(n)), dim3(256), ksize * sizeof(float), st, (const float*)tmp, disp, n, S, ksize, sigma, 1);
  hipLaunchKernelGGL(elastic_grid_kernel, dim3(grid_for((int64_t)N * S * S)), dim3(256), 0, st, (const float*)disp, grid, N, S, alpha_x, alpha_y);
  FFVC_LAUNCH_CHECK();
  return 0;
}
