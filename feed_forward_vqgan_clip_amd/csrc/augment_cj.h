// augment_cj.h — kornia 0.5.10 ColorJitter on ONE pixel, with forward-mode derivatives (reference main.py:166-168:
// K.ColorJitter(hue=0.1, saturation=0.1, p=0.7) / 'Ji2').  kornia/augmentation/augmentation.py::ColorJitter.apply_transform runs
//   0 adjust_brightness(x, b - 1)   x + (b - 1), clamp [0,1]
//   1 adjust_contrast(x, c)         x * c,       clamp [0,1]
//   2 adjust_saturation(x, s)       rgb -> hsv, s' = clamp(s * f, 0, 1), hsv -> rgb
//   3 adjust_hue(x, h * 2 pi)       rgb -> hsv, h' = fmod(h + 2 pi * f, 2 pi), hsv -> rgb
// in a batch-wide random order (torch.randperm(4)); hsv as in kornia/color/hsv.py (first maximal channel, eps 1e-6 in s).
// The map is piecewise smooth in RGB; the backward pass of the fused augmentation kernel needs J^T g, so every value carries
// its three partial derivatives with respect to the input (r, g, b) (a dual number with three tangents).
#pragma once
#include "common.h"

namespace ffvc_cj {

struct D3 {
  float v, d[3];
};
__device__ __forceinline__ D3 mk(float v, float a = 0.f, float b = 0.f, float c = 0.f) { return D3{v, {a, b, c}}; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return D3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return D3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ D3 operator*(D3 a, D3 b) {
  return D3{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ D3 operator*(D3 a, float s) { return D3{a.v * s, {a.d[0] * s, a.d[1] * s, a.d[2] * s}}; }
__device__ __forceinline__ D3 operator+(D3 a, float s) { return D3{a.v + s, {a.d[0], a.d[1], a.d[2]}}; }
__device__ __forceinline__ D3 operator/(D3 a, D3 b) {
  const float ib = 1.0f / b.v, q = a.v * ib;
  return D3{q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
__device__ __forceinline__ D3 clamp01(D3 a) {
  if (a.v < 0.f) return mk(0.f);
  if (a.v > 1.f) return mk(1.f);
  return a;
}
__device__ __forceinline__ D3 one_minus(D3 a) { return D3{1.0f - a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }

// kornia rgb_to_hsv; the hue is returned in sextant units h6 in [0, 6) (kornia: 2 pi * ((h6 / 6) mod 1))
__device__ __forceinline__ void rgb_to_hsv(const D3 (&c)[3], D3& h6, D3& s, D3& v) {
  const int mi = (c[0].v >= c[1].v && c[0].v >= c[2].v) ? 0 : (c[1].v >= c[2].v ? 1 : 2);      // FIRST maximal channel
  const D3 maxc = c[mi];
  D3 minc = c[0];
  if (c[1].v < minc.v) minc = c[1];
  if (c[2].v < minc.v) minc = c[2];
  v = maxc;
  D3 deltac = maxc - minc;
  s = deltac / (v + 1e-6f);
  if (deltac.v == 0.f) deltac = mk(1.0f);
  const D3 rc = maxc - c[0], gc = maxc - c[1], bc = maxc - c[2];
  D3 h = mi == 0 ? (bc - gc) : (mi == 1 ? (deltac * 2.0f + rc - bc) : (deltac * 4.0f + gc - rc));
  h = h / deltac;
  h.v -= 6.0f * floorf(h.v * (1.0f / 6.0f));                                                    // (h / 6) % 1, python sign rule
  if (h.v >= 6.0f) h.v = 0.f;
  h6 = h;
}

__device__ __forceinline__ void hsv_to_rgb(D3 h6, D3 s, D3 v, D3 (&o)[3]) {
  h6.v -= 6.0f * floorf(h6.v * (1.0f / 6.0f));                                                  // (h * 6) % 6
  int hi = (int)floorf(h6.v);
  if (hi > 5) hi = 5;
  const D3 f = h6 + (-(float)hi);
  const D3 p = v * one_minus(s), q = v * one_minus(f * s), t = v * one_minus(one_minus(f) * s);
  switch (hi) {
    case 0: o[0] = v; o[1] = t; o[2] = p; break;
    case 1: o[0] = q; o[1] = v; o[2] = p; break;
    case 2: o[0] = p; o[1] = v; o[2] = t; break;
    case 3: o[0] = p; o[1] = q; o[2] = v; break;
    case 4: o[0] = t; o[1] = p; o[2] = v; break;
    default: o[0] = v; o[1] = p; o[2] = q; break;
  }
}

// cj: [on, brightness, contrast, saturation, hue (turns), order code o0 + 4 o1 + 16 o2 + 64 o3, -, -]
// in: rgb values; out: jittered values and J[i][j] = d out_i / d in_j
__device__ __forceinline__ void color_jitter(const float* __restrict__ cj, const float (&rgb)[3], float (&out)[3], float (&J)[3][3]) {
  D3 c[3] = {mk(rgb[0], 1.f, 0.f, 0.f), mk(rgb[1], 0.f, 1.f, 0.f), mk(rgb[2], 0.f, 0.f, 1.f)};
  const int code = (int)cj[5];
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int op = (code >> (2 * k)) & 3;
    if (op == 0) {
      const float b = cj[1] - 1.0f;
      for (int i = 0; i < 3; ++i) c[i] = clamp01(c[i] + b);
    } else if (op == 1) {
      for (int i = 0; i < 3; ++i) c[i] = clamp01(c[i] * cj[2]);
    } else {
      D3 h6, s, v;
      rgb_to_hsv(c, h6, s, v);
      if (op == 2) {
        s = clamp01(s * cj[3]);
      } else {
        // fmod(h + 2 pi f, 2 pi) keeps the dividend's sign; hsv_to_rgb's python-style (h * 6) % 6 then folds into [0, 6)
        h6.v = fmodf(h6.v + 6.0f * cj[4], 6.0f);
      }
      hsv_to_rgb(h6, s, v, c);
    }
  }
  for (int i = 0; i < 3; ++i) {
    out[i] = c[i].v;
    for (int j = 0; j < 3; ++j) J[i][j] = c[i].d[j];
  }
}

}  // namespace ffvc_cj
