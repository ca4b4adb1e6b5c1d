// common.h — shared device/host helpers for libffvc_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ffvc.h"

#define FFVC_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

// ---- error plumbing -------------------------------------------------------
void ffvc_set_error(const char* fmt, ...);

#define FFVC_CHECK_ARG(cond, ...)      \
  do {                                 \
    if (!(cond)) {                     \
      ffvc_set_error(__VA_ARGS__);     \
      return FFVC_E_BADARG;            \
    }                                  \
  } while (0)

#define FFVC_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      ffvc_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                     hipGetErrorString(e__));                            \
      return (int)e__;                                                   \
    }                                                                    \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round-to-nearest-even, two values per instruction) — the software
// sequence costs ~10 VALU instructions per pair, which shows in every bf16 epilogue.
typedef __attribute__((ext_vector_type(2))) float f32pair_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16pair_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32pair_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16pair_t));
}
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) { return (uint16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

// ---- f16 storage (FFVC_F16): IEEE half, 11 significant bits (8x finer than bf16), same MFMA rate -------------------
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  const f32pair_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));   // v_cvt_pk_f16_f32 (round-to-nearest-even)
}
__device__ __forceinline__ f32pair_t unpack_f16x2(uint32_t u) {
  return __builtin_convertvector(__builtin_bit_cast(f16x2_t, u), f32pair_t);
}

// Storage-type traits: T = uint16_t (bf16 bits), f16_t (IEEE half) or float.
template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<f16_t> {
  static constexpr int kDtype = FFVC_F16;
  static constexpr int kPerChunk = 8;
  __device__ static __forceinline__ float load(const f16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void store(f16_t* p, float v) { *p = (f16_t)v; }
};
// 16-bit formats by tag, for kernels that move raw 16-bit words (GEMM fragments, attention panels):
//   lo_pack2<T>(a, b) -> two values rounded to T in one dword;  lo_round<T>(v) -> v rounded to T, as float
template <typename T>
__device__ __forceinline__ uint32_t lo_pack2(float a, float b);
template <typename T>
__device__ __forceinline__ float lo_unpack(uint16_t bits);
template <>
struct ElemTraits<uint16_t> {
  static constexpr int kDtype = FFVC_BF16;
  static constexpr int kPerChunk = 8;  // elements per 16-byte chunk
  __device__ static __forceinline__ float load(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  __device__ static __forceinline__ void store(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }
};
template <>
struct ElemTraits<float> {
  static constexpr int kDtype = FFVC_F32;
  static constexpr int kPerChunk = 4;
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};

template <>
__device__ __forceinline__ uint32_t lo_pack2<uint16_t>(float a, float b) { return pack_bf16x2(a, b); }
template <>
__device__ __forceinline__ uint32_t lo_pack2<f16_t>(float a, float b) { return pack_f16x2(a, b); }
template <>
__device__ __forceinline__ float lo_unpack<uint16_t>(uint16_t bits) { return bf16_bits_to_f32(bits); }
template <>
__device__ __forceinline__ float lo_unpack<f16_t>(uint16_t bits) { return (float)__builtin_bit_cast(f16_t, bits); }
template <typename T>
__device__ __forceinline__ float lo_round(float v) { return lo_unpack<T>((uint16_t)(lo_pack2<T>(v, 0.0f) & 0xffffu)); }

// Load / store 4 consecutive elements as floats (8 B for bf16 / f16, 16 B for f32).
__device__ __forceinline__ f32x4_t load4(const f16_t* p) {
  const u32x2_t v = *(const u32x2_t*)p;
  const f32pair_t a = unpack_f16x2(v[0]), b = unpack_f16x2(v[1]);
  f32x4_t r = {a[0], a[1], b[0], b[1]};
  return r;
}
__device__ __forceinline__ void store4(f16_t* p, f32x4_t v) {
  u32x2_t o;
  o[0] = pack_f16x2(v[0], v[1]);
  o[1] = pack_f16x2(v[2], v[3]);
  *(u32x2_t*)p = o;
}
__device__ __forceinline__ f32x4_t load4(const uint16_t* p) {
  u32x2_t v = *(const u32x2_t*)p;
  f32x4_t r;
  r[0] = __uint_as_float(v[0] << 16);
  r[1] = __uint_as_float(v[0] & 0xffff0000u);
  r[2] = __uint_as_float(v[1] << 16);
  r[3] = __uint_as_float(v[1] & 0xffff0000u);
  return r;
}
__device__ __forceinline__ f32x4_t load4(const float* p) { return *(const f32x4_t*)p; }
__device__ __forceinline__ void store4(uint16_t* p, f32x4_t v) {
  u32x2_t o;
  o[0] = pack_bf16x2(v[0], v[1]);
  o[1] = pack_bf16x2(v[2], v[3]);
  *(u32x2_t*)p = o;
}
__device__ __forceinline__ void store4(float* p, f32x4_t v) { *(f32x4_t*)p = v; }

// Load / store 8 consecutive elements as floats.
struct f32x8 {
  float v[8];
};
__device__ __forceinline__ f32x8 load8(const uint16_t* p) {
  u32x4_t u = *(const u32x4_t*)p;
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __uint_as_float(u[i] << 16);
    r.v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ f32x8 load8(const f16_t* p) {
  const u32x4_t u = *(const u32x4_t*)p;
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32pair_t a = unpack_f16x2(u[i]);
    r.v[2 * i] = a[0];
    r.v[2 * i + 1] = a[1];
  }
  return r;
}
// the same conversions on a 16-byte chunk that is already in registers
template <typename T>
__device__ __forceinline__ f32x8 unpack8(const u32x4_t& u);
template <>
__device__ __forceinline__ f32x8 unpack8<uint16_t>(const u32x4_t& u) {
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __uint_as_float(u[i] << 16);
    r.v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
  return r;
}
template <>
__device__ __forceinline__ f32x8 unpack8<f16_t>(const u32x4_t& u) {
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32pair_t a = unpack_f16x2(u[i]);
    r.v[2 * i] = a[0];
    r.v[2 * i + 1] = a[1];
  }
  return r;
}
__device__ __forceinline__ void store8(f16_t* p, const f32x8& r) {
  u32x4_t u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = pack_f16x2(r.v[2 * i], r.v[2 * i + 1]);
  *(u32x4_t*)p = u;
}
__device__ __forceinline__ f32x8 load8(const float* p) {
  f32x4_t a = *(const f32x4_t*)p, b = *(const f32x4_t*)(p + 4);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
__device__ __forceinline__ void store8(uint16_t* p, const f32x8& r) {
  u32x4_t u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = pack_bf16x2(r.v[2 * i], r.v[2 * i + 1]);
  *(u32x4_t*)p = u;
}
__device__ __forceinline__ void store8(float* p, const f32x8& r) {
  f32x4_t a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = r.v[i];
    b[i] = r.v[4 + i];
  }
  *(f32x4_t*)p = a;
  *(f32x4_t*)(p + 4) = b;
}

// Non-temporal (streaming) variants for one-pass HBM-bound kernels (`nt` cache policy): enabled per file with
// -DFFVC_STREAM_NT=1.
#if defined(FFVC_STREAM_NT) && FFVC_STREAM_NT
__device__ __forceinline__ f32x4_t load4s(const uint16_t* p) {
  u32x2_t v = __builtin_nontemporal_load((const u32x2_t*)p);
  f32x4_t r;
  r[0] = __uint_as_float(v[0] << 16);
  r[1] = __uint_as_float(v[0] & 0xffff0000u);
  r[2] = __uint_as_float(v[1] << 16);
  r[3] = __uint_as_float(v[1] & 0xffff0000u);
  return r;
}
__device__ __forceinline__ f32x4_t load4s(const f16_t* p) {
  const u32x2_t v = __builtin_nontemporal_load((const u32x2_t*)p);
  const f32pair_t a = unpack_f16x2(v[0]), b = unpack_f16x2(v[1]);
  f32x4_t r = {a[0], a[1], b[0], b[1]};
  return r;
}
__device__ __forceinline__ void store4s(f16_t* p, f32x4_t v) {
  u32x2_t o;
  o[0] = pack_f16x2(v[0], v[1]);
  o[1] = pack_f16x2(v[2], v[3]);
  __builtin_nontemporal_store(o, (u32x2_t*)p);
}
__device__ __forceinline__ f32x8 load8s(const f16_t* p) {
  const u32x4_t u = __builtin_nontemporal_load((const u32x4_t*)p);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32pair_t a = unpack_f16x2(u[i]);
    r.v[2 * i] = a[0];
    r.v[2 * i + 1] = a[1];
  }
  return r;
}
__device__ __forceinline__ void store8s(f16_t* p, const f32x8& r) {
  u32x4_t u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = pack_f16x2(r.v[2 * i], r.v[2 * i + 1]);
  __builtin_nontemporal_store(u, (u32x4_t*)p);
}
__device__ __forceinline__ f32x4_t load4s(const float* p) { return __builtin_nontemporal_load((const f32x4_t*)p); }
__device__ __forceinline__ void store4s(uint16_t* p, f32x4_t v) {
  u32x2_t o;
  o[0] = pack_bf16x2(v[0], v[1]);
  o[1] = pack_bf16x2(v[2], v[3]);
  __builtin_nontemporal_store(o, (u32x2_t*)p);
}
__device__ __forceinline__ void store4s(float* p, f32x4_t v) { __builtin_nontemporal_store(v, (f32x4_t*)p); }
__device__ __forceinline__ f32x8 load8s(const uint16_t* p) {
  u32x4_t u = __builtin_nontemporal_load((const u32x4_t*)p);
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[2 * i] = __uint_as_float(u[i] << 16);
    r.v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
  return r;
}
__device__ __forceinline__ f32x8 load8s(const float* p) {
  f32x4_t a = __builtin_nontemporal_load((const f32x4_t*)p), b = __builtin_nontemporal_load((const f32x4_t*)(p + 4));
  f32x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r.v[i] = a[i];
    r.v[4 + i] = b[i];
  }
  return r;
}
__device__ __forceinline__ void store8s(uint16_t* p, const f32x8& r) {
  u32x4_t u;
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = pack_bf16x2(r.v[2 * i], r.v[2 * i + 1]);
  __builtin_nontemporal_store(u, (u32x4_t*)p);
}
__device__ __forceinline__ void store8s(float* p, const f32x8& r) {
  f32x4_t a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = r.v[i];
    b[i] = r.v[4 + i];
  }
  __builtin_nontemporal_store(a, (f32x4_t*)p);
  __builtin_nontemporal_store(b, (f32x4_t*)(p + 4));
}
#else
#define load4s load4
#define store4s store4
#define load8s load8
#define store8s store8
#endif

// ---- wave / block reductions ---------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- fp8 bytes straight from a producing kernel (delayed per-tensor scaling, the convention of ffvc_fp8_quant: state[0] = scale applied
// before the conversion, state[1] = running max |value| for the next update).  fmt 0 = OCP e4m3 (saturating at 448), 1 = e5m2 (57344);
// a NaN stays a NaN.
__device__ __forceinline__ float f8_sat_(float a, float lim) { return a != a ? a : fminf(fmaxf(a, -lim), lim); }
__device__ __forceinline__ uint32_t f8_pack4(int fmt, float a, float b, float c, float d) {
  int r = 0;
  if (fmt == 0) {
    r = __builtin_amdgcn_cvt_pk_fp8_f32(f8_sat_(a, 448.0f), f8_sat_(b, 448.0f), r, false);
    r = __builtin_amdgcn_cvt_pk_fp8_f32(f8_sat_(c, 448.0f), f8_sat_(d, 448.0f), r, true);
  } else {
    r = __builtin_amdgcn_cvt_pk_bf8_f32(f8_sat_(a, 57344.0f), f8_sat_(b, 57344.0f), r, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(f8_sat_(c, 57344.0f), f8_sat_(d, 57344.0f), r, true);
  }
  return (uint32_t)r;
}
// fold a wave's max |value| into state[1]: same-address atomics serialise in L2, so a wave whose maximum is not above what is already
// there (a plain, possibly stale read: the value only grows) skips it — after the first few waves almost all do
__device__ __forceinline__ void f8_amax_wave(float m, float* __restrict__ state) {
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.0f && !(m <= *(volatile float*)(state + 1)))
    atomicMax((unsigned int*)(state + 1), __float_as_uint(m));      // m >= 0: uint order == float order (a NaN maximum goes through)
}
// The same for a whole workgroup (every thread of the workgroup must call it): waves -> one LDS slot each -> ONE global atomic, skipped when
// the workgroup's maximum is not above what is already there.  (One atomic per wave — 16k on the same address within a few microseconds at
// 16448 LayerNorm rows — cost 18 us per launch in the step: 28.7 -> 46.9 us, rocprofv3 r4.)
__device__ __forceinline__ void f8_amax_block(float m, float* __restrict__ state) {
  __shared__ float f8_part[16];
  m = wave_max(m);
  const int nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) f8_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < nw; ++i) m = fmaxf(m, f8_part[i]);
    if (m > 0.0f && !(m <= *(volatile float*)(state + 1))) atomicMax((unsigned int*)(state + 1), __float_as_uint(m));
  }
}

// ---- activations ----------------------------------------------------------
__device__ __forceinline__ float act_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float act_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7), sharing exp(-x^2/2) between the cdf and the pdf.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __frcp_rn(1.0f + 0.3275911f * z);
  e = __expf(-z * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float act_gelu_fast(float x) {
  float cdf, e;
  gelu_parts_fast(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float act_gelu_grad_fast(float x) {
  float cdf, e;
  gelu_parts_fast(x, cdf, e);
  return cdf + x * 0.39894228040143267794f * e;
}
// Two-at-a-time variants for bf16-storage epilogues: the polynomial parts compile to v_pk_*_f32 (two lanes of fp32
// per issue slot) and the reciprocal is the 1-ulp v_rcp_f32 instead of the ~11-instruction IEEE division.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t rcp2_(f32x2_t v) {
  f32x2_t r;
  r[0] = __builtin_amdgcn_rcpf(v[0]);
  r[1] = __builtin_amdgcn_rcpf(v[1]);
  return r;
}
__device__ __forceinline__ f32x2_t exp2_2_(f32x2_t v) {
  f32x2_t r;
  r[0] = __builtin_amdgcn_exp2f(v[0]);
  r[1] = __builtin_amdgcn_exp2f(v[1]);
  return r;
}
__device__ __forceinline__ void gelu_parts_fast2(f32x2_t x, f32x2_t& cdf, f32x2_t& e) {
  f32x2_t ax;
  ax[0] = __builtin_fabsf(x[0]);
  ax[1] = __builtin_fabsf(x[1]);
  const f32x2_t z = ax * 0.70710678118654752440f;
  const f32x2_t t = rcp2_(z * 0.3275911f + 1.0f);
  e = exp2_2_(z * z * -1.44269504088896341f);
  const f32x2_t poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const f32x2_t erf_abs = 1.0f - poly * e;
  f32x2_t sg;
  sg[0] = __builtin_copysignf(erf_abs[0], x[0]);
  sg[1] = __builtin_copysignf(erf_abs[1], x[1]);
  cdf = sg * 0.5f + 0.5f;
}
__device__ __forceinline__ f32x2_t act_gelu_fast2(f32x2_t x) {
  f32x2_t cdf, e;
  gelu_parts_fast2(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ f32x2_t act_gelu_grad_fast2(f32x2_t x) {
  f32x2_t cdf, e;
  gelu_parts_fast2(x, cdf, e);
  return cdf + x * 0.39894228040143267794f * e;
}
__device__ __forceinline__ f32x2_t sigmoid_fast2(f32x2_t x) { return rcp2_(exp2_2_(x * -1.44269504088896341f) + 1.0f); }
__device__ __forceinline__ f32x2_t act_quickgelu_fast2(f32x2_t x) { return x * sigmoid_fast2(x * 1.702f); }
__device__ __forceinline__ f32x2_t act_quickgelu_grad_fast2(f32x2_t x) {
  const f32x2_t s = sigmoid_fast2(x * 1.702f);
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_quickgelu(float x) { return x * sigmoidf_(1.702f * x); }
__device__ __forceinline__ float act_quickgelu_grad(float x) {
  const float s = sigmoidf_(1.702f * x);
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
// bf16-storage variants: v_exp_f32 + v_rcp_f32 (1 ulp) instead of the IEEE division sequence
__device__ __forceinline__ float sigmoid_fast_(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
template <typename T>
__device__ __forceinline__ float act_swish_t(float x) { return x * (sizeof(T) == 2 ? sigmoid_fast_(x) : sigmoidf_(x)); }
template <typename T>
__device__ __forceinline__ float act_swish_grad_t(float x) {
  const float s = sizeof(T) == 2 ? sigmoid_fast_(x) : sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float act_swish(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float act_swish_grad(float x) {
  const float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

// One 32x32x16 MFMA on two 16-byte fragments of 16-bit operands (bf16 or f16: same rate, fp32 accumulate).
template <typename T>
__device__ __forceinline__ void mma_lo(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b);
template <>
__device__ __forceinline__ void mma_lo<uint16_t>(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_lo<f16_t>(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// One 16x16x32 MFMA (same FLOP rate; measured ~8 % less energy per FLOP than 32x32x16 under the power cap: the chip holds
// ~1.9 GHz instead of ~1.7 GHz, profiles/r03_power_ceiling.txt).  A / B fragment: lane (r = lane & 15, k = 8 * (lane >> 4) .. + 7);
// C: lane owns column (lane & 15) of the B side and rows 4 * (lane >> 4) .. + 3 of the A side.
template <typename T>
__device__ __forceinline__ void mma16_lo(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b);
template <>
__device__ __forceinline__ void mma16_lo<uint16_t>(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16_lo<f16_t>(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}

// Run a statement with `T` bound to the storage type of a dtype code.
#define DISPATCH_DT(code, T, ...)          \
  do {                                     \
    if ((code) == FFVC_BF16) {             \
      using T = uint16_t;                  \
      __VA_ARGS__;                         \
    } else if ((code) == FFVC_F16) {       \
      using T = f16_t;                     \
      __VA_ARGS__;                         \
    } else {                               \
      using T = float;                     \
      __VA_ARGS__;                         \
    }                                      \
  } while (0)

// dtype code -> element size
static inline int ffvc_dtype_size(int code) { return code == FFVC_F32 ? 4 : 2; }
static inline bool ffvc_dtype_ok(int code) { return code == FFVC_BF16 || code == FFVC_F32 || code == FFVC_F16; }

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
