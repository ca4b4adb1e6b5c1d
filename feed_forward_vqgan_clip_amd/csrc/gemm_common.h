// gemm_common.h — pieces shared by the GEMM kernel generations (gemm.hip: register-staged v1 for every mode /
// dtype; gemm2.hip: LDS-DMA staged bf16 fast path): activation helpers and the fused epilogue.
#pragma once
#include "common.h"

namespace ffvc_gemm_detail {

// bf16 storage rounds the result to 2^-8 relative: the 1.5e-7-accurate erf (common.h) is exact for that purpose
// and ~3x cheaper in the epilogue; fp32 ("parity") storage keeps libm's erff.
template <typename T>
__device__ __forceinline__ float apply_act(int act, float v) {
  if (act == FFVC_ACT_GELU) return sizeof(T) == 2 ? act_gelu_fast(v) : act_gelu(v);
  if (act == FFVC_ACT_QUICKGELU) return act_quickgelu(v);
  if (act == FFVC_ACT_LRELU) return v > 0.0f ? v : 0.01f * v;
  if (act == FFVC_ACT_TANH) return tanhf(v);
  return v;
}
template <typename T>
__device__ __forceinline__ float apply_act_grad(int act, float pre) {
  if (act == FFVC_ACT_GELU) return sizeof(T) == 2 ? act_gelu_grad_fast(pre) : act_gelu_grad(pre);
  if (act == FFVC_ACT_QUICKGELU) return act_quickgelu_grad(pre);
  if (act == FFVC_ACT_LRELU) return pre > 0.0f ? 1.0f : 0.01f;
  if (act == FFVC_ACT_TANH) {
    const float t = tanhf(pre);
    return 1.0f - t * t;
  }
  return 1.0f;
}


// Fused epilogue for a (64*MT)x128 tile owned by 4 waves (2(M) x 2(N), (32*MT)x64 each, 2 x MT MFMA 32x32 tiles):
// lane holds, per (nt, mt, q), 4 consecutive n for one m (MFMA roles swapped: A = W rows, B = X rows).
// VEC_ONLY drops the element-wise fallback (callers guarantee vec_ok and N % 4 == 0), which keeps the fully
// unrolled code small enough for the accumulators to stay in registers at MT = 4.
template <typename T>
__device__ __forceinline__ void epilogue_quad(const ffvc_gemm_desc& p, f32x4_t v, int n, int64_t yrow, int64_t rrow,
                                              int64_t arow, int flags, bool vec) {
  const bool out_f32 = flags & FFVC_F_OUT_F32;
  const bool res_f32 = flags & FFVC_F_RES_F32;
  if (vec) {
    if (p.bias && !(flags & FFVC_F_BIAS_ALONG_M)) v += *(const f32x4_t*)(p.bias + n);
    if (flags & FFVC_F_MUL_ACT_GRAD) {
      const f32x4_t pre = load4((const T*)p.aux + arow + n);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= (flags & FFVC_F_AUX_ACTGRAD) ? pre[j] : apply_act_grad<T>(p.act, pre[j]);
    } else if (p.act != FFVC_ACT_NONE) {
      if (flags & FFVC_F_WRITE_PREACT) store4((T*)p.aux + arow + n, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = apply_act<T>(p.act, v[j]);
    }
    if (p.residual)
      v += res_f32 ? load4((const float*)p.residual + rrow + n) : load4((const T*)p.residual + rrow + n);
    if (flags & FFVC_F_ATOMIC_OUT) {
      float* yp = (float*)p.y + yrow + n;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(yp + j, v[j]);
    } else if (flags & FFVC_F_ACCUM_OUT) {
      float* yp = (float*)p.y + yrow + n;
      store4(yp, load4(yp) + v);
    } else if (out_f32) {
      store4((float*)p.y + yrow + n, v);
    } else {
      store4((T*)p.y + yrow + n, v);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (n + j >= p.N) continue;
      float u = v[j];
      if (p.bias && !(flags & FFVC_F_BIAS_ALONG_M)) u += p.bias[n + j];
      if (flags & FFVC_F_MUL_ACT_GRAD) {
        const float av = ElemTraits<T>::load((const T*)p.aux + arow + n + j);
        u *= (flags & FFVC_F_AUX_ACTGRAD) ? av : apply_act_grad<T>(p.act, av);
      } else if (p.act != FFVC_ACT_NONE) {
        if (flags & FFVC_F_WRITE_PREACT) ElemTraits<T>::store((T*)p.aux + arow + n + j, u);
        u = apply_act<T>(p.act, u);
      }
      if (p.residual)
        u += res_f32 ? ((const float*)p.residual)[rrow + n + j] : ElemTraits<T>::load((const T*)p.residual + rrow + n + j);
      if (flags & FFVC_F_ATOMIC_OUT)
        atomicAdd((float*)p.y + yrow + n + j, u);
      else if (flags & FFVC_F_ACCUM_OUT)
        ((float*)p.y)[yrow + n + j] += u;
      else if (out_f32)
        ((float*)p.y)[yrow + n + j] = u;
      else
        ElemTraits<T>::store((T*)p.y + yrow + n + j, u);
    }
  }
}

// NSPLIT: the wave's two 32-column blocks sit 128 columns apart (n0 + nt*128 + wn*32: gemm8_kernel) instead of side by
// side (n0 + wn*64 + nt*32).
template <typename T, int MT = 2, bool VEC_ONLY = false, bool NSPLIT = false>
__device__ __forceinline__ void gemm_epilogue(const ffvc_gemm_desc& p, f32x16_t (&acc)[2][MT], int m0, int n0, int wm,
                                              int wn, int lane, int zo, int zi, int vec_ok, int zs = -1) {
  const int l31 = lane & 31, h = lane >> 5;
  const int flags = p.flags;
  if (zs < 0) zs = blockIdx.z;    // split-K slab index (explicit for persistent kernels)
  const int64_t ybz = zo * p.ybo + zi * p.ybi + (int64_t)zs * p.slab_stride;
  const int64_t rbz = zo * p.rbo + zi * p.rbi;
  const int64_t abz = zo * p.abo + zi * p.abi;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (32 * MT) + mt * 32 + l31;
    const bool mok = m < p.M;
    const int mm = mok ? m : 0;
    const int64_t yrow = ybz + (p.y_mi ? (int64_t)(mm / p.y_mi) * p.y_so + (int64_t)(mm % p.y_mi) * p.y_sm
                                       : (int64_t)mm * p.y_sm);
    const int64_t rrow = rbz + (p.r_mi ? (int64_t)(mm / p.r_mi) * p.r_so + (int64_t)(mm % p.r_mi) * p.r_sm
                                       : (int64_t)mm * p.r_sm);
    const int64_t arow = abz + (int64_t)mm * p.ldaux;
    const float bias_m = (p.bias && (flags & FFVC_F_BIAS_ALONG_M)) ? p.bias[mm] : 0.0f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + (NSPLIT ? nt * 128 + wn * 32 : wn * 64 + nt * 32) + 8 * q + 4 * h;
        f32x4_t v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][4 * q + j] * p.alpha + bias_m;
        if (mok && n < p.N) {
          if constexpr (VEC_ONLY)
            epilogue_quad<T>(p, v, n, yrow, rrow, arow, flags, true);
          else
            epilogue_quad<T>(p, v, n, yrow, rrow, arow, flags, vec_ok && (n + 3 < p.N));
        }
      }
    }
  }
}

// The same epilogue for 16x16x32 accumulators: acc16[a][b] = W rows (n) block a (16 wide) x X rows (m) block b; the lane owns
// m = 16 b + (lane & 15) and the 4 consecutive n = 16 a + 4 (lane >> 4) .. + 3.
template <typename T, int MT = 2, bool VEC_ONLY = false>
__device__ __forceinline__ void gemm_epilogue16(const ffvc_gemm_desc& p, f32x4_t (&acc)[4][2 * MT], int m0, int n0, int wm,
                                                int wn, int lane, int zo, int zi, int vec_ok, int zs = -1) {
  const int l15 = lane & 15, g4 = lane >> 4;
  const int flags = p.flags;
  if (zs < 0) zs = blockIdx.z;
  const int64_t ybz = zo * p.ybo + zi * p.ybi + (int64_t)zs * p.slab_stride;
  const int64_t rbz = zo * p.rbo + zi * p.rbi;
  const int64_t abz = zo * p.abo + zi * p.abi;
#pragma unroll
  for (int b = 0; b < 2 * MT; ++b) {
    const int m = m0 + wm * (32 * MT) + b * 16 + l15;
    const bool mok = m < p.M;
    const int mm = mok ? m : 0;
    const int64_t yrow = ybz + (p.y_mi ? (int64_t)(mm / p.y_mi) * p.y_so + (int64_t)(mm % p.y_mi) * p.y_sm
                                       : (int64_t)mm * p.y_sm);
    const int64_t rrow = rbz + (p.r_mi ? (int64_t)(mm / p.r_mi) * p.r_so + (int64_t)(mm % p.r_mi) * p.r_sm
                                       : (int64_t)mm * p.r_sm);
    const int64_t arow = abz + (int64_t)mm * p.ldaux;
    const float bias_m = (p.bias && (flags & FFVC_F_BIAS_ALONG_M)) ? p.bias[mm] : 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = n0 + wn * 64 + a * 16 + 4 * g4;
      f32x4_t v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = acc[a][b][j] * p.alpha + bias_m;
      if (mok && n < p.N) {
        if constexpr (VEC_ONLY)
          epilogue_quad<T>(p, v, n, yrow, rrow, arow, flags, true);
        else
          epilogue_quad<T>(p, v, n, yrow, rrow, arow, flags, vec_ok && (n + 3 < p.N));
      }
    }
  }
}

// FFVC_F_VQ_ARGMIN (main.py:133-139): acc16[a][b] as in gemm_epilogue16 — the lane owns row m = 16 b + (lane & 15) and the columns
// n = 16 a + 4 (lane >> 4) + j.  d = (xn[m] + cn[n]) - 2 acc is the expression of vq_argmin_kernel (elementwise.hip) on the value the
// plain fp32 epilogue would have stored (alpha = 1), so the winner is the same index bit for bit.  Per row: 16 columns in the lane
// (ascending n, strict < keeps the first), the other 48 of the wave in three lanes (two exchanges), then one 64-bit atomic minimum of
// (order-preserving bits of d) << 32 | n — the unsigned order of that word is (d, n) lexicographic, so the first minimum over all
// column tiles wins however the atomics arrive.
__device__ __forceinline__ unsigned long long vq_pack(float d, int n) {
  const uint32_t u = __float_as_uint(d);
  const uint32_t key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)key << 32) | (uint32_t)n;
}
template <int MT>
__device__ __forceinline__ void gemm_epilogue_vq16(const ffvc_gemm_desc& p, f32x4_t (&acc)[4][2 * MT], int m0, int n0, int wm, int wn,
                                                   int lane) {
  const int l15 = lane & 15, g4 = lane >> 4;
  float cn[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + a * 16 + 4 * g4 + j;
      cn[a][j] = n < p.N ? p.vq_cn[n] : 0.0f;
    }
#pragma unroll
  for (int b = 0; b < 2 * MT; ++b) {
    const int m = m0 + wm * (32 * MT) + b * 16 + l15;
    const float x2 = p.vq_xn[m < p.M ? m : 0];
    float best = INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + a * 16 + 4 * g4 + j;
        const float d = (x2 + cn[a][j]) - 2.0f * acc[a][b][j];
        if (n < p.N && d < best) {
          best = d;
          bi = n;
        }
      }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob < best || (ob == best && oi < bi)) {
        best = ob;
        bi = oi;
      }
    }
    if (g4 == 0 && m < p.M && bi != 0x7fffffff) atomicMin((unsigned long long*)p.vq_out + m, vq_pack(best, bi));
  }
}

// ---- row-store epilogue --------------------------------------------------------------------------------------------
// The MFMA C layout gives a lane 4 consecutive n of ONE row per register quad, 32 rows per wave instruction: stored
// directly, every store touches 32 lines with 16 bytes each and the epilogue is store-issue bound (~7 B/clk/CU
// measured).  Here each 32x32 block goes through a wave-private 4 KiB LDS pad (XOR-swizzled 16-byte chunks, no bank
// conflicts either way) and comes back as 8 consecutive n per lane, 16 rows per instruction: one 16-byte store (bf16)
// per lane, aux / residual reads coalesced the same way, activations evaluated two-at-a-time.
template <typename T>
__device__ __forceinline__ f32x2_t apply_act2(int act, f32x2_t v) {
  if constexpr (sizeof(T) == 2) {
    if (act == FFVC_ACT_GELU) return act_gelu_fast2(v);
    if (act == FFVC_ACT_QUICKGELU) return act_quickgelu_fast2(v);
    return v;         // LeakyReLU / tanh exist only on the fp32 path (the Net2Net prior): 16 unrolled tanhf copies are bloat
  } else {
    f32x2_t r;
    r[0] = apply_act<T>(act, v[0]);
    r[1] = apply_act<T>(act, v[1]);
    return r;
  }
}
template <typename T>
__device__ __forceinline__ f32x2_t apply_act_grad2(int act, f32x2_t pre) {
  if constexpr (sizeof(T) == 2) {
    if (act == FFVC_ACT_GELU) return act_gelu_grad_fast2(pre);
    if (act == FFVC_ACT_QUICKGELU) return act_quickgelu_grad_fast2(pre);
    f32x2_t one = {1.0f, 1.0f};
    return one;
  } else {
    f32x2_t r;
    r[0] = apply_act_grad<T>(act, pre[0]);
    r[1] = apply_act_grad<T>(act, pre[1]);
    return r;
  }
}

// activation a and derivative d of the pre-activation x in one go (shared erf / exp / sigmoid); 16-bit kernels only
template <typename T>
__device__ __forceinline__ void apply_act_both2(int act, f32x2_t x, f32x2_t& a, f32x2_t& d) {
  if (act == FFVC_ACT_GELU) {
    f32x2_t cdf, e;
    gelu_parts_fast2(x, cdf, e);
    a = x * cdf;
    d = cdf + x * 0.39894228040143267794f * e;
  } else {
    const f32x2_t s = sigmoid_fast2(x * 1.702f);
    a = x * s;
    d = s * (1.0f + 1.702f * x * (1.0f - s));
  }
}

// EPI: which optional epilogue code a kernel instantiation carries (the row-store epilogue is unrolled 4 * MT times per
// kernel, and its size is not free: adding an activation variant to every copy cost the whole step 10 %, removing the
// activation code from kernels that never use it bought 7 % on plain GEMMs — profiles/r02_epilogue_code_size.txt).
//   EPI_ACT   activation / activation-gradient / pre-activation write / bias-gradient column sums
//   EPI_GN    GroupNorm moment accumulation
//   EPI_K_*   ONE activation fixed at compile time (no dynamic switch, no other activation's code): the Mixer / ViT MLP
//             launches (GELU or QuickGELU, forward with optional pre-activation write, or activation-gradient + column sums)
constexpr int EPI_ACT = 1, EPI_GN = 2, EPI_ALL = 3, EPI_LEAN = 0;
constexpr int EPI_K_GELU_FWD = 4, EPI_K_QGELU_FWD = 8, EPI_K_GELU_BWD = 16, EPI_K_QGELU_BWD = 32;
constexpr int EPI_K_GELU_FWDG = 64, EPI_K_QGELU_FWDG = 128;   // forward that stores act'(pre) (FFVC_F_AUX_ACTGRAD)
constexpr int EPI_K_MULAUX = 256;                              // backward: acc *= aux (aux already holds act'(pre))
constexpr int EPI_K_FWD = EPI_K_GELU_FWD | EPI_K_QGELU_FWD, EPI_K_BWD = EPI_K_GELU_BWD | EPI_K_QGELU_BWD;
constexpr int EPI_K_FWDG = EPI_K_GELU_FWDG | EPI_K_QGELU_FWDG;
// The kinds also fix the REST of the epilogue (the launcher checks it): forward kinds = column bias, no residual, output in the
// 16-bit storage type; backward kinds = no bias, no residual, 16-bit output.  No flag tests are left in their unrolled copies.
constexpr int EPI_K_ANY_FWD = EPI_K_FWD | EPI_K_FWDG, EPI_K_ANY_BWD = EPI_K_BWD | EPI_K_MULAUX;
constexpr int EPI_K_ANY = EPI_K_ANY_FWD | EPI_K_ANY_BWD;
// Output classes of the lean kernels (again launcher-checked): 16-bit plain store without residual (dgrads, qkv), fp32 store
// with an fp32 residual (the projections back into the fp32 residual stream), plain fp32 store (weight-gradient slabs).
constexpr int EPI_O_T = 512, EPI_O_F32R = 1024, EPI_O_F32 = 2048;
// fp8 output of the MLP kinds (frozen towers, ffvc_gemm_fp8): y gets e4m3 / e5m2 bytes scaled by p.y8_state[0]; the running amax goes
// to p.y8_state[1] (desc fields y8_state / y8_fmt)
constexpr int EPI_O_F8E4 = 4096, EPI_O_F8E5 = 8192, EPI_O_F8 = EPI_O_F8E4 | EPI_O_F8E5;
// GroupNorm-BACKWARD statistics of the node whose output gradient this launch stores (FFVC_F_GNB_SUMS, register-exchange epilogue only)
constexpr int EPI_GNB = 16384;
constexpr int EPI_K_VQ = 32768;          // FFVC_F_VQ_ARGMIN: no store, per-row argmin of the distance (gemm_epilogue_vq16)

template <typename T, int EPI = EPI_ALL>
__device__ __forceinline__ void epilogue_oct(const ffvc_gemm_desc& p, f32x8& v, int n, int64_t yrow, int64_t rrow,
                                             int64_t arow, int flags) {
  if constexpr ((EPI & EPI_K_ANY_FWD) != 0) {
    const f32x8 b = load8(p.bias + n);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] += b.v[j];
  } else if constexpr ((EPI & EPI_K_ANY_BWD) != 0) {
    // no bias
  } else if (p.bias && !(flags & FFVC_F_BIAS_ALONG_M)) {
    const f32x8 b = load8(p.bias + n);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] += b.v[j];
  }
  if constexpr ((EPI & EPI_K_FWDG) != 0) {
    constexpr int A = (EPI & EPI_K_GELU_FWDG) ? FFVC_ACT_GELU : FFVC_ACT_QUICKGELU;
    f32x8 g;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      f32x2_t a, dv;
      apply_act_both2<T>(A, f32x2_t{v.v[j], v.v[j + 1]}, a, dv);
      v.v[j] = a[0];
      v.v[j + 1] = a[1];
      g.v[j] = dv[0];
      g.v[j + 1] = dv[1];
    }
    store8((T*)p.aux + arow + n, g);
  } else if constexpr ((EPI & EPI_K_MULAUX) != 0) {
    const f32x8 pre = load8((const T*)p.aux + arow + n);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] *= pre.v[j];
  } else if constexpr ((EPI & EPI_K_FWD) != 0) {
    constexpr int A = (EPI & EPI_K_GELU_FWD) ? FFVC_ACT_GELU : FFVC_ACT_QUICKGELU;
    if (flags & FFVC_F_WRITE_PREACT) store8((T*)p.aux + arow + n, v);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2_t a = apply_act2<T>(A, f32x2_t{v.v[j], v.v[j + 1]});
      v.v[j] = a[0];
      v.v[j + 1] = a[1];
    }
  } else if constexpr ((EPI & EPI_K_BWD) != 0) {
    constexpr int A = (EPI & EPI_K_GELU_BWD) ? FFVC_ACT_GELU : FFVC_ACT_QUICKGELU;
    const f32x8 pre = load8((const T*)p.aux + arow + n);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2_t g = apply_act_grad2<T>(A, f32x2_t{pre.v[j], pre.v[j + 1]});
      v.v[j] *= g[0];
      v.v[j + 1] *= g[1];
    }
  } else if constexpr (!(EPI & EPI_ACT)) {
    // lean instantiation: no activation code at all
  } else if (flags & FFVC_F_MUL_ACT_GRAD) {
    const f32x8 pre = load8((const T*)p.aux + arow + n);
    if (flags & FFVC_F_AUX_ACTGRAD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] *= pre.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const f32x2_t g = apply_act_grad2<T>(p.act, f32x2_t{pre.v[j], pre.v[j + 1]});
        v.v[j] *= g[0];
        v.v[j + 1] *= g[1];
      }
    }
  } else if (p.act != FFVC_ACT_NONE) {
    if (flags & FFVC_F_WRITE_PREACT) store8((T*)p.aux + arow + n, v);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2_t a = apply_act2<T>(p.act, f32x2_t{v.v[j], v.v[j + 1]});
      v.v[j] = a[0];
      v.v[j + 1] = a[1];
    }
  }
  if constexpr ((EPI & (EPI_K_ANY | EPI_O_T)) != 0) {
    if constexpr ((EPI & EPI_O_F8) == 0) store8((T*)p.y + yrow + n, v);      // (fp8 output: the caller converts and stores v)
    return;
  } else if constexpr ((EPI & EPI_O_F32R) != 0) {
    const f32x8 r = load8((const float*)p.residual + rrow + n);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] += r.v[j];
    store8((float*)p.y + yrow + n, v);
    return;
  } else if constexpr ((EPI & EPI_O_F32) != 0) {
    store8((float*)p.y + yrow + n, v);
    return;
  }
  if (p.residual) {
    const f32x8 r = (flags & FFVC_F_RES_F32) ? load8((const float*)p.residual + rrow + n) : load8((const T*)p.residual + rrow + n);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] += r.v[j];
  }
  if (flags & FFVC_F_ATOMIC_OUT) {
    float* yp = (float*)p.y + yrow + n;
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(yp + j, v.v[j]);
  } else if (flags & FFVC_F_ACCUM_OUT) {
    float* yp = (float*)p.y + yrow + n;
    const f32x8 o = load8(yp);
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] += o.v[j];
    store8(yp, v);
  } else if (flags & FFVC_F_OUT_F32) {
    store8((float*)p.y + yrow + n, v);
  } else {
    store8((T*)p.y + yrow + n, v);
  }
}

// pad: this wave's 4 KiB of LDS.  Requires N % 8 == 0 and 16-byte aligned rows of y / aux / residual (host-checked).
template <typename T, int MT, bool NSPLIT, int EPI, typename WR>
__device__ __forceinline__ void gemm_epilogue_rows_impl(const ffvc_gemm_desc& p, WR&& write_block, int m0, int n0,
                                                        int wm, int wn, int lane, int zo, int zi, unsigned char* pad,
                                                        int zs) {
  const int l31 = lane & 31, h = lane >> 5;
  const int rr = lane >> 2, cc = lane & 3;
  const int flags = p.flags;
  if (zs < 0) zs = blockIdx.z;
  const int64_t ybz = zo * p.ybo + zi * p.ybi + (int64_t)zs * p.slab_stride;
  const int64_t rbz = zo * p.rbo + zi * p.rbi;
  const int64_t abz = zo * p.abo + zi * p.abi;
  (void)l31;
  (void)h;
  const bool gn = (EPI & EPI_GN) && (flags & FFVC_F_GN_SUMS);
  float gs1[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gs2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // [nt][4-channel half]
  float f8_amax = 0.0f, f8_scale = 1.0f;
  if constexpr ((EPI & EPI_O_F8) != 0) f8_scale = p.y8_state[0];
  (void)f8_amax;
  (void)f8_scale;
  const bool cs_on = (EPI & (EPI_ACT | EPI_K_BWD | EPI_K_MULAUX)) && (flags & FFVC_F_COLSUM);
  float cs[2][8];                                                                      // [nt][column of this lane]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[nt][j] = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int mbase = m0 + wm * (32 * MT) + mt * 32;
    int64_t yrow[2], rrow[2], arow[2];
    bool mok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mbase + rr + 16 * i;
      mok[i] = m < p.M;
      const int mm = mok[i] ? m : 0;
      yrow[i] = ybz + (p.y_mi ? (int64_t)(mm / p.y_mi) * p.y_so + (int64_t)(mm % p.y_mi) * p.y_sm : (int64_t)mm * p.y_sm);
      rrow[i] = rbz + (p.r_mi ? (int64_t)(mm / p.r_mi) * p.r_so + (int64_t)(mm % p.r_mi) * p.r_sm : (int64_t)mm * p.r_sm);
      arow[i] = abz + (int64_t)mm * p.ldaux;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      write_block(nt, mt, mbase);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int n = n0 + (NSPLIT ? nt * 128 + wn * 32 : wn * 64 + nt * 32) + 8 * cc;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = rr + 16 * i;
        const unsigned char* rd = pad + row * 128;
        const f32x4_t a = *(const f32x4_t*)(rd + (((2 * cc) ^ (row & 7)) << 4));
        const f32x4_t b = *(const f32x4_t*)(rd + (((2 * cc + 1) ^ (row & 7)) << 4));
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v.v[j] = a[j];
          v.v[4 + j] = b[j];
        }
        if (mok[i] && n < p.N) {
          epilogue_oct<T, EPI>(p, v, n, yrow[i], rrow[i], arow[i], flags);
          if constexpr ((EPI & EPI_O_F8) != 0) {
            constexpr float LIM = (EPI & EPI_O_F8E4) ? 448.0f : 57344.0f;
            float q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              f8_amax = fmaxf(f8_amax, fabsf(v.v[j]));
              const float s = v.v[j] * f8_scale;
              q[j] = s != s ? s : fminf(fmaxf(s, -LIM), LIM);                 // saturate, keep NaN a NaN
            }
            int lo = 0, hi = 0;
            if constexpr ((EPI & EPI_O_F8E4) != 0) {
              lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], lo, false);
              lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
              hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], hi, false);
              hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
            } else {
              lo = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], lo, false);
              lo = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], lo, true);
              hi = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], hi, false);
              hi = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], hi, true);
            }
            *(u32x2_t*)((unsigned char*)p.y + yrow[i] + n) = u32x2_t{(uint32_t)lo, (uint32_t)hi};
          }
          if (gn) {   // moments of the fp32 values before the bf16 store (the rounding noise adds ~1e-6 of E[x^2])
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              gs1[nt][j >> 2] += v.v[j];
              gs2[nt][j >> 2] += v.v[j] * v.v[j];
            }
          }
          if (cs_on) {   // bias gradient of the layer whose pre-activation gradient this GEMM just produced
#pragma unroll
            for (int j = 0; j < 8; ++j) cs[nt][j] += v.v[j];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if constexpr ((EPI & EPI_O_F8) != 0) {
    // running amax of the tensor just written (sets the scale of the NEXT step): one atomic per wave; values >= 0, so the
    // unsigned order of the bit patterns is the float order
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) f8_amax = fmaxf(f8_amax, __shfl_xor(f8_amax, o, 64));
    if (lane == 0 && f8_amax > 0.0f) atomicMax((unsigned int*)(p.y8_state + 1), __float_as_uint(f8_amax));
  }
  if (cs_on) {
    // column sums of everything this wave stored: fold the 16 row-lanes, one fp32 atomic per column and wave
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int n = n0 + (NSPLIT ? nt * 128 + wn * 32 : wn * 64 + nt * 32) + 8 * cc;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = cs[nt][j];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
        if (rr == 0 && n < p.N) atomicAdd(p.colsum + n + j, a);
      }
    }
  }
  if (gn) {
    // lanes with the same (lane & 3) own the same 8 columns: fold the 16 row-lanes, then one fp64 atomic pair per
    // (column half, nt) into sums[image][group][2]; a tile lies inside one image (gn_hw is a multiple of the tile rows)
    const int64_t img = (int64_t)(m0 / p.gn_hw) * (p.N / p.gn_cpg);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float a = gs1[nt][hf], b = gs2[nt][hf];
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        const int n = n0 + (NSPLIT ? nt * 128 + wn * 32 : wn * 64 + nt * 32) + 8 * cc + 4 * hf;
        if (rr == 0 && n < p.N) {
          double* o2 = p.gn_sums + (img + n / p.gn_cpg) * 2;
          atomicAdd(o2, (double)a);
          atomicAdd(o2 + 1, (double)b);
        }
      }
  }
}

// 32x32x16 accumulators: the lane owns row (lane & 31) of the block and the 4 consecutive n = 8 q + 4 (lane >> 5) .. + 3 per quad q
template <typename T, int MT, bool NSPLIT = false, int EPI = EPI_ALL>
__device__ __forceinline__ void gemm_epilogue_rows(const ffvc_gemm_desc& p, f32x16_t (&acc)[2][MT], int m0, int n0,
                                                   int wm, int wn, int lane, int zo, int zi, unsigned char* pad,
                                                   int zs = -1) {
  const int l31 = lane & 31, h = lane >> 5;
  unsigned char* wr = pad + l31 * 128;
  const int wsw = l31 & 7;
  gemm_epilogue_rows_impl<T, MT, NSPLIT, EPI>(
      p,
      [&](int nt, int mt, int mbase) {
        float bias_m = 0.0f;
        if constexpr ((EPI & EPI_K_ANY) == 0)
          if (p.bias && (p.flags & FFVC_F_BIAS_ALONG_M)) bias_m = p.bias[min(mbase + l31, p.M - 1)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4_t v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][4 * q + j] * p.alpha + bias_m;
          *(f32x4_t*)(wr + (((2 * q + h) ^ wsw) << 4)) = v;
        }
      },
      m0, n0, wm, wn, lane, zo, zi, pad, zs);
}

// 16x16x32 accumulators (acc16[a][b], see gemm_epilogue16): the 32x32 block (nt, mt) is the four accumulators
// a = 2 nt + i, b = 2 mt + j; the lane writes row 16 j + (lane & 15), 16-byte chunk 4 i + (lane >> 4) of the same pad image.
template <typename T, int MT, int EPI = EPI_ALL>
__device__ __forceinline__ void gemm_epilogue_rows16(const ffvc_gemm_desc& p, f32x4_t (&acc)[4][2 * MT], int m0, int n0,
                                                     int wm, int wn, int lane, int zo, int zi, unsigned char* pad,
                                                     int zs = -1) {
  const int l15 = lane & 15, g4 = lane >> 4;
  gemm_epilogue_rows_impl<T, MT, false, EPI>(
      p,
      [&](int nt, int mt, int mbase) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int row = 16 * j + l15;
          float bias_m = 0.0f;
          if constexpr ((EPI & EPI_K_ANY) == 0)
            if (p.bias && (p.flags & FFVC_F_BIAS_ALONG_M)) bias_m = p.bias[min(mbase + row, p.M - 1)];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            f32x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[2 * nt + i][2 * mt + j][e] * p.alpha + bias_m;
            *(f32x4_t*)(pad + row * 128 + (((4 * i + g4) ^ (row & 7)) << 4)) = v;
          }
        }
      },
      m0, n0, wm, wn, lane, zo, zi, pad, zs);
}

// ---- register-exchange epilogue for 16x16x32 accumulators (round 6) --------------------------------------------------
// The row-store epilogue above moves every 32x32 block through a wave-private LDS pad (4 ds_write_b128 + 4 ds_read_b128 per lane
// and block, two wave barriers, 4 KiB of LDS per wave) only to turn "4 consecutive n per lane" into "8 consecutive n per lane".
// For 16x16x32 accumulators ONE v_permlane16_swap per register does the same: acc[a][b] gives lane (l15, g4) the columns
// 16 a + 4 g4 .. + 3 of row 16 b + l15; swapping the odd 16-lane rows of acc[2p][b] with the even rows of acc[2p + 1][b] leaves
// lane group g4 with the 8 consecutive columns 32 p + {0, 16, 8, 24}[g4] .. + 7 (first four in the register that held acc[2p][b],
// the next four in the one that held acc[2p + 1][b]).  One store instruction then covers 16 rows x 64 contiguous bytes (16-bit
// output), exactly the pad version's granularity, with no LDS traffic, no barrier and no 4 KiB pads — which is what lets two
// workgroups of the 256x128 ring kernel (gemm3_kernel) fit a CU and keeps the partner workgroup's fragment reads undisturbed.
// Without fences in the way, the side input of the next 8-column group (aux of the multiply kinds, the fp32 residual of the
// projection kinds) is requested before the current group is finished: the epilogue no longer pays one memory round trip per group.
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_oct_pre(const ffvc_gemm_desc& p, f32x8& v, int n, int64_t yrow, const f32x8& pre) {
  if constexpr ((EPI & EPI_K_MULAUX) != 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] *= pre.v[j];
    store8((T*)p.y + yrow + n, v);
  } else {           // EPI_O_F32R: column bias (when present), + fp32 residual, fp32 store
    if (p.bias) {
      const f32x8 b = load8(p.bias + n);
#pragma unroll
      for (int j = 0; j < 8; ++j) v.v[j] += b.v[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v.v[j] += pre.v[j];
    store8((float*)p.y + yrow + n, v);
  }
}

template <typename T, int MT, int EPI = EPI_ALL>
__device__ __forceinline__ void gemm_epilogue_perm16(const ffvc_gemm_desc& p, f32x4_t (&acc)[4][2 * MT], int m0, int n0, int wm,
                                                     int wn, int lane, int zo, int zi, int zs = -1) {
  const int l15 = lane & 15, g4 = lane >> 4;
  const int cofs = ((g4 & 1) << 4) | ((g4 & 2) << 2);          // {0, 16, 8, 24}[g4]
  const int flags = p.flags;
  if (zs < 0) zs = blockIdx.z;
  const int64_t ybz = zo * p.ybo + zi * p.ybi + (int64_t)zs * p.slab_stride;
  const int64_t rbz = zo * p.rbo + zi * p.rbi;
  const int64_t abz = zo * p.abo + zi * p.abi;
  const bool gn = (EPI & EPI_GN) && (flags & FFVC_F_GN_SUMS);
  float gs1[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gs2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};   // [p][4-channel half]
  float f8_amax = 0.0f, f8_scale = 1.0f;
  if constexpr ((EPI & EPI_O_F8) != 0) f8_scale = p.y8_state[0];
  (void)f8_amax;
  (void)f8_scale;
  const bool cs_on = (EPI & (EPI_ACT | EPI_K_BWD | EPI_K_MULAUX)) && (flags & FFVC_F_COLSUM);
  float cs[2][8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[q][j] = 0.f;
  constexpr bool GNB = (EPI & EPI_GNB) != 0;
  constexpr bool PRE = (EPI & (EPI_K_MULAUX | EPI_O_F32R)) != 0;        // kinds with exactly one streamed side input
  constexpr bool UNIT_ALPHA = (EPI & EPI_K_ANY) != 0;                   // the activation kinds are launched with alpha == 1 only
  // Row offsets once per 16-row block (not per 8-column group): with the plain row maps (y_mi == r_mi == 0, every launch of the
  // step) block b is block 0 plus b * 16 rows — one 64-bit add instead of three 64-bit multiplies
  const bool plain_maps = p.y_mi == 0 && p.r_mi == 0;                   // wave-uniform
  const int mrow0 = m0 + wm * (32 * MT) + l15;
  const int64_t y_first = ybz + (int64_t)mrow0 * p.y_sm, r_first = rbz + (int64_t)mrow0 * p.r_sm, a_first = abz + (int64_t)mrow0 * p.ldaux;
  const int64_t y_step = 16 * (int64_t)p.y_sm, r_step = 16 * (int64_t)p.r_sm, a_step = 16 * (int64_t)p.ldaux;
  struct RowOff {
    int64_t y, r, a;
    bool ok;
  };
  auto rows_of = [&](int b) -> RowOff {
    RowOff o;
    const int m = mrow0 + 16 * b;
    o.ok = m < p.M;
    if (plain_maps) {
      o.y = y_first + b * y_step;          // rows beyond M are never dereferenced (o.ok guards every access)
      o.r = r_first + b * r_step;
    } else {
      const int mm = o.ok ? m : 0;
      o.y = ybz + (p.y_mi ? (int64_t)(mm / p.y_mi) * p.y_so + (int64_t)(mm % p.y_mi) * p.y_sm : (int64_t)mm * p.y_sm);
      o.r = rbz + (p.r_mi ? (int64_t)(mm / p.r_mi) * p.r_so + (int64_t)(mm % p.r_mi) * p.r_sm : (int64_t)mm * p.r_sm);
    }
    o.a = a_first + b * a_step;
    return o;
  };
  const int ncol0 = n0 + wn * 64 + cofs;
  auto side = [&](const RowOff& ro, int pp) -> f32x8 {
    f32x8 r;
    const int n = ncol0 + 32 * pp;
    if (ro.ok && n < p.N) {
      if constexpr ((EPI & EPI_K_MULAUX) != 0) r = load8((const T*)p.aux + ro.a + n);
      else r = load8((const float*)p.residual + ro.r + n);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[j] = 0.f;
    }
    return r;
  };
  // GNB: per-lane constants of its 16 columns — gamma / beta per channel, (rstd, -mean * rstd) per 4-channel half (gn_cpg % 4 == 0: four
  // aligned consecutive channels share a group); a tile lies inside one image (gn_hw % 256 == 0)
  float gb_g[2][8], gb_b[2][8], gb_r[2][2], gb_c[2][2];
  f32x2_t acc_a1[2][2], acc_a2[2][2];
  if constexpr (GNB) {
    const int groups = p.N / p.gn_cpg;
    const int64_t img = (int64_t)(m0 / p.gn_hw) * groups;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = ncol0 + 32 * q;
      const bool nok = n < p.N;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gb_g[q][j] = nok ? p.gnb_gamma[n + j] : 0.f;
        gb_b[q][j] = nok ? p.gnb_beta[n + j] : 0.f;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int g = nok ? (n + 4 * hf) / p.gn_cpg : 0;
        const float rs = p.gnb_rstd[img + g], mu = p.gnb_mean[img + g];
        gb_r[q][hf] = rs;
        gb_c[q][hf] = -mu * rs;
        acc_a1[q][hf] = acc_a2[q][hf] = f32x2_t{0.f, 0.f};
      }
    }
  }
  // GNB: the wave's x chunks are requested GD groups ahead (16 bytes per lane and group, raw: the fragment registers are free now) —
  // with a prefetch distance of one group every group waited ~1 us on its load and the fusion cost 230 us per 256^2 launch
  constexpr int GD = 4;
  u32x4_t gb_raw[GNB ? GD : 1];
  auto gnb_fetch = [&](int it) -> u32x4_t {
    if constexpr (GNB && sizeof(T) == 2) {
      const RowOff ro = rows_of(it >> 1);
      const int n = ncol0 + 32 * (it & 1);
      return (ro.ok && n < p.N) ? *(const u32x4_t*)((const T*)p.gnb_x + ro.y + n) : u32x4_t{0u, 0u, 0u, 0u};
    } else {
      return u32x4_t{0u, 0u, 0u, 0u};
    }
  };
  if constexpr (GNB) {
#pragma unroll
    for (int it = 0; it < GD; ++it) gb_raw[it] = gnb_fetch(it);
  }
  RowOff rcur = rows_of(0), rnxt = rcur;
  f32x8 pre_cur, pre_nxt;
  if constexpr (PRE) pre_cur = side(rcur, 0);
#pragma unroll
  for (int b = 0; b < 2 * MT; ++b) {
    if (b + 1 < 2 * MT) rnxt = rows_of(b + 1);
    float bias_m = 0.0f;
    if constexpr ((EPI & (EPI_K_ANY | EPI_O_T | EPI_O_F32R | EPI_O_F32)) == 0)
      if (p.bias && (flags & FFVC_F_BIAS_ALONG_M)) bias_m = p.bias[min(mrow0 + 16 * b, p.M - 1)];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      if constexpr (PRE) {
        if (pp == 0) pre_nxt = side(rcur, 1);
        else if (b + 1 < 2 * MT) pre_nxt = side(rnxt, 0);
      }
      const int n = ncol0 + 32 * pp;
      f32x8 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // bias along M is a per-row constant: it may be added before the exchange (the row does not change, only the columns)
        float x0 = acc[2 * pp][b][e], x1 = acc[2 * pp + 1][b][e];
        if constexpr (!UNIT_ALPHA) {
          x0 = x0 * p.alpha + bias_m;
          x1 = x1 * p.alpha + bias_m;
        }
        const u32x2_t s = __builtin_amdgcn_permlane16_swap(__float_as_uint(x0), __float_as_uint(x1), false, false);
        v.v[e] = __uint_as_float(s[0]);
        v.v[4 + e] = __uint_as_float(s[1]);
      }
      if (rcur.ok && n < p.N) {
        if constexpr (GNB && sizeof(T) == 2) {
          epilogue_oct<T, EPI>(p, v, n, rcur.y, rcur.r, rcur.a, flags);      // the ordinary store of dy (bias / residual as asked)
          pre_cur = unpack8<T>(gb_raw[(2 * b + pp) % GD]);
          if (2 * b + pp + GD < 4 * MT) gb_raw[(2 * b + pp) % GD] = gnb_fetch(2 * b + pp + GD);
          // two columns at a time on packed fp32 (v_pk_fma / v_pk_mul: the epilogue's VALU time is what the fusion costs — 4 cycles per
          // wave instruction, 16 per transcendental); the fp32 value is used as it is (its 16-bit rounding moves a sum by ~1e-4 of
          // the rounding noise of its terms)
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            const f32x2_t xv = {pre_cur.v[j], pre_cur.v[j + 1]}, gm = {gb_g[pp][j], gb_g[pp][j + 1]}, bt = {gb_b[pp][j], gb_b[pp][j + 1]};
            const f32x2_t xh = xv * gb_r[pp][j >> 2] + gb_c[pp][j >> 2];
            f32x2_t d = f32x2_t{v.v[j], v.v[j + 1]} * gm;
            if (p.gnb_swish) {
              const f32x2_t yv = xh * gm + bt;
              const f32x2_t sg = sigmoid_fast2(yv);
              d *= sg * (1.0f + yv * (1.0f - sg));
            }
            acc_a1[pp][j >> 2] += d;
            acc_a2[pp][j >> 2] += d * xh;
          }
        } else if constexpr (PRE) epilogue_oct_pre<T, EPI>(p, v, n, rcur.y, pre_cur);
        else epilogue_oct<T, EPI>(p, v, n, rcur.y, rcur.r, rcur.a, flags);
        if constexpr ((EPI & EPI_O_F8) != 0) {
          constexpr float LIM = (EPI & EPI_O_F8E4) ? 448.0f : 57344.0f;
          float q[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            f8_amax = fmaxf(f8_amax, fabsf(v.v[j]));
            const float sq = v.v[j] * f8_scale;
            q[j] = sq != sq ? sq : fminf(fmaxf(sq, -LIM), LIM);
          }
          int lo = 0, hi = 0;
          if constexpr ((EPI & EPI_O_F8E4) != 0) {
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], hi, true);
          } else {
            lo = __builtin_amdgcn_cvt_pk_bf8_f32(q[0], q[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_bf8_f32(q[2], q[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_bf8_f32(q[4], q[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_bf8_f32(q[6], q[7], hi, true);
          }
          *(u32x2_t*)((unsigned char*)p.y + rcur.y + n) = u32x2_t{(uint32_t)lo, (uint32_t)hi};
        }
        if (gn) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            gs1[pp][j >> 2] += v.v[j];
            gs2[pp][j >> 2] += v.v[j] * v.v[j];
          }
        }
        if (cs_on) {
#pragma unroll
          for (int j = 0; j < 8; ++j) cs[pp][j] += v.v[j];
        }
      }
      if constexpr (PRE) pre_cur = pre_nxt;
    }
    rcur = rnxt;
  }
  if constexpr ((EPI & EPI_O_F8) != 0) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) f8_amax = fmaxf(f8_amax, __shfl_xor(f8_amax, o, 64));
    if (lane == 0 && f8_amax > 0.0f) atomicMax((unsigned int*)(p.y8_state + 1), __float_as_uint(f8_amax));
  }
  if constexpr (GNB) {
    const int groups = p.N / p.gn_cpg;
    const int64_t img = (int64_t)(m0 / p.gn_hw) * groups;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float a = acc_a1[q][hf][0] + acc_a1[q][hf][1], b = acc_a2[q][hf][0] + acc_a2[q][hf][1];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        const int n = ncol0 + 32 * q + 4 * hf;
#if defined(FFVC_GN_EXP) && FFVC_GN_EXP == 1       // timing experiment (wrong results): no statistics atomics
        if (l15 == 0 && n < p.N && p.gn_cpg == 12345) {
#else
        if (l15 == 0 && n < p.N) {
#endif
          double* o2 = p.gnb_sums + (img + n / p.gn_cpg) * 2;
          atomicAdd(o2, (double)a);
          atomicAdd(o2 + 1, (double)b);
        }
      }
  }
  if (cs_on) {
    // column sums of everything this wave stored: the 16 lanes of a lane group own the same 8 columns -> fold them, one fp32
    // atomic per column and wave
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = n0 + wn * 64 + 32 * q + cofs;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = cs[q][j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o, 64);
        if (l15 == 0 && n < p.N) atomicAdd(p.colsum + n + j, a);
      }
    }
  }
  if (gn) {
    const int64_t img = (int64_t)(m0 / p.gn_hw) * (p.N / p.gn_cpg);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float a = gs1[q][hf], b = gs2[q][hf];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        const int n = n0 + wn * 64 + 32 * q + cofs + 4 * hf;
#if defined(FFVC_GN_EXP) && FFVC_GN_EXP == 1       // timing experiment (wrong results): no moment atomics
        if (l15 == 0 && n < p.N && p.gn_cpg == 12345) {
#else
        if (l15 == 0 && n < p.N) {
#endif
          double* o2 = p.gn_sums + (img + n / p.gn_cpg) * 2;
          atomicAdd(o2, (double)a);
          atomicAdd(o2 + 1, (double)b);
        }
      }
  }
}

// Which epilogue the 16x16x32 kernels use (measured, profiles/r06_epilogue_ab.txt): the register exchange wins on the convolution
// kernels (+2.5-3 % isolated on every decoder level: plain store + GroupNorm moments, and no pads to alias), the LDS pads stay on the
// K-major x K-major GEMMs (the exchange is faster on the forward activation kinds and the fp32 residual projections, slower on the
// aux-multiply kinds, and the step as a whole was 0.6 ms slower with it).  -DFFVC_EPI_PERM=0: pads everywhere (rounds 1-5);
// -DFFVC_EPI_PERM_NT=1: exchange everywhere (A/B builds).
#ifndef FFVC_EPI_PERM
#define FFVC_EPI_PERM 1
#endif
#ifndef FFVC_EPI_PERM_NT
#define FFVC_EPI_PERM_NT 0
#endif
template <typename T, int MT, int EPI = EPI_ALL, bool PERM = (FFVC_EPI_PERM != 0)>
__device__ __forceinline__ void gemm_epilogue_out16(const ffvc_gemm_desc& p, f32x4_t (&acc)[4][2 * MT], int m0, int n0, int wm,
                                                    int wn, int lane, int zo, int zi, unsigned char* pad, int zs = -1) {
  if constexpr (PERM) {
    (void)pad;
    gemm_epilogue_perm16<T, MT, EPI>(p, acc, m0, n0, wm, wn, lane, zo, zi, zs);
  } else {
    gemm_epilogue_rows16<T, MT, EPI>(p, acc, m0, n0, wm, wn, lane, zo, zi, pad, zs);
  }
}

}  // namespace ffvc_gemm_detail
