// attn_tiny.hip — softmax attention for a handful of tokens and an arbitrary head width (VitGAN mapper, vitgan.py:44-97:
// T = 16 latent tokens, 6 heads of 170 = 1024 // 6 channels; SimpleGenerator: 64 tokens).
//
// As seven batched GEMMs + two softmax launches per block (forward + backward) this is pure launch latency: 192 problems of
// 16 x 170 x 16 (0.6 TFLOP/s, 27 us each, 6.4 ms of the 73 ms cfg3 step), plus the transposes that bring the reference's
// interleaved '(d k h)' projection layout into per-head panels and back.  Here ONE workgroup owns a (sample, head) pair and
// does the whole thing out of LDS in fp32 VALU arithmetic (0.35 MFLOP per pair: matrix cores would not even fill a tile):
//
//   forward :  S = scale q k^T,  P = softmax(S),  o = P v
//   backward:  P recomputed;  dv = P^T do,  dP = do v^T,  dS = P o (dP - rowsum(dP o P)),  dq = scale dS k,  dk = scale dS^T q
//
// q / k / v are addressed through element strides, so the kernel reads the projection output where it lies — either the
// reference's (d k h) interleave (vitgan.py:81-82 rearrange 'b t (d k h) -> k b h t d') or the usual (k h d) — and writes
// dqkv in the same layout; rows may be padded (row stride > valid length, pad zeroed) so that the GEMMs on either side see
// 16-byte aligned rows.
#include "common.h"

namespace {

struct TinyMap {
  int64_t sb, st, sk, sh, sd;   // element strides of qkv[b, t, which, h, d]
};

template <typename T>
__device__ __forceinline__ float ld_elem(const T* p) { return ElemTraits<T>::load(p); }
template <typename T>
__device__ __forceinline__ void st_elem(T* p, float v) { ElemTraits<T>::store(p, v); }

// LDS: q, k, v (and do) as [T][dh + 1] fp32, scores [T][T + 1]
template <typename T>
__global__ __launch_bounds__(256) void attn_tiny_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, TinyMap m, int Tn,
                                                            int heads, int dh, int64_t out_ld, float scale) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int tid = threadIdx.x, ldp = dh + 1, lds_ = Tn + 1;
  float* q = sm;
  float* k = q + Tn * ldp;
  float* v = k + Tn * ldp;
  float* S = v + Tn * ldp;
  const T* base = qkv + b * m.sb + h * m.sh;
  for (int i = tid; i < 3 * Tn * dh; i += 256) {
    // d fastest when it is the contiguous axis, else the token axis (neighbouring lanes then share 32-byte sectors)
    int w, t, d;
    if (m.sd == 1) {
      d = i % dh;
      t = (i / dh) % Tn;
      w = i / (dh * Tn);
    } else {
      w = i % 3;
      d = (i / 3) % dh;
      t = i / (3 * dh);
    }
    sm[w * Tn * ldp + t * ldp + d] = ld_elem(base + t * m.st + w * m.sk + d * m.sd);
  }
  __syncthreads();
  for (int e = tid; e < Tn * Tn; e += 256) {
    const int i = e / Tn, j = e - i * Tn;
    float acc = 0.0f;
    for (int d = 0; d < dh; ++d) acc += q[i * ldp + d] * k[j * ldp + d];
    S[i * lds_ + j] = acc * scale;
  }
  __syncthreads();
  if (tid < Tn) {
    float* row = S + tid * lds_;
    float mx = row[0];
    for (int j = 1; j < Tn; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.0f;
    for (int j = 0; j < Tn; ++j) {
      const float p = __expf(row[j] - mx);
      row[j] = p;
      sum += p;
    }
    const float inv = 1.0f / sum;
    for (int j = 0; j < Tn; ++j) row[j] *= inv;
  }
  __syncthreads();
  T* ob = out + (int64_t)b * Tn * out_ld + h * dh;
  for (int e = tid; e < Tn * dh; e += 256) {
    const int i = e / dh, d = e - i * dh;
    float acc = 0.0f;
    for (int j = 0; j < Tn; ++j) acc += S[i * lds_ + j] * v[j * ldp + d];
    st_elem(ob + (int64_t)i * out_ld + d, acc);
  }
  if (h == heads - 1) {                       // padded output rows: zero the tail once per sample
    const int pad = (int)(out_ld - (int64_t)heads * dh);
    for (int e = tid; e < Tn * pad; e += 256) {
      const int i = e / pad, c = e - i * pad;
      st_elem(out + ((int64_t)b * Tn + i) * out_ld + heads * dh + c, 0.0f);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_tiny_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv,
                                                            TinyMap m, int Tn, int heads, int dh, int64_t out_ld, int64_t row_len,
                                                            float scale) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int tid = threadIdx.x, ldp = dh + 1, lds_ = Tn + 1;
  float* q = sm;
  float* k = q + Tn * ldp;
  float* v = k + Tn * ldp;
  float* go = v + Tn * ldp;
  float* P = go + Tn * ldp;
  float* dS = P + Tn * lds_;
  const T* base = qkv + b * m.sb + h * m.sh;
  for (int i = tid; i < 3 * Tn * dh; i += 256) {
    int w, t, d;
    if (m.sd == 1) {
      d = i % dh;
      t = (i / dh) % Tn;
      w = i / (dh * Tn);
    } else {
      w = i % 3;
      d = (i / 3) % dh;
      t = i / (3 * dh);
    }
    sm[w * Tn * ldp + t * ldp + d] = ld_elem(base + t * m.st + w * m.sk + d * m.sd);
  }
  const T* gb = dout + (int64_t)b * Tn * out_ld + h * dh;
  for (int e = tid; e < Tn * dh; e += 256) {
    const int i = e / dh, d = e - i * dh;
    go[i * ldp + d] = ld_elem(gb + (int64_t)i * out_ld + d);
  }
  __syncthreads();
  for (int e = tid; e < Tn * Tn; e += 256) {
    const int i = e / Tn, j = e - i * Tn;
    float acc = 0.0f, accp = 0.0f;
    for (int d = 0; d < dh; ++d) {
      acc += q[i * ldp + d] * k[j * ldp + d];
      accp += go[i * ldp + d] * v[j * ldp + d];
    }
    P[i * lds_ + j] = acc * scale;
    dS[i * lds_ + j] = accp;               // dP for now
  }
  __syncthreads();
  if (tid < Tn) {
    float* row = P + tid * lds_;
    float* drow = dS + tid * lds_;
    float mx = row[0];
    for (int j = 1; j < Tn; ++j) mx = fmaxf(mx, row[j]);
    float sum = 0.0f;
    for (int j = 0; j < Tn; ++j) {
      const float p = __expf(row[j] - mx);
      row[j] = p;
      sum += p;
    }
    const float inv = 1.0f / sum;
    float dot = 0.0f;
    for (int j = 0; j < Tn; ++j) {
      row[j] *= inv;
      dot += row[j] * drow[j];
    }
    for (int j = 0; j < Tn; ++j) drow[j] = row[j] * (drow[j] - dot) * scale;      // dS (scale folded in)
  }
  __syncthreads();
  T* db = dqkv + b * m.sb + h * m.sh;
  for (int i0 = tid; i0 < 3 * Tn * dh; i0 += 256) {
    int w, t, d;
    if (m.sd == 1) {
      d = i0 % dh;
      t = (i0 / dh) % Tn;
      w = i0 / (dh * Tn);
    } else {
      w = i0 % 3;
      d = (i0 / 3) % dh;
      t = i0 / (3 * dh);
    }
    float acc = 0.0f;
    if (w == 0) {                            // dq[t, d] = sum_j dS[t, j] k[j, d]
      for (int j = 0; j < Tn; ++j) acc += dS[t * lds_ + j] * k[j * ldp + d];
    } else if (w == 1) {                     // dk[t, d] = sum_i dS[i, t] q[i, d]
      for (int i = 0; i < Tn; ++i) acc += dS[i * lds_ + t] * q[i * ldp + d];
    } else {                                 // dv[t, d] = sum_i P[i, t] do[i, d]
      for (int i = 0; i < Tn; ++i) acc += P[i * lds_ + t] * go[i * ldp + d];
    }
    st_elem(db + t * m.st + w * m.sk + d * m.sd, acc);
  }
  if (h == 0 && m.st > row_len) {            // padded projection rows: zero the tail once per sample
    const int pad = (int)(m.st - row_len);
    for (int e = tid; e < Tn * pad; e += 256) {
      const int t = e / pad, c = e - t * pad;
      st_elem(dqkv + b * m.sb + t * m.st + row_len + c, 0.0f);
    }
  }
}

size_t tiny_lds(int T, int dh, bool bwd) {
  return sizeof(float) * ((size_t)(bwd ? 4 : 3) * T * (dh + 1) + (size_t)(bwd ? 2 : 1) * T * (T + 1));
}

template <typename T>
int tiny_launch(bool bwd, const void* qkv, const void* dout, void* out, const TinyMap& m, int B, int Tn, int heads, int dh,
                int64_t out_ld, int64_t row_len, float scale, hipStream_t st) {
  const size_t lds = tiny_lds(Tn, dh, bwd);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_tiny_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)attn_tiny_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr = true;
  }
  if (bwd)
    hipLaunchKernelGGL(attn_tiny_bwd_kernel<T>, dim3(B * heads), dim3(256), lds, st, (const T*)qkv, (const T*)dout, (T*)out, m, Tn,
                       heads, dh, out_ld, row_len, scale);
  else
    hipLaunchKernelGGL(attn_tiny_fwd_kernel<T>, dim3(B * heads), dim3(256), lds, st, (const T*)qkv, (T*)out, m, Tn, heads, dh,
                       out_ld, scale);
  FFVC_LAUNCH_CHECK();
  return 0;
}

int tiny_check(const char* who, int dtype, int B, int T, int heads, int dh, int64_t out_ld, bool bwd) {
  FFVC_CHECK_ARG(dtype == FFVC_BF16 || dtype == FFVC_F16 || dtype == FFVC_F32, "%s: dtype %d", who, dtype);
  FFVC_CHECK_ARG(B > 0 && heads > 0 && T >= 1 && dh >= 1 && (int64_t)B * heads < (1ll << 31), "%s: bad shape", who);
  FFVC_CHECK_ARG(out_ld >= (int64_t)heads * dh, "%s: output row stride %lld < heads * dh", who, (long long)out_ld);
  FFVC_CHECK_ARG(tiny_lds(T, dh, bwd) <= 163840, "%s: T=%d, head_dim=%d do not fit the CU's LDS (use the tiled kernels)", who, T, dh);
  return 0;
}

}  // namespace

extern "C" int ffvc_attn_tiny_supported(int T, int head_dim) {
  return T >= 1 && head_dim >= 1 && tiny_lds(T, head_dim, true) <= 163840 ? 1 : 0;
}

extern "C" int ffvc_attn_tiny_fwd(const void* qkv, void* out, int dtype, int B, int T, int heads, int head_dim, int64_t sb,
                                  int64_t st, int64_t sk, int64_t sh, int64_t sd, int64_t out_ld, float scale, void* stream) {
  FFVC_CHECK_ARG(qkv && out, "ffvc_attn_tiny_fwd: null pointer");
  if (int e = tiny_check("ffvc_attn_tiny_fwd", dtype, B, T, heads, head_dim, out_ld, false)) return e;
  const TinyMap m{sb, st, sk, sh, sd};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FFVC_F16) return tiny_launch<f16_t>(false, qkv, nullptr, out, m, B, T, heads, head_dim, out_ld, 0, scale, s);
  if (dtype == FFVC_BF16) return tiny_launch<uint16_t>(false, qkv, nullptr, out, m, B, T, heads, head_dim, out_ld, 0, scale, s);
  return tiny_launch<float>(false, qkv, nullptr, out, m, B, T, heads, head_dim, out_ld, 0, scale, s);
}

extern "C" int ffvc_attn_tiny_bwd(const void* qkv, const void* dout, void* dqkv, int dtype, int B, int T, int heads, int head_dim,
                                  int64_t sb, int64_t st, int64_t sk, int64_t sh, int64_t sd, int64_t out_ld, int64_t row_len,
                                  float scale, void* stream) {
  FFVC_CHECK_ARG(qkv && dout && dqkv, "ffvc_attn_tiny_bwd: null pointer");
  if (int e = tiny_check("ffvc_attn_tiny_bwd", dtype, B, T, heads, head_dim, out_ld, true)) return e;
  FFVC_CHECK_ARG(row_len > 0 && row_len <= st, "ffvc_attn_tiny_bwd: row_len %lld vs token stride %lld", (long long)row_len, (long long)st);
  const TinyMap m{sb, st, sk, sh, sd};
  hipStream_t s = (hipStream_t)stream;
  if (dtype == FFVC_F16) return tiny_launch<f16_t>(true, qkv, dout, dqkv, m, B, T, heads, head_dim, out_ld, row_len, scale, s);
  if (dtype == FFVC_BF16) return tiny_launch<uint16_t>(true, qkv, dout, dqkv, m, B, T, heads, head_dim, out_ld, row_len, scale, s);
  return tiny_launch<float>(true, qkv, dout, dqkv, m, B, T, heads, head_dim, out_ld, row_len, scale, s);
}
