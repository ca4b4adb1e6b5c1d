// attn_text.hip — exact-fp32 attention for SHORT sequences (round 4): the frozen, forward-only CLIP text tower (77 tokens, causal,
// cloob.py:199-200 nn.MultiheadAttention under cloob.py:510-516's mask; reference main.py:733 evaluates it every step).
//
// The text features enter the loss directly, so the tower is kept at fp32 grade (split-precision Linear layers, fp32 LayerNorm and
// attention).  As batched fp32 GEMMs + a softmax launch the attention cost three launches and 1.4 ms per cfg2 step (77 x 64 x 77
// products at 10 TFLOP/s: the matrices are too small for any tiled kernel).  Here ONE workgroup owns a (prompt, head): K and V of the
// head are staged in LDS once (fp32), thread t owns query row t — its q row and its 64 output accumulators live in registers, the K / V
// rows it walks are LDS broadcasts — and the scores make one round trip through an LDS row of their own: plain two-pass softmax in the
// same order of operations as ffvc_softmax_fwd (max, exp, sum, divide), all fp32 FMAs on the vector ALUs.
#include "common.h"

namespace {

constexpr int AT_TMAX = 128;   // tokens (threads) per workgroup
constexpr int AT_DH = 64;

// qkv: fp32 [B, T, 3 * heads * 64] (q | k | v blocks, heads inside); out: fp32 [B, T, heads * 64].  Dynamic LDS: (2 * T * 64 + T * Tp) floats,
// Tp = T rounded up to 4 (+1 against bank conflicts of the per-thread score rows).
__global__ __launch_bounds__(AT_TMAX) void attn_text_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int heads,
                                                                 float scale, int causal, int Tp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + (size_t)T * AT_DH;
  float* Ss = Vs + (size_t)T * AT_DH;
  const int t = threadIdx.x;
  const int b = blockIdx.x / heads, h = blockIdx.x - b * heads;
  const int D = heads * AT_DH;
  const int64_t ld = 3 * (int64_t)D;
  const float* base = qkv + (int64_t)b * T * ld + h * AT_DH;
  for (int i = t; i < T * (AT_DH / 4); i += AT_TMAX) {
    const int j = i / (AT_DH / 4), c = (i % (AT_DH / 4)) * 4;
    *(f32x4_t*)(Ks + j * AT_DH + c) = *(const f32x4_t*)(base + (int64_t)j * ld + D + c);
    *(f32x4_t*)(Vs + j * AT_DH + c) = *(const f32x4_t*)(base + (int64_t)j * ld + 2 * D + c);
  }
  float q[AT_DH];
  if (t < T) {
#pragma unroll
    for (int c = 0; c < AT_DH; c += 4) {
      const f32x4_t v = *(const f32x4_t*)(base + (int64_t)t * ld + c);
      q[c] = v[0], q[c + 1] = v[1], q[c + 2] = v[2], q[c + 3] = v[3];
    }
  }
  __syncthreads();
  if (t >= T) return;
  const int nk = causal ? t + 1 : T;               // key j visible to query t iff j <= t (cloob.py:510-516)
  float* S = Ss + (size_t)t * Tp;
  float mx = -INFINITY;
  for (int j = 0; j < nk; ++j) {
    const float* kr = Ks + j * AT_DH;
    // four independent partial sums: with one or two waves per SIMD a single 64-long FMA chain is all latency (84 us per launch in the step)
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
    for (int c = 0; c < AT_DH; c += 4) {
      const f32x4_t kv = *(const f32x4_t*)(kr + c);
      s0 = fmaf(q[c], kv[0], s0);
      s1 = fmaf(q[c + 1], kv[1], s1);
      s2 = fmaf(q[c + 2], kv[2], s2);
      s3 = fmaf(q[c + 3], kv[3], s3);
    }
    float s = ((s0 + s1) + (s2 + s3)) * scale;
    S[j] = s;
    mx = fmaxf(mx, s);
  }
  float sum = 0.0f;
  for (int j = 0; j < nk; ++j) {
    const float p = expf(S[j] - mx);
    S[j] = p;
    sum += p;
  }
  const float inv = 1.0f / sum;
  float o[AT_DH];
#pragma unroll
  for (int c = 0; c < AT_DH; ++c) o[c] = 0.0f;
  for (int j = 0; j < nk; ++j) {
    const float p = S[j] * inv;
    const float* vr = Vs + j * AT_DH;
#pragma unroll
    for (int c = 0; c < AT_DH; c += 4) {
      const f32x4_t vv = *(const f32x4_t*)(vr + c);
      o[c] = fmaf(p, vv[0], o[c]);
      o[c + 1] = fmaf(p, vv[1], o[c + 1]);
      o[c + 2] = fmaf(p, vv[2], o[c + 2]);
      o[c + 3] = fmaf(p, vv[3], o[c + 3]);
    }
  }
  float* orow = out + ((int64_t)b * T + t) * D + h * AT_DH;
#pragma unroll
  for (int c = 0; c < AT_DH; c += 4) *(f32x4_t*)(orow + c) = f32x4_t{o[c], o[c + 1], o[c + 2], o[c + 3]};
}

}  // namespace

// out[B, T, heads*64] = softmax(scale * q k^T (+ causal mask)) v per head, everything fp32, T <= 128, head_dim 64; qkv packed as in
// ffvc_attn_small_fwd.  Forward only (the text tower is frozen and takes no gradient: main.py:733 under no_grad semantics of encode_text).
extern "C" int ffvc_attn_text_fwd(const float* qkv, float* out, int B, int T, int heads, int head_dim, float scale, int causal,
                                  void* stream) {
  FFVC_CHECK_ARG(qkv && out && B > 0 && heads > 0 && T > 0, "ffvc_attn_text_fwd: bad args");
  FFVC_CHECK_ARG(head_dim == AT_DH && T <= AT_TMAX, "ffvc_attn_text_fwd: head_dim must be 64 and T <= 128 (got %d, %d)", head_dim, T);
  FFVC_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "ffvc_attn_text_fwd: misaligned pointers");
  const int Tp = ((T + 3) & ~3) + 1;
  const int lds = (2 * T * AT_DH + T * Tp) * (int)sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_text_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * AT_TMAX * AT_DH + AT_TMAX * (AT_TMAX + 1)) * 4);
    attr = true;
  }
  hipLaunchKernelGGL(attn_text_fwd_kernel, dim3(B * heads), dim3(AT_TMAX), lds, (hipStream_t)stream, qkv, out, T, heads, scale, causal, Tp);
  FFVC_LAUNCH_CHECK();
  return 0;
}
