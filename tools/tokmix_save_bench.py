"""Token-mixing MLP per layer at cfg2's shapes: fused forward (recompute in the backward) vs the forward that saves h / act'.
usage: python tools/tokmix_save_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dt = torch.float16
B, T, D, O = 64, 256, 1024, 1024
g = torch.Generator(device="cuda").manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, device="cuda", generator=g) * sc).to(dt)  # noqa: E731
xn, dy = r(B, T, D), r(B, T, D)
w1, w2 = r(O, T, sc=T ** -0.5), r(T, O, sc=O ** -0.5)
w2t = w2.t().contiguous()
b1, b2 = torch.randn(O, device="cuda"), torch.randn(T, device="cuda")
res = torch.randn(B, T, D, device="cuda")
t_f = timeit(lambda: K.tokmix_fwd(xn, w1, b1, w2, b2, res))
t_s = timeit(lambda: K.tokmix_fwd_save(xn, w1, b1, w2, b2, res))
t_b = timeit(lambda: K.tokmix_bwd_hidden(xn, dy, w1, b1, w2t))
y, h, ga = K.tokmix_fwd_save(xn, w1, b1, w2, b2, res)
dh = torch.empty_like(h)
t_g = timeit(lambda: K.gemm(w2t, dy, dh, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, aux=ga, ldaux=D, act=K.ACT_GELU,
                            flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD, batch=B, wb=(T * D, 0), yb=(O * D, 0), ab=(O * D, 0)))
print(f"tokmix fwd (recompute form) {t_f * 1e6:7.1f} us | fwd + save h, act' {t_s * 1e6:7.1f} us | bwd hidden (recompute) {t_b * 1e6:7.1f} us | "
      f"bwd dh GEMM x act' {t_g * 1e6:7.1f} us  ->  per layer {1e6 * (t_f + t_b):7.1f} vs {1e6 * (t_s + t_g):7.1f} us")
