"""Developer micro-benchmark: cost of the fused epilogue variants on the two GELU GEMMs of the Mixer
(token fc1 = batched NN 1024x1024x256 b64, channel fc1 = NT 16384x4096x1024), per tile configuration.
usage (GPU box): python tools/gemm_epi.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16


def r(*s):
    return torch.randn(*s, device=dev).to(dt)


B, T, D, O = 64, 256, 1024, 1024
Wm, xn = r(O, T), r(B, T, D)
out, pre = torch.empty(B, O, D, device=dev, dtype=dt), torch.empty(B, O, D, device=dev, dtype=dt)
bias_m = torch.randn(O, device=dev)
M, N, Kd = 16384, 4096, 1024
x2, w2 = r(M, Kd), r(N, Kd)
y2, pre2 = torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, N, device=dev, dtype=dt)
bias_n = torch.randn(N, device=dev)

variants = {
    "plain": dict(),
    "bias": dict(bias=True),
    "gelu": dict(bias=True, act=K.ACT_GELU),
    "gelu+preact": dict(bias=True, act=K.ACT_GELU, flags=K.F_WRITE_PREACT, aux=True),
    "mul_act_grad": dict(act=K.ACT_GELU, flags=K.F_MUL_ACT_GRAD, aux=True),
    "quickgelu+preact": dict(bias=True, act=K.ACT_QUICKGELU, flags=K.F_WRITE_PREACT, aux=True),
}
for tile in (128, 256, 512):
    K.set_option("gemm2_tile", tile)
    for name, v in variants.items():
        fl = v.get("flags", 0)
        t1 = timeit(lambda: K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, batch=B, wb=(T * D, 0),
                                   yb=(O * D, 0), ab=(O * D, 0), bias=bias_m if v.get("bias") else None,
                                   act=v.get("act", K.ACT_NONE), aux=pre if v.get("aux") else None, ldaux=D,
                                   flags=fl | (K.F_BIAS_ALONG_M if v.get("bias") else 0)))
        t2 = timeit(lambda: K.gemm(x2, w2, y2, M, N, Kd, ldx=Kd, ldw=Kd, bias=bias_n if v.get("bias") else None,
                                   act=v.get("act", K.ACT_NONE), aux=pre2 if v.get("aux") else None, ldaux=N, flags=fl))
        print(f"tile {tile:3d} {name:18s} tok-fc1 {t1 * 1e6:7.1f} us {2.0 * B * O * D * T / t1 / 1e12:7.1f} TF | "
              f"chan-fc1 {t2 * 1e6:7.1f} us {2.0 * M * N * Kd / t2 / 1e12:7.1f} TF")
