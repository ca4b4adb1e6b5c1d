"""Developer micro-benchmark: the ViT residual-stream GEMMs (N = 768 wide, fp32 output + fp32 residual) per tile configuration,
against the same shape with a plain 16-bit output.  usage (GPU box): python tools/gemm_resid.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
for (M, N, Kd) in [(25600, 768, 3072), (25600, 768, 2304), (25600, 768, 768), (16384, 1024, 4096)]:
    x, w = torch.randn(M, Kd, device=dev).to(dt), (torch.randn(N, Kd, device=dev) * 0.05).to(dt)
    y16 = torch.empty(M, N, device=dev, dtype=dt)
    y32 = torch.empty(M, N, device=dev, dtype=torch.float32)
    res = torch.randn(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    for tile in (0, 128, 256, 512):
        K.set_option("gemm2_tile", tile if tile else 1)
        t0 = timeit(lambda: K.gemm(x, w, y16, M, N, Kd, ldx=Kd, ldw=Kd))
        t1 = timeit(lambda: K.gemm(x, w, y32, M, N, Kd, ldx=Kd, ldw=Kd, bias=bias, residual=res))
        t2 = timeit(lambda: K.gemm(x, w, y32, M, N, Kd, ldx=Kd, ldw=Kd, residual=res))
        f = 2.0 * M * N * Kd / 1e12
        print(f"NT {M}x{N}x{Kd} tile {tile or 'auto':>4}: plain16 {t0 * 1e6:6.1f} us {f / t0:6.0f} TF | f32 out + bias + f32 residual "
              f"{t1 * 1e6:6.1f} us {f / t1:6.0f} TF | f32 out + f32 residual {t2 * 1e6:6.1f} us {f / t2:6.0f} TF")
