"""Developer micro-benchmark: fp8 GEMM (e4m3 x e4m3, 16-bit output) per tile configuration on the ViT-L/14 linears of cfg5.
FFVC_FP8_BM is read once per process: run as  FFVC_FP8_BM={0,128,256,512} python tools/fp8_tile_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
tag = os.environ.get("FFVC_FP8_BM", "0")
for (M, N, Kd) in [(16448, 1024, 4096), (16448, 4096, 1024), (16448, 1024, 1024), (16448, 3072, 1024), (16448, 1024, 3072),
                   (6400, 768, 3072), (25600, 768, 3072)]:
    x = torch.randn(M, Kd, device=dev).half()
    w = (torch.randn(N, Kd, device=dev) * 0.05).half()
    sx, sw = K.Fp8Scale(K.E4M3, dev), K.Fp8Scale(K.E4M3, dev)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    y = torch.empty(M, N, device=dev, dtype=torch.float16)
    y32 = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    t0 = timeit(lambda: K.gemm_fp8(x8, w8, y, M, N, Kd, sx, sw, lo_dtype=torch.float16))
    t1 = timeit(lambda: K.gemm_fp8(x8, w8, y32, M, N, Kd, sx, sw, lo_dtype=torch.float16, residual=res))
    f = 2.0 * M * N * Kd / 1e12
    print(f"fp8 NT {M}x{N}x{Kd} tile {tag:>3}: plain16 {t0 * 1e6:6.1f} us {f / t0:5.0f} TF | f32 + residual {t1 * 1e6:6.1f} us {f / t1:5.0f} TF")
