"""Developer micro-benchmark: fp8 MFMA GEMM (ffvc_gemm_fp8, v_mfma_f32_32x32x64_f8f6f4) against the 16-bit LDS-DMA GEMM on
the ViT-L/14 linears of cfg5 (per-GPU batch 8 x cutn 8 x 257 tokens = 16448 rows) and two square references, incl. the
quantisation pass of the activation.  usage (GPU box): python tools/fp8_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
for (M, N, Kd, name) in ((16448, 3072, 1024, "L/14 qkv"), (16448, 1024, 1024, "L/14 out_proj"), (16448, 4096, 1024, "L/14 c_fc"),
                         (16448, 1024, 4096, "L/14 c_proj"), (131584, 4096, 1024, "L/14 c_fc b64"), (4096, 4096, 4096, "4096^3"),
                         (8192, 8192, 8192, "8192^3")):
    x, w = torch.randn(M, Kd, device=dev).to(dt), torch.randn(N, Kd, device=dev).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    t16 = timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd))
    sx, sw = K.Fp8Scale(K.E4M3, dev), K.Fp8Scale(K.E4M3, dev)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    t8 = timeit(lambda: K.gemm_fp8(x8, w8, y, M, N, Kd, sx, sw, lo_dtype=dt))
    tq = timeit(lambda: K.fp8_quant(x, sx))
    fl = 2.0 * M * N * Kd
    print(f"{name:16s} {M}x{N}x{Kd}: f16 {t16 * 1e6:7.1f} us {fl / t16 / 1e12:6.0f} TF | fp8 {t8 * 1e6:7.1f} us {fl / t8 / 1e12:6.0f} TF "
          f"({t16 / t8:.2f}x) | + quant {tq * 1e6:5.1f} us -> {fl / (t8 + tq) / 1e12:6.0f} TF ({t16 / (t8 + tq):.2f}x)")
