"""Developer micro-benchmark: LayerNorm forward of the ViT-L/14 tower rows (16448 x 1024 fp32 -> f16) plain, + fp8 bytes, fp8 bytes only, and the
separate quantisation pass it replaces.  usage: python tools/ln_f8_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
for rows, dim in ((16448, 1024), (25600, 768)):
    x = torch.randn(rows, dim, device=dev)
    g, b = torch.ones(dim, device=dev), torch.zeros(dim, device=dev)
    y, _, _ = K.layernorm_fwd(x, g, b, torch.float16)
    sc = K.Fp8Scale(K.E4M3, dev)
    K.fp8_quant(y, sc)
    t0 = timeit(lambda: K.layernorm_fwd(x, g, b, torch.float16), iters=30)
    tq = timeit(lambda: K.fp8_quant(y, sc), iters=30)
    t1 = timeit(lambda: K.layernorm_fwd(x, g, b, torch.float16, f8=sc), iters=30)
    t2 = timeit(lambda: K.layernorm_fwd(x, g, b, torch.float16, f8=sc, f8_only=True), iters=30)
    print(f"ln_fwd {rows}x{dim}: plain {t0 * 1e6:6.1f} us | quant pass {tq * 1e6:6.1f} us | + fp8 {t1 * 1e6:6.1f} us | fp8 only {t2 * 1e6:6.1f} us")
