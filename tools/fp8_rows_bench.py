"""Developer micro-benchmark: the fp8 GEMMs of the ViT-L/14 tower at 64 x 257 = 16448 rows — one launch vs 16384 + 64 rows (kernels.gemm_fp8's
row split, FFVC_FP8_ROWSPLIT) vs the 16384-row part alone, with and without the bias / residual epilogue.  usage: python tools/fp8_rows_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
for (N, Kd, name) in ((3072, 1024, "in_proj"), (1024, 1024, "out_proj"), (1024, 3072, "in_proj^T"), (1024, 4096, "c_proj / c_fc^T"), (4096, 1024, "c_fc")):
    M = 16448
    x, w = torch.randn(M, Kd, device=dev).to(dt), torch.randn(N, Kd, device=dev).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    sx, sw = K.Fp8Scale(K.E4M3, dev), K.Fp8Scale(K.E4M3, dev)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    bias = torch.randn(N, device=dev)
    res = {}
    for split in (False, True):
        K._FP8_ROWSPLIT = split
        res[split] = timeit(lambda: K.gemm_fp8(x8, w8, y, M, N, Kd, sx, sw, lo_dtype=dt, bias=bias))
    K._FP8_ROWSPLIT = False
    t_main = timeit(lambda: K.gemm_fp8(x8[:16384], w8, y[:16384], 16384, N, Kd, sx, sw, lo_dtype=dt, bias=bias))
    t_tail = timeit(lambda: K.gemm_fp8(x8[16384:], w8, y[16384:], 64, N, Kd, sx, sw, lo_dtype=dt, bias=bias))
    t_sk = timeit(lambda: K.gemm_fp8_skinny(x8[16384:], w8, y[16384:], 64, N, Kd, sx, sw, bias=bias))
    fl = 2.0 * M * N * Kd
    print(f"{name:16s} {M}x{N}x{Kd}: one launch {res[False] * 1e6:6.1f} us ({fl / res[False] / 1e12:5.0f} TF) | 16384 + 64 {res[True] * 1e6:6.1f} us | "
          f"16384 alone {t_main * 1e6:6.1f} us, 64 alone {t_tail * 1e6:5.1f} us (tiled) / {t_sk * 1e6:5.1f} us (skinny)")
