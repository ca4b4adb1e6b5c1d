"""Developer micro-benchmark: the 16-bit GEMMs of the ViT-L/14 tower at 64 x 257 = 16448 rows vs 16384 rows (what the 64-row remainder costs the
f16 path, whose kernel folds the last partly filled round into an in-kernel split-K "tail mode").  usage: python tools/f16_rows_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
for (N, Kd, name) in ((3072, 1024, "in_proj"), (1024, 1024, "out_proj"), (1024, 3072, "in_proj^T"), (1024, 4096, "c_proj / c_fc^T"), (4096, 1024, "c_fc")):
    res = {}
    for M in (16448, 16384, 64):
        x, w = torch.randn(M, Kd, device=dev).to(dt), torch.randn(N, Kd, device=dev).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt)
        bias = torch.randn(N, device=dev)
        res[M] = timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, bias=bias))
    fl = 2.0 * 16448 * N * Kd
    print(f"{name:16s} x{N}x{Kd}: 16448 rows {res[16448] * 1e6:6.1f} us ({fl / res[16448] / 1e12:5.0f} TF) | 16384 rows {res[16384] * 1e6:6.1f} us | 64 rows {res[64] * 1e6:5.1f} us")
