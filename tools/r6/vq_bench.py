"""Round 6: vector_quantize's distance GEMM + argmin (main.py:133-139) at the step's sizes — two launches (fp32 distance matrix written and
read back) against the argmin inside the GEMM's epilogue (FFVC_F_VQ_ARGMIN).  usage (GPU box): python tools/r6/vq_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
for rows, codes, C in [(16384, 16384, 256), (8192, 16384, 256), (2048, 16384, 256), (16384, 1024, 256)]:
    g = torch.Generator().manual_seed(rows)
    x, cb = torch.randn(rows, C, generator=g).to(dev), torch.randn(codes, C, generator=g).to(dev)
    xn, cn = K.rownorm_sq(x), K.rownorm_sq(cb)
    x3, cb3 = K.split3(x, torch.float16), K.split3(cb, torch.float16, weight_order=True)
    dot = torch.empty(rows, codes, dtype=torch.float32, device=dev)

    def two():
        K.gemm(x3, cb3, dot, rows, codes, 3 * C, ldx=3 * C, ldw=3 * C)
        return K.vq_argmin(dot, xn, cn)

    a, b = two(), K.vq_argmin_fused(x3, cb3, xn, cn)
    t2 = timeit(two, iters=10) * 1e6
    t1 = timeit(lambda: K.vq_argmin_fused(x3, cb3, xn, cn), iters=10) * 1e6
    print(f"rows {rows} codes {codes} depth {3 * C}: two launches {t2:7.1f} us | fused {t1:7.1f} us (incl. fill + mask) | equal {bool(torch.equal(a, b))}", flush=True)
