"""Developer micro-benchmark: ffvc_colsum (bias gradients) on the shapes of the step.  usage: FFVC_COLSUM_WGS=n python tools/colsum_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

tag = os.environ.get("FFVC_COLSUM_WGS", "default")
for rows, cols, dt in ((16384, 256, torch.float16), (16384, 1024, torch.float16), (16384, 4096, torch.float16), (16384, 1024, torch.float32),
                       (512, 4096, torch.float16), (8192, 1024, torch.float16), (512, 2048, torch.float32)):
    x = torch.randn(rows, cols, device="cuda").to(dt)
    out = torch.zeros(cols, device="cuda")
    K.colsum(x, out)
    err = (out.double() - x.double().sum(0)).abs().max().item() / (x.double().sum(0).abs().max().item() + 1e-30)
    t = timeit(lambda: K.colsum(x, out), iters=30)
    print(f"wgs={tag} colsum {rows}x{cols} {str(dt)[6:]}: {t * 1e6:6.1f} us  {x.numel() * x.element_size() / t / 1e9:6.0f} GB/s  err {err:.1e}", flush=True)
