"""Developer A/B: epilogue desynchronisation of the LDS-DMA GEMM (FFVC_STAGGER_TICKS, read once per process) on the
short-K / heavy-epilogue shapes of the step.  usage (GPU box): FFVC_STAGGER_TICKS=2000 python tools/stagger_ab.py [tile]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K.set_option("gemm2_tile", tile)
res = []
for (M, N, Kd, name) in ((16384, 4096, 1024, "chan-fc1"), (16384, 1024, 4096, "chan-fc2"), (25600, 3072, 768, "clip-fc1"),
                         (25600, 768, 3072, "clip-fc2"), (25600, 2304, 768, "clip-qkv")):
    x, w = torch.randn(M, Kd, device=dev).to(dt), torch.randn(N, Kd, device=dev).to(dt)
    y, pre = torch.empty(M, N, device=dev, dtype=dt), torch.empty(M, N, device=dev, dtype=dt)
    bias = torch.randn(N, device=dev)
    for vn, kw in (("plain", {}), ("gelu+preact", dict(bias=bias, act=K.ACT_GELU, flags=K.F_WRITE_PREACT, aux=pre, ldaux=N)),
                   ("act_grad", dict(act=K.ACT_GELU, flags=K.F_MUL_ACT_GRAD, aux=pre, ldaux=N))):
        t = timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, **kw))
        res.append(f"{name}/{vn} {t * 1e6:6.1f}us {2.0 * M * N * Kd / t / 1e12:5.0f}TF")
print(f"ticks={os.environ.get('FFVC_STAGGER_TICKS', '0'):>5s} tile={tile}: " + " | ".join(res))
