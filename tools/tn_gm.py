"""TN (weight-gradient) GEMM at the channel-MLP shapes for one FFVC_TILE_GM value (read once per process): usage
FFVC_TILE_GM=<g> python tools/tn_gm.py [f16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
for (M, N, Kd) in [(4096, 1024, 16384), (1024, 4096, 16384)]:
    xt = torch.randn(Kd, M, device="cuda").to(dt)
    wt = torch.randn(Kd, N, device="cuda").to(dt)
    y = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    for sk in (2, 4):
        t = timeit(lambda: K.gemm_splitk_accumulate(xt, wt, y, M, N, Kd, sk, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS), iters=20)
        print(f"gm={os.environ.get('FFVC_TILE_GM', 'default')} TN {M}x{N}x{Kd} sk={sk}: {2.0 * M * N * Kd / t / 1e12:7.1f} TFLOP/s {t * 1e6:8.1f} us", flush=True)
