"""Developer micro-benchmark: flash attention at the ViT-L/14 tower's 257 tokens vs 256 (how much the +1 CLS row / key costs).  usage: python tools/attn257_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

B, H = 64, 16
for T in (256, 257, 320):
    qkv = (torch.randn(B, T, 3 * H * 64, device="cuda") * 0.5).half()
    o, lse = K.attn_flash_fwd(qkv, H, 0.125, False)
    do = torch.randn_like(o)
    tf = timeit(lambda: K.attn_flash_fwd(qkv, H, 0.125, False), iters=20)
    tb = timeit(lambda: K.attn_flash_bwd(qkv, o, do, lse, H, 0.125, False), iters=20)
    print(f"T={T}: fwd {tf * 1e6:6.1f} us  bwd {tb * 1e6:6.1f} us")
