"""Developer micro-benchmark: flash-style attention kernels vs the batched GEMM + softmax path (fwd + bwd), on the
cfg4 (x-transformer, 1024 tokens causal) and cfg5 (ViT-L/14, 257 tokens) shapes.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import ops  # noqa: E402


def run(B, T, heads, causal, flash, dt=torch.float16, iters=10):
    os.environ["FFVC_ATTN_FLASH"] = "1" if flash else "0"
    qkv = (torch.randn(B, T, 3 * heads * 64, device="cuda") * 0.5).to(dt).requires_grad_(True)
    do = torch.randn(B, T, heads * 64, device="cuda").to(dt)
    for i in range(iters + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        o = ops.attention(qkv, heads, 0.125, causal)
        o.backward(do)
        qkv.grad = None
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def run_parts(B, T, heads, causal, dt=torch.float16, iters=20):
    from feed_forward_vqgan_clip_amd import kernels as K
    from tools.gemm_bench import timeit
    qkv = (torch.randn(B, T, 3 * heads * 64, device="cuda") * 0.5).to(dt)
    do = torch.randn(B, T, heads * 64, device="cuda").to(dt)
    o, lse = K.attn_flash_fwd(qkv, heads, 0.125, causal)
    tf = timeit(lambda: K.attn_flash_fwd(qkv, heads, 0.125, causal))
    tb = timeit(lambda: K.attn_flash_bwd(qkv, o, do, lse, heads, 0.125, causal))
    return tf * 1e3, tb * 1e3


for name, B, T, heads, causal in (("cfg4 x-transformer", 16, 1024, 6, True), ("cfg5 ViT-L/14", 512, 257, 16, False),
                                  ("ViT-L/14 b64", 64, 257, 16, False)):
    a, b = run(B, T, heads, causal, True), run(B, T, heads, causal, False)
    fl = 4 * B * heads * T * T * 64 * 3.5 * (0.5 if causal else 1.0)
    tf, tb = run_parts(B, T, heads, causal)
    f1 = 4 * B * heads * T * T * 64 * (0.5 if causal else 1.0)
    print(f"{name}: B={B} T={T} heads={heads} causal={causal}: flash {a:.3f} ms ({fl / a / 1e9:.1f} TFLOP/s) | gemm+softmax {b:.3f} ms"
          f" | kernels: fwd {tf:.3f} ms ({f1 / tf / 1e9:.0f} TF), bwd {tb:.3f} ms ({2.5 * f1 / tb / 1e9:.0f} TF)")
