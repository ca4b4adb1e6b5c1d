"""Developer micro-benchmark of the HBM-bound normalisation kernels at the cfg2 shapes: achieved GB/s against the
algorithmic bytes of each pass.  usage (GPU box): python tools/norm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
bf, f32 = torch.bfloat16, torch.float32


def report(name, t, nbytes):
    print(f"{name:52s} {t * 1e6:8.1f} us  {nbytes / t / 1e9:8.0f} GB/s  ({nbytes / 1e6:.0f} MB)")


for rows, dim, tag in ((16384, 1024, "mixer"), (25600, 768, "vit")):
    x = torch.randn(rows, dim, device=dev)
    g, b = torch.randn(dim, device=dev), torch.randn(dim, device=dev)
    y, mean, rstd = K.layernorm_fwd(x, g, b, bf)
    report(f"ln_fwd {tag} {rows}x{dim} f32->bf16", timeit(lambda: K.layernorm_fwd(x, g, b, bf)), rows * dim * 6)
    dy = torch.randn(rows, dim, device=dev).to(bf)
    dres = torch.randn(rows, dim, device=dev)
    report(f"ln_bwd {tag} +dres +param grads", timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_param_grads=True)),
           rows * dim * 14)
    dg, db = torch.zeros(dim, device=dev), torch.zeros(dim, device=dev)
    report(f"ln_bwd_acc {tag} +dres +param grads (bucket atomics) +lo copy",
           timeit(lambda: K.layernorm_bwd_acc(dy, x, g, mean, rstd, dg, db, dres=dres, want_lo=True)), rows * dim * 16)
    report(f"ln_bwd {tag} +dres", timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres)), rows * dim * 14)
    report(f"ln_bwd {tag} plain", timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd)), rows * dim * 10)

for B, H, C in ((64, 256, 128), (64, 128, 256), (64, 64, 256), (64, 32, 512), (64, 16, 512)):
    x = torch.randn(B, H, H, C, device=dev).to(bf)
    g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
    n = x.numel()
    y, mean, rstd = K.groupnorm_fwd(x, g, b, swish=True)
    report(f"gn_fwd b{B} {H}^2 x{C} (stats + apply)", timeit(lambda: K.groupnorm_fwd(x, g, b, swish=True), iters=10), n * 6)
    dy = torch.randn(B, H, H, C, device=dev).to(bf)
    dres = torch.randn(B, H, H, C, device=dev).to(bf)
    report(f"gn_bwd b{B} {H}^2 x{C} (stats + apply, +dres)",
           timeit(lambda: K.groupnorm_bwd(dy, x, g, b, mean, rstd, dres=dres, swish=True), iters=10), n * 12)
    for nb in (2, 4, 8, 16):       # the same backward as a host loop over groups of `nb` images (do the re-reads of pass 2 hit the Infinity Cache?)
        def loop():
            for i in range(0, B, nb):
                K.groupnorm_bwd(dy[i:i + nb], x[i:i + nb], g, b, mean[i:i + nb], rstd[i:i + nb], dres=dres[i:i + nb], swish=True)
        report(f"   ... in groups of {nb} images", timeit(loop, iters=5), n * 12)
    del x, y, dy, dres

for rows, cols in ((16384, 4096), (16384, 1024), (25600, 3072)):
    m = torch.randn(rows, cols, device=dev).to(bf)
    out = torch.zeros(cols, device=dev)
    report(f"colsum bf16 {rows}x{cols}", timeit(lambda: K.colsum(m, out, accumulate=True)), rows * cols * 2)
m = torch.randn(64, 1024, 1024, device=dev).to(bf)
out = torch.zeros(1024, device=dev)
report("rowsum bf16 b64 1024x1024", timeit(lambda: K.rowsum(m, out, 1024, accumulate=True)), m.numel() * 2)
