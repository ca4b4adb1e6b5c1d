"""Is LayerNorm backward's parameter-gradient tail (one fp32 atomic per column and workgroup: 512 same-address atomics per column at the
Mixer's shape) exposed?  The same launch with and without parameter gradients, at 16384 x 1024 (fp32 x / dres, f16 dy), us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

for rows, dim in ((16384, 1024), (25600, 768), (512, 1024)):
    dy = torch.randn(rows, dim, device="cuda").half()
    x = torch.randn(rows, dim, device="cuda")
    dres = torch.randn(rows, dim, device="cuda")
    g = torch.randn(dim, device="cuda")
    _, mean, rstd = K.layernorm_fwd(x, g, g, torch.float16)
    dg, db = torch.zeros(dim, device="cuda"), torch.zeros(dim, device="cuda")
    t_frozen = timeit(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_lo=True), iters=30)
    t_acc = timeit(lambda: K.layernorm_bwd_acc(dy, x, g, mean, rstd, dg, db, dres=dres, want_lo=True), iters=30)
    nb = rows * dim * (2 + 4 + 4 + 4 + 2)
    print(f"{rows} x {dim}: without parameter gradients {t_frozen * 1e6:7.1f} us ({nb / t_frozen / 1e12:.2f} TB/s)   with (atomics) {t_acc * 1e6:7.1f} us "
          f"({nb / t_acc / 1e12:.2f} TB/s)")
