"""A/B of the weight-gradient GEMM (both operands reduction-major, fp32 slab output) on the register-staged kernel vs the
LDS-DMA tiles, per split-K factor, at the step's wgrad shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else torch.bfloat16
dev = "cuda"
SHAPES = {"cfg2": [(4096, 1024, 16384), (1024, 4096, 16384), (3072, 768, 25600), (512, 65536, 64)],
          "cfg3": [(4096, 1024, 512), (1024, 4096, 512), (1024, 1024, 512), (3072, 1024, 512)],          # VitGAN at 32 samples: 512-row reductions
          "cfg4": [(1024, 256, 16384), (256, 1024, 16384), (768, 256, 16384), (256, 256, 16384)]}        # x-transformer 256 wide, 16384 rows
for (M, N, Kd) in SHAPES[sys.argv[2] if len(sys.argv) > 2 else "cfg2"]:
    xt = torch.randn(Kd, M, device=dev).to(dt)
    wt = torch.randn(Kd, N, device=dev).to(dt)
    y = torch.zeros(M, N, device=dev, dtype=torch.float32)
    ref = None
    for tile in (0, 128, 256, 512):
        for sk in (1, 2, 4, 8, 16, 32):
            if Kd // sk < 64:
                continue
            K.set_option("gemm2_tile", tile)
            try:
                y.zero_()
                K.gemm_splitk_accumulate(xt, wt, y, M, N, Kd, sk, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS)
                if ref is None:
                    ref = y.clone()
                err = ((y - ref).abs().max() / ref.abs().max()).item()
                t = timeit(lambda: K.gemm_splitk_accumulate(xt, wt, y, M, N, Kd, sk, ldx=M, ldw=N, x_mode=K.OP_TRANS,
                                                            w_mode=K.OP_TRANS), iters=10)
            finally:
                K.set_option("gemm2_tile", 1)
            print(f"TN {M}x{N}x{Kd} tile={tile:3d} sk={sk}: {2.0 * M * N * Kd / t / 1e12:7.1f} TFLOP/s {t * 1e6:8.1f} us  relerr {err:.1e}")
