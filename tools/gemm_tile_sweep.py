"""Developer micro-benchmark: forced tile configurations (128x128 / 256x128 / 256x256) against the heuristic on the NT shapes of
the cfg2 step.  usage (GPU box): python tools/gemm_tile_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
shapes = [(25600, 768, 3072), (25600, 3072, 768), (25600, 768, 768), (25600, 768, 2304), (25600, 2304, 768), (4928, 512, 6144),
          (4928, 1536, 1536), (4928, 2048, 1536), (4928, 512, 1536), (16384, 1536, 512), (16384, 4096, 1024), (16384, 1024, 4096),
          (16384, 1024, 256), (16384, 256, 1024), (3200, 768, 3072), (3200, 3072, 768), (12800, 768, 3072), (8192, 512, 512),
          (65536, 128, 128), (16384, 512, 512), (4096, 512, 512), (16448, 1024, 4096), (16448, 4096, 1024), (16448, 1024, 1024)]
for (M, N, Kd) in shapes:
    x, w = torch.randn(M, Kd, device=dev).to(dt), (torch.randn(N, Kd, device=dev) * 0.05).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    row = []
    for tile in (1, 128, 256, 512):
        K.set_option("gemm2_tile", tile)
        t = timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd))
        row.append(t * 1e6)
    best = min(row[1:])
    print(f"NT {M}x{N}x{Kd}: auto {row[0]:7.1f} | 128 {row[1]:7.1f} | 256 {row[2]:7.1f} | 512 {row[3]:7.1f} us   "
          f"tiles512 {((M + 255) // 256) * ((N + 255) // 256):5d}  auto/best {row[0] / best:.2f}")

# 3x3 convolutions of the decoder at batch 64 (implicit GEMM): forced generic tiles vs the heuristic (which may take the
# haloed row-tile kernel where the geometry allows)
Bc = 64
for (H, Cin, Cout, ups) in [(16, 512, 512, False), (32, 512, 512, False), (32, 512, 512, True), (64, 512, 256, False), (64, 256, 256, False),
                            (64, 256, 256, True), (128, 256, 256, False), (128, 256, 128, False), (128, 128, 128, False),
                            (256, 128, 128, False)]:
    Hin = H // 2 if ups else H
    x = torch.randn(Bc, Hin, Hin, Cin, device=dev).to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).to(dt)
    y = torch.empty(Bc, H, H, Cout, device=dev, dtype=dt)
    row = []
    for tile in (1, 128, 256, 512):
        K.set_option("gemm2_tile", tile)
        K.set_option("conv_row", 1 if tile == 1 else 0)
        t = timeit(lambda: K.gemm(x, w, y, Bc * H * H, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, conv=(H, H, Cin),
                                  flags=K.F_UPSAMPLE2X if ups else 0), iters=10)
        row.append(t * 1e6)
    f = 2.0 * Bc * H * H * Cout * 9 * Cin / 1e6
    print(f"conv b{Bc} {H}^2 {Cin}->{Cout}{' ups' if ups else ''}: auto {row[0]:7.1f} ({f / row[0]:5.0f} TF) | 128 {row[1]:7.1f} | 256 {row[2]:7.1f} | "
          f"512 {row[3]:7.1f} us   tiles512 {((Bc * H * H + 255) // 256) * ((Cout + 255) // 256):6d}")
K.set_option("gemm2_tile", 1)
K.set_option("conv_row", 1)
