"""Round 6: forced tile configurations on the K-major x K-major launches of cfg3's CLIP ViT-B/32 tower (12800 rows = 32 samples x 8
cutouts x 50 tokens) and of its 512-row VitGAN generator.  usage (GPU box): python tools/r6/cfg3_tiles.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
shapes = [(12800, 3072, 768), (12800, 768, 3072), (12800, 2304, 768), (12800, 768, 768),
          (512, 4096, 1024), (512, 1024, 4096), (512, 3064, 1024), (512, 1024, 1024)]
for (M, N, Kd) in shapes:
    x, w = torch.randn(M, Kd, device=dev).to(dt), (torch.randn(N, Kd, device=dev) * 0.05).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    row = []
    for tile in (1, 64, 128, 256, 512):
        K.set_option("gemm2_tile", tile)
        try:
            row.append(timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd)) * 1e6)
        except Exception:
            row.append(float("nan"))
    K.set_option("gemm2_tile", 1)
    K.set_option("gemm3", 1)
    try:
        g3 = timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd)) * 1e6
    except Exception:
        g3 = float("nan")
    K.set_option("gemm3", -100)
    f = 2.0 * M * N * Kd / 1e6
    print(f"NT {M}x{N}x{Kd}: auto {row[0]:7.1f} ({f / row[0]:5.0f} TF) | 64x128 {row[1]:7.1f} | 128 {row[2]:7.1f} | 256x128 {row[3]:7.1f} | "
          f"256x256 {row[4]:7.1f} | gemm3 {g3:7.1f} us", flush=True)
# weight gradients of the generator: [N, K] += dy[512, N]^T x[512, K]
for (N, Kd, ldx, ldw) in [(4096, 1024, 4096, 1024), (1024, 4096, 1024, 4096), (3060, 1024, 3064, 1024), (1024, 1020, 1024, 1024)]:
    rows = 512
    dy, x = torch.randn(rows, ldx, device=dev).to(dt), torch.randn(rows, ldw, device=dev).to(dt)
    wg = torch.zeros(N, Kd, device=dev)
    t = timeit(lambda: K.gemm(dy, x, wg, N, Kd, rows, ldx=ldx, ldw=ldw, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS, flags=K.F_ACCUM_OUT)) * 1e6
    print(f"TT {N}x{Kd} from {rows} rows: {t:7.1f} us ({2.0 * N * Kd * rows / 1e6 / t:5.0f} TF)", flush=True)
for G in (4, 8):
    for (N, Kd) in [(4096, 1024), (1024, 4096)]:
        rows = 512
        dys = [torch.randn(rows, N, device=dev).to(dt) for _ in range(G)]
        xs = [torch.randn(rows, Kd, device=dev).to(dt) for _ in range(G)]
        wg = torch.zeros(G, N, Kd, device=dev)
        t = timeit(lambda: K.gemm_grouped_wgrad(dys, xs, wg, N * Kd, N, Kd, rows, N, Kd)) * 1e6
        print(f"TT grouped x{G} {N}x{Kd} from {rows} rows: {t:7.1f} us = {t / G:6.1f} per layer ({2.0 * G * N * Kd * rows / 1e6 / t:5.0f} TF)", flush=True)
