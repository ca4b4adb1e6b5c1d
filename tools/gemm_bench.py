"""Developer micro-benchmark of ffvc_gemm at the hot-path shapes.

Usage (GPU box): python tools/gemm_bench.py [--dtype bf16|f32]
Prints achieved TFLOP/s per shape (algorithmic FLOPs / HIP-event time).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=16, help="images for the conv shapes")
    ap.add_argument("--uniform", action="store_true", help="uniform [-1, 1) operands instead of N(0, 1)")
    ap.add_argument("--only", default="", help="substring filter on the row names (NT / TN / conv / tokmix)")
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}.get(a.dtype, torch.float32)
    dev = torch.device("cuda:0")
    print("device:", K.device_info())

    def r(*s):
        if a.uniform:
            return (torch.rand(*s, device=dev, dtype=torch.float32) * 2 - 1).to(dt)
        return torch.randn(*s, device=dev, dtype=torch.float32).to(dt)

    rows = []
    want = lambda tag: (not a.only) or tag in a.only.split(',')
    # NT linears (mixer channel-mix, ViT)
    for (M, N, Kd) in [] if not want('NT') else [(16384, 4096, 1024), (16384, 1024, 4096), (25600, 2304, 768), (25600, 3072, 768),
                       (25600, 768, 3072), (16384, 1024, 256), (4096, 4096, 4096), (8192, 8192, 8192)]:
        x, w = r(M, Kd), r(N, Kd)
        y = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd))
        rows.append((f"NT {M}x{N}x{Kd}", 2.0 * M * N * Kd / t / 1e12, t * 1e3))
    # TN wgrad with split-K
    for (M, N, Kd, sk) in [] if not want('TN') else [(4096, 1024, 16384, 1), (4096, 1024, 16384, 3), (1024, 4096, 16384, 3), (4096, 1024, 16384, 2),
                           (2304, 768, 25600, 4)]:
        xt, wt = r(Kd, M), r(Kd, N)
        y = torch.zeros(M, N, device=dev, dtype=torch.float32)
        t = timeit(lambda: K.gemm_splitk_accumulate(xt, wt, y, M, N, Kd, sk, ldx=M, ldw=N, x_mode=K.OP_TRANS,
                                                    w_mode=K.OP_TRANS))
        rows.append((f"TN {M}x{N}x{Kd} slab-sk{sk}", 2.0 * M * N * Kd / t / 1e12, t * 1e3))
    # batched NN token mix: out[b][o,d] = W[o,t] xn[b][t,d]
    B, T, D, O = (64, 256, 1024, 1024) if want('tokmix') else (1, 256, 256, 256)
    Wm, xn = r(O, T), r(B, T, D)
    out = torch.empty(B, O, D, device=dev, dtype=dt)
    t = timeit(lambda: K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, batch=B,
                              wb=(T * D, 0), yb=(O * D, 0)))
    rows.append((f"NN tokmix b{B} {O}x{D}x{T}", 2.0 * B * O * D * T / t / 1e12, t * 1e3))
    # 3x3 convs (decoder), NHWC
    Bc = a.batch
    for (H, Cin, Cout, ups) in [] if not want('conv') else [(256, 128, 128, False), (128, 256, 256, False), (64, 256, 256, False),
                                (32, 512, 512, False), (256, 128, 128, True), (16, 512, 512, False)]:
        Hin = H // 2 if ups else H
        x = r(Bc, Hin, Hin, Cin)
        w = r(Cout, 3, 3, Cin)
        y = torch.empty(Bc, H, H, Cout, device=dev, dtype=dt)
        t = timeit(lambda: K.gemm(x, w, y, Bc * H * H, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3,
                                  conv=(H, H, Cin), flags=K.F_UPSAMPLE2X if ups else 0), iters=10)
        rows.append((f"conv3x3 b{Bc} {H}^2 {Cin}->{Cout}{' ups' if ups else ''}",
                     2.0 * Bc * H * H * Cout * 9 * Cin / t / 1e12, t * 1e3))
    for name, tf, ms in rows:
        print(f"{name:44s} {tf:8.1f} TFLOP/s  {ms:8.3f} ms")


if __name__ == "__main__":
    main()
