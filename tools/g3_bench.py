"""Round 6: isolated timing of the epilogue-heavy NT launches of cfg2 — 256x256 ring kernel vs the two-workgroups-per-CU 256x128
kernel (csrc/gemm3.hip) — per epilogue kind, with the stagger swept.  Usage (GPU box): python tools/g3_bench.py [--stagger a,b,c]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20, warmup=4):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--modes", default="-1,1", help="gemm3 option values to time: -1 never, 0 heuristic, 1 forced")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    dev = torch.device("cuda:0")
    print("device:", K.device_info(), "FFVC_G3_STAGGER =", os.environ.get("FFVC_G3_STAGGER"), "FFVC_LIB =", os.environ.get("FFVC_LIB"))
    shapes = [(16384, 4096, 1024), (16384, 1024, 4096), (25600, 768, 3072), (25600, 3072, 768), (25600, 768, 768), (25600, 2304, 768),
              (4096, 4096, 4096), (8192, 8192, 1024)]
    rows = []
    for (M, N, Kd) in shapes:
        x = torch.randn(M, Kd, device=dev).to(dt)
        w = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(dt)
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, dtype=dt, device=dev)
        y32 = torch.empty(M, N, dtype=torch.float32, device=dev)
        aux = torch.empty(M, N, dtype=dt, device=dev)
        cs = torch.zeros(N, dtype=torch.float32, device=dev)
        kinds = {
            "plain": lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd),
            "f32res": lambda: K.gemm(x, w, y32, M, N, Kd, ldx=Kd, ldw=Kd, residual=res, bias=b),
            "gelu+act'": lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=K.ACT_GELU, aux=aux, ldaux=N,
                                        flags=K.F_WRITE_PREACT | K.F_AUX_ACTGRAD),
            "mulaux+cs": lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, act=K.ACT_GELU, aux=aux, ldaux=N,
                                        flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD, colsum=cs),
            "qgelu+act'": lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=K.ACT_QUICKGELU, aux=aux, ldaux=N,
                                         flags=K.F_WRITE_PREACT | K.F_AUX_ACTGRAD),
        }
        for kind, fn in kinds.items():
            cells = []
            for mode in [int(v) for v in a.modes.split(",")]:
                try:
                    K.set_option("gemm3", mode)
                except Exception:            # an A/B library built before the kernel existed
                    if mode != -1:
                        continue
                us = timeit(fn)
                cells.append(f"g3={mode:2d}: {us:7.1f} us {2.0 * M * N * Kd / us / 1e6:7.1f} TF")
            try:
                K.set_option("gemm3", -100)
            except Exception:
                pass
            print(f"{M:6d}x{N:5d}x{Kd:5d} {kind:11s} | " + " | ".join(cells), flush=True)
        del x, w, res, y, y32, aux


if __name__ == "__main__":
    main()
