"""The small-M linears of the VitGAN / x-transformer mappers in isolation: M = rows of one per-GPU batch (512 = 32
samples x 16 tokens), 2-8 MB of weights.  Per shape and direction: us per launch, TFLOP/s, and the weight-streaming floor
(weight bytes / 5 TB/s).  Compare library variants through the environment (FFVC_SMALLM=0|1, FFVC_SK_FIXUP, FFVC_GEMM2_BM)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

dt = torch.float16
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def r(*s):
    return (torch.randn(*s, device=dev) * 0.3).to(dt)


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


print(f"# M = {M} rows, f16; env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("FFVC_")))
print("# kind                      M     N     K |      us   TFLOP/s | floor us (weights / 5 TB/s)")
rows = []
for name, N, Kd in (("qkv", 3064, 1024), ("attn out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096), ("xf qkv 256", 768, 256),
                    ("head 1024->256", 256, 1024)):
    x, w = r(M, Kd), r(N, Kd)
    b = torch.randn(N, device=dev)
    y16 = torch.empty(M, N, device=dev, dtype=dt)
    y32 = torch.empty(M, N, device=dev)
    res = torch.randn(M, N, device=dev)
    aux = torch.empty(M, N, device=dev, dtype=dt)
    variants = [("fwd plain 16-bit out", lambda: K.gemm(x, w, y16, M, N, Kd, ldx=Kd, ldw=Kd, bias=b)),
                ("fwd +fp32 residual", lambda: K.gemm(x, w, y32, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, residual=res))]
    if name == "fc1":
        variants.append(("fwd GELU + act' store", lambda: K.gemm(x, w, y16, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=K.ACT_GELU, aux=aux, ldaux=N,
                                                                 flags=K.F_WRITE_PREACT | K.F_AUX_ACTGRAD)))
    if name == "fc2":      # its dgrad: dh[M, 4096] = dy[M, 1024] W2 with the aux multiply
        dy, wt = r(M, N), r(Kd, N)
        dh, ag = torch.empty(M, Kd, device=dev, dtype=dt), r(M, Kd)
        variants.append(("dgrad x act' (aux mul)", lambda: K.gemm(dy, wt, dh, M, Kd, N, ldx=N, ldw=N, aux=ag, ldaux=Kd, act=K.ACT_GELU,
                                                                  flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD)))
    for vn, fn in variants:
        us = timeit(fn)
        rows.append((f"{name}: {vn}", M, N, Kd, us))
    # weight gradient dW[N, Kd] += dy[M, N]^T x[M, Kd]
    dy2, wg = r(M, N), torch.zeros(N, Kd, device=dev)
    from feed_forward_vqgan_clip_amd import ops
    sk = ops._split_k(N, Kd, M, 64, big_tiles=False)
    us = timeit(lambda: K.gemm_splitk_accumulate(dy2, x, wg, N, Kd, M, sk, ldx=N, ldw=Kd, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS))
    rows.append((f"{name}: wgrad (sk {sk})", N, Kd, M, us))
for nm, m, n, k, us in rows:
    fl = 2.0 * m * n * k
    wbytes = (n * k * 2) if "wgrad" not in nm else (m * n * 4 * 2)
    print(f"{nm:32s} {m:5d} {n:5d} {k:5d} | {us:7.1f} {fl / us / 1e6:9.1f} | {wbytes / 5e12 * 1e6:6.1f}")
