"""Run ONE ffvc_gemm shape repeatedly (for rocprofv3 --pmc passes).  usage: gemm_one.py conv|nt|nn [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "conv"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
dt = torch.bfloat16


def r(*s):
    return torch.randn(*s, device=dev).to(dt)


if kind == "conv":
    B, H, C = 16, 256, 128
    x, w, y = r(B, H, H, C), r(C, 3, 3, C), torch.empty(B, H, H, C, device=dev, dtype=dt)
    fn = lambda: K.gemm(x, w, y, B * H * H, C, 9 * C, ldw=9 * C, x_mode=K.OP_CONV3X3, conv=(H, H, C))  # noqa: E731
elif kind == "nt":
    M, N, Kd = [int(v) for v in os.environ.get("MNK", "16384,1024,4096").split(",")]
    x, w, y = r(M, Kd), r(N, Kd), torch.empty(M, N, device=dev, dtype=dt)
    fn = lambda: K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd)  # noqa: E731
else:
    B, T, D, O = 64, 256, 1024, 1024
    Wm, xn, out = r(O, T), r(B, T, D), torch.empty(B, O, D, device=dev, dtype=dt)
    fn = lambda: K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, batch=B, wb=(T * D, 0), yb=(O * D, 0))  # noqa: E731
for _ in range(iters):
    fn()
torch.cuda.synchronize()
