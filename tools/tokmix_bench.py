"""Time the fused token-mixing kernels against the two-GEMM path at the cfg2 shape (B=64, T=256, D=1024, O=1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

B, T, D, O = 64, 256, 1024, 1024


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dt in (torch.bfloat16, torch.float16):
    g = torch.Generator().manual_seed(0)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()  # noqa: E731
    xn, dy = mk(B, T, D).to(dt), mk(B, T, D).to(dt)
    w1, w2 = mk(O, T, sc=T ** -0.5).to(dt), mk(T, O, sc=O ** -0.5).to(dt)
    w2t = w2.t().contiguous()
    b1, b2, res = mk(O, sc=0.1), mk(T, sc=0.1), mk(B, T, D)
    h_pre, h = torch.empty(B, O, D, dtype=dt, device="cuda"), torch.empty(B, O, D, dtype=dt, device="cuda")
    y = torch.empty(B, T, D, dtype=torch.float32, device="cuda")
    dh = torch.empty_like(h)

    def unfused_fwd():
        K.gemm(w1, xn, h, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, bias=b1, act=K.ACT_GELU, aux=h_pre, ldaux=D,
               flags=K.F_WRITE_PREACT | K.F_BIAS_ALONG_M, batch=B, wb=(T * D, 0), yb=(O * D, 0), ab=(O * D, 0))
        K.gemm(w2, h, y, T, D, O, ldx=O, ldw=D, w_mode=K.OP_TRANS, bias=b2, residual=res, flags=K.F_BIAS_ALONG_M,
               batch=B, wb=(O * D, 0), yb=(T * D, 0), rb=(T * D, 0))

    def unfused_bwd_hidden():
        K.gemm(w2t, dy, dh, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, aux=h_pre, ldaux=D, act=K.ACT_GELU,
               flags=K.F_MUL_ACT_GRAD, batch=B, wb=(T * D, 0), yb=(O * D, 0), ab=(O * D, 0))

    flop = 2.0 * B * O * D * T
    tf = timeit(lambda: K.tokmix_fwd(xn, w1, b1, w2, b2, res))
    tu = timeit(unfused_fwd)
    tb = timeit(lambda: K.tokmix_bwd_hidden(xn, dy, w1, b1, w2t))
    tub = timeit(unfused_bwd_hidden)
    print(f"[{dt}] fwd fused {tf:7.1f} us ({2 * flop / tf / 1e6:6.1f} TFLOP/s) | two GEMMs {tu:7.1f} us ({2 * flop / tu / 1e6:6.1f})")
    print(f"[{dt}] bwd hidden fused (recompute + W2^T dy, writes h, dh) {tb:7.1f} us ({2 * flop / tb / 1e6:6.1f} TFLOP/s) | "
          f"dh GEMM alone (reads h_pre) {tub:7.1f} us")
