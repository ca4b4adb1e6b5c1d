"""Round 6: what the GroupNorm-backward statistics cost inside the dgrad convolution (conv3, FFVC_F_GNB_SUMS) against the separate
statistics pass they replace — decoder levels of cfg2 at batch 64, f16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dt, dev, G = torch.float16, torch.device("cuda:0"), 32
for (B, H, C, Cd) in [(64, 256, 128, 128), (64, 128, 128, 128), (64, 128, 256, 256), (64, 64, 256, 256)]:
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    gamma, beta = torch.randn(C, device=dev), torch.randn(C, device=dev)
    gout = (torch.randn(B, H, H, Cd, device=dev) * 0.1).to(dt)
    wd = (torch.randn(C, 9 * Cd, device=dev) * (9 * Cd) ** -0.5).to(dt)
    y, mean, rstd = K.groupnorm_fwd(x, gamma, beta, G, 1e-6, True)
    dy = torch.empty(B, H, H, C, dtype=dt, device=dev)
    sums = torch.zeros(B, G, 2, dtype=torch.float64, device=dev)
    dres = torch.randn(B, H, H, C, device=dev).to(dt)
    ok = K.conv_gnb_ok(gout, wd, dy, x, mean, rstd, gamma, beta, B, H, H, Cd, C)
    t_plain = timeit(lambda: K.gemm(gout, wd, dy, B * H * H, C, 9 * Cd, ldw=9 * Cd, x_mode=K.OP_CONV3X3, conv=(H, H, Cd)))
    t_gnb = timeit(lambda: K.gemm(gout, wd, dy, B * H * H, C, 9 * Cd, ldw=9 * Cd, x_mode=K.OP_CONV3X3, conv=(H, H, Cd),
                                  gnb=(x, mean, rstd, gamma, beta, sums, True, H * H, C // G))) if ok else float("nan")
    t_two = timeit(lambda: K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=dres, G=G, swish=True))
    t_one = timeit(lambda: K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=dres, G=G, swish=True, sums=sums))
    print(f"b{B} {H}^2 GN C={C} conv Cout={Cd}: dgrad plain {t_plain:7.1f} us | with statistics {t_gnb:7.1f} us (+{t_gnb - t_plain:6.1f}) | "
          f"groupnorm_bwd two passes {t_two:7.1f} us | apply only {t_one:7.1f} us (-{t_two - t_one:6.1f}) | net {t_gnb - t_plain - (t_two - t_one):+7.1f} us", flush=True)
    del x, gout, dy, dres, y
