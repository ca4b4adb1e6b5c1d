"""GroupNorm backward: the one-pass register-resident kernel (FFVC_GN_BWD_FUSED=1) vs the two-pass form, per decoder level.
usage: FFVC_GN_BWD_FUSED=0|1 python tools/gn_bwd_bench.py   (prints time, achieved GB/s on 3(+1) tensor passes, checksum)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dt = torch.float16
tag = os.environ.get("FFVC_GN_BWD_FUSED", "1")
for B, H, C, swish in ((64, 256, 128, True), (64, 128, 128, True), (64, 128, 256, True), (64, 64, 256, True), (64, 32, 512, True),
                       (64, 16, 512, True), (64, 16, 512, False), (3, 48, 64, True)):
    g_ = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, H, H, C, device="cuda", generator=g_).to(dt)
    g, b = 1 + 0.1 * torch.randn(C, device="cuda", generator=g_), 0.1 * torch.randn(C, device="cuda", generator=g_)
    y, mean, rstd = K.groupnorm_fwd(x, g, b, swish=swish)
    dy = torch.randn(B, H, H, C, device="cuda", generator=g_).to(dt)
    dres = torch.randn(B, H, H, C, device="cuda", generator=g_).to(dt)
    dx = K.groupnorm_bwd(dy, x, g, b, mean, rstd, dres=dres, swish=swish)
    # fp64 reference on the device
    xd, dyd = x.double(), dy.double()
    xg = xd.view(B, H * H, 32, C // 32)
    mu = xg.mean(dim=(1, 3), keepdim=True)
    var = xg.var(dim=(1, 3), unbiased=False, keepdim=True)
    xh = ((xg - mu) / torch.sqrt(var + 1e-6)).view(B, H, H, C)
    u = xh * g.double() + b.double()
    sg = torch.sigmoid(u)
    d = dyd * (sg * (1 + u * (1 - sg)) if swish else 1.0) * g.double()
    dg_, xg_ = d.view(B, H * H, 32, C // 32), xh.view(B, H * H, 32, C // 32)
    m1 = dg_.mean(dim=(1, 3), keepdim=True)
    m2 = (dg_ * xg_).mean(dim=(1, 3), keepdim=True)
    ref = ((dg_ - m1 - xg_ * m2) / torch.sqrt(var + 1e-6)).view(B, H, H, C) + dres.double()
    err = ((dx.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    t = timeit(lambda: K.groupnorm_bwd(dy, x, g, b, mean, rstd, dres=dres, swish=swish), iters=10)
    n = x.numel() * 2
    print(f"fused={tag} gn_bwd b{B} {H}^2 x{C} swish={int(swish)}: {t * 1e6:8.1f} us  {4 * n / t / 1e9:7.0f} GB/s on 4 passes  rel-rms err {err:.2e}", flush=True)
    del x, y, dy, dres, dx, xd, dyd, xg, xh, u, sg, d, ref
