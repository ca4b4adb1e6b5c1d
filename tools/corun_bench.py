"""Developer micro-benchmark: does an HBM-bound kernel (GroupNorm backward) share the chip with an MFMA-bound one (3x3 convolution) when they are
launched on two HIP streams?  Prints each alone, both back to back on one stream, and both on two streams.  usage: python tools/corun_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K, ops  # noqa: E402

dt = torch.float16
B, H, C = 32, 256, 128
g_ = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(B, H, H, C, device="cuda", generator=g_).to(dt)
x2 = torch.randn(B, H, H, C, device="cuda", generator=g_).to(dt)
dy = torch.randn(B, H, H, C, device="cuda", generator=g_).to(dt)
gm, bt = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
_, mean, rstd = K.groupnorm_fwd(x, gm, bt)
P = ops.ConvWeights(torch.randn(C, C, 3, 3) * 0.03, torch.zeros(C), dt)
y = torch.empty(B, H, H, C, dtype=dt, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def conv():
    K.gemm(x2, P.w, y, B * H * H, C, 9 * C, ldw=9 * C, x_mode=K.OP_CONV3X3, bias=P.bias, conv=(H, H, C))


def gn():
    K.groupnorm_bwd(dy, x, gm, bt, mean, rstd)


CLK = {}


def timed(fn, iters=40, name=None):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0 = K.clock_sample()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    c1 = K.clock_sample()
    torch.cuda.synchronize()
    if name:
        CLK[name] = K.effective_clock_mhz(c0, c1)
    return e0.elapsed_time(e1) / iters * 1e3


def both_two_streams():
    main = torch.cuda.current_stream()
    s1.wait_stream(main)
    s2.wait_stream(main)
    with torch.cuda.stream(s1):
        conv()
        conv()
    with torch.cuda.stream(s2):
        gn()
        gn()
    main.wait_stream(s1)
    main.wait_stream(s2)


tc, tg = timed(conv, name="conv"), timed(gn, name="gn")
tser = timed(lambda: (conv(), conv(), gn(), gn()), name="serial")
tpar = timed(both_two_streams, name="two streams")
print(f"conv 128->128 @256^2 b{B}: {tc:7.1f} us | groupnorm_bwd: {tg:7.1f} us | 2 conv + 2 gn on one stream: {tser:7.1f} us | "
      f"conv stream || gn stream: {tpar:7.1f} us (ideal overlap {2 * max(tc, tg):7.1f})")
print("effective shader clock, MHz:", {k: round(v) for k, v in CLK.items()})
