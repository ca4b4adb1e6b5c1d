"""conv_row_kernel variants at the decoder's shapes: time + a checksum of the output (run once per FFVC_CR_WPF value; equal
checksums = bit-identical results, the variants only differ in staging).  usage: FFVC_CR_WPF=0|1 python tools/conv_wpf.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dt = torch.float16
g = torch.Generator(device="cuda").manual_seed(3)
for (H, Cin, Cout, ups, gn) in [(256, 128, 128, False, False), (256, 128, 128, False, True), (128, 256, 256, False, False),
                                (128, 256, 128, False, True), (128, 128, 128, False, False), (64, 256, 256, False, True),
                                (256, 128, 128, True, True)]:
    Hin = H // 2 if ups else H
    x = torch.randn(B, Hin, Hin, Cin, device="cuda", generator=g).to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device="cuda", generator=g) * 0.05).to(dt)
    b = torch.randn(Cout, device="cuda", generator=g)
    y = torch.empty(B, H, H, Cout, device="cuda", dtype=dt)
    sums = torch.zeros(B, 32, 2, dtype=torch.float64, device="cuda")

    def run():
        K.gemm(x, w, y, B * H * H, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, conv=(H, H, Cin), bias=b,
               flags=K.F_UPSAMPLE2X if ups else 0, gn_sums=(sums, H * H, Cout // 32) if gn else None)
    sums.zero_()
    run()
    torch.cuda.synchronize()
    chk = y.view(torch.int16).to(torch.int64).sum().item()
    schk = sums.sum().item()
    t = timeit(run, iters=10)
    print(f"WPF={os.environ.get('FFVC_CR_WPF', '0')} conv b{B} {H}^2 {Cin}->{Cout}{' ups' if ups else ''}{' gn' if gn else ''}: "
          f"{2.0 * B * H * H * Cout * 9 * Cin / t / 1e12:7.1f} TFLOP/s {t * 1e3:7.3f} ms  checksum {chk} {schk:.6e}", flush=True)
