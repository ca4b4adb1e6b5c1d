"""Round 6: run-to-run reproducibility of the small-M GEMMs that split K inside the launch (gemm2_kernels.h SKFIX), alone and with a second
stream keeping the chip busy.  usage (GPU box): python tools/r6/skfix_stress.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.float16
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev).to(dt)
b = torch.randn(4096, 4096, device=dev).to(dt)
c = torch.empty(4096, 4096, device=dev, dtype=dt)
N_IT = int(os.environ.get("STRESS_N", "1500"))
for (M, N, Kd, out32) in [(512, 1024, 3064, False), (512, 1024, 4096, True), (512, 1024, 1024, False), (512, 4096, 1024, False), (512, 1024, 4096, False)]:
    g = torch.Generator().manual_seed(M + N + Kd)
    xs = [torch.randn(M, Kd, generator=g).to(dt).to(dev) for _ in range(3)]          # alternating inputs: a stale partial tile of the
    w = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(dt).to(dev)                  # previous launch would be a different value
    res = torch.randn(M, N, generator=g).to(dev)
    for busy in (False, True):
        refs = [None, None, None]
        bad = 0
        worst = 0.0
        for it in range(N_IT):
            x = xs[it % 3]
            y = torch.empty(M, N, device=dev, dtype=torch.float32 if out32 else dt)
            if busy and it % 4 == 0:
                with torch.cuda.stream(side):
                    K.gemm(a, b, c, 4096, 4096, 4096, ldx=4096, ldw=4096)
            if out32:
                K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, residual=res)
            else:
                K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd)
            if refs[it % 3] is None:
                refs[it % 3] = y.clone()
            elif not torch.equal(y, refs[it % 3]):
                bad += 1
                worst = max(worst, float((y.float() - refs[it % 3].float()).abs().max()))
        torch.cuda.synchronize()
        print(f"{M}x{N}x{Kd} out {'fp32+res' if out32 else 'f16'} busy={busy}: {bad} of {N_IT} launches differ from their reference (max abs diff {worst:.3e})", flush=True)
