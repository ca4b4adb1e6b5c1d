"""ffvc_augment_{,seq_}{fwd,bwd} at cfg2's size (512 cutouts of 224 x 224 from 64 pooled images): the composed single interpolation
(opt-in), kornia's two interpolations in ONE launch (default, round 5), and the same as two launches.  us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import augment as A  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

B, cutn, S, P = 64, 8, 224, 32
n = B * cutn
g = torch.Generator().manual_seed(1)
chain = A.draw_chain(n, S, A.DEFAULT, g)
fused = A.to_device(A.plan(chain, n, S, S, sequential=False), "cuda")[0][1]
seq = A.to_device(A.plan(chain, n, S, S, sequential=True), "cuda")[0][1]
keep, A._merge_sequential = A._merge_sequential, (lambda s: s)
two = A.to_device(A.plan(chain, n, S, S, sequential=True), "cuda")
A._merge_sequential = keep
pooled = torch.rand(B, 3, S, S, generator=g).cuda()
gout = torch.randn(n, (S // P) ** 2, 3 * P * P, generator=g).cuda().half()
mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def fwd(prm, src=pooled, c=cutn, patch=P, dt=torch.float16, mn=mean, sd=std):
    return K.augment_fwd(src, prm["pinv"], prm["ainv"], prm["cmat"], prm["erase"], c, patch, mn, sd, dt, coff=prm.get("coff"), cj=prm.get("cj"),
                         seq=bool(prm.get("seq", 0)))


def bwd(prm, go=gout, src=pooled, c=cutn, patch=P, sd=std, b=B):
    return K.augment_bwd(go, prm["pinv"], prm["ainv"], prm["cmat"], prm["erase"], b, S, c, patch, sd, pooled=src, coff=prm.get("coff"),
                         cj=prm.get("cj"), seq=bool(prm.get("seq", 0)))


for name, prm in (("composed single interpolation (opt-in)", fused), ("two interpolations, ONE launch (default)", seq)):
    tf, tb = timeit(lambda: fwd(prm), iters=10), timeit(lambda: bwd(prm), iters=10)
    print(f"{name:44s} fwd {tf * 1e6:8.1f} us   bwd {tb * 1e6:8.1f} us")
inter = fwd(two[0][1], patch=S, dt=torch.float32, mn=(0.0, 0.0, 0.0), sd=(1.0, 1.0, 1.0)).view(n, 3, S, S)
gint = torch.randn(n, 1, 3 * S * S).cuda()
one3 = (1.0, 1.0, 1.0)
tf = timeit(lambda: fwd(two[0][1], patch=S, dt=torch.float32, mn=(0.0, 0.0, 0.0), sd=one3), iters=10) + timeit(lambda: fwd(two[1][1], src=inter, c=1), iters=10)
tb = timeit(lambda: bwd(two[1][1], src=inter, c=1, b=n), iters=10) + timeit(lambda: bwd(two[0][1], go=gint, patch=S, sd=one3), iters=10)
print(f"{'two interpolations, two launches':44s} fwd {tf * 1e6:8.1f} us   bwd {tb * 1e6:8.1f} us")
