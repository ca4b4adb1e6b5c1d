"""ffvc_augment_bwd at cfg2's size (512 cutouts of 224 x 224 from 64 pooled images): tiled LDS form vs direct global atomics.
usage: FFVC_AUG_BWD_TILED=0|1 python tools/aug_bwd_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import augment as A  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

B, cutn, S, P = 64, 8, 224, 32
g = torch.Generator().manual_seed(1)
prm = {k: v.cuda() for k, v in A.draw_params(cutn * B, S, generator=g).items()}
pooled = torch.rand(B, 3, S, S, generator=g).cuda()
gout = torch.randn(cutn * B, (S // P) ** 2, 3 * P * P, generator=g).cuda().half()
std = (0.26862954, 0.26130258, 0.27577711)
run = lambda: K.augment_bwd(gout, prm["pinv"], prm["ainv"], prm["cmat"], prm["erase"], B, S, cutn, P, std, pooled=pooled,  # noqa: E731
                            coff=prm["coff"], cj=prm["cj"])
d = run()
torch.cuda.synchronize()
t = timeit(run, iters=10)
print(f"tiled={os.environ.get('FFVC_AUG_BWD_TILED', '1')} augment_bwd {t * 1e6:8.1f} us  checksum {float(d.double().sum()):.6f} abs {float(d.double().abs().sum()):.4f}")
