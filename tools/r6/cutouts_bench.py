"""Round 6: ffvc_cutouts_fwd / _bwd at cfg2's size — the (adaptive average + adaptive max) / 2 pooling 256 -> 224 of 64 decoded images that
feeds the augmentation chain (cutn 1, one 224-pixel "patch").  usage (GPU box): python tools/r6/cutouts_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

B, H, cut = 64, 256, 224
g = torch.Generator().manual_seed(1)
xr = torch.rand(B, H, H, 3, generator=g).cuda()
mean, std = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
y = K.cutouts_fwd(xr, cut, 1, cut, mean, std, torch.float32)
gout = torch.randn(*y.shape, generator=g).cuda()
d = K.cutouts_bwd(xr, gout, cut, 1, cut, std)
torch.cuda.synchronize()
tf = timeit(lambda: K.cutouts_fwd(xr, cut, 1, cut, mean, std, torch.float32), iters=20)
tb = timeit(lambda: K.cutouts_bwd(xr, gout, cut, 1, cut, std), iters=20)
print(f"lib={os.path.basename(os.environ.get('FFVC_LIB', 'default'))} cutouts_fwd {tf * 1e6:7.1f} us  cutouts_bwd {tb * 1e6:7.1f} us   "
      f"fwd checksum {float(y.double().sum()):.6f}  bwd checksum {float(d.double().sum()):.6f} abs {float(d.double().abs().sum()):.4f}")
