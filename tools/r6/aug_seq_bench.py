"""Round 6: ffvc_augment_seq_fwd / _bwd at cfg2's size (512 cutouts of 224 x 224 from 64 pooled images, the default augmentation set in
kornia's order: affine, then perspective + colour jitter + erasing in one launch).  usage (GPU box): python tools/r6/aug_seq_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import augment as A  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

B, cutn, S, P = 64, 8, 224, 32
g = torch.Generator().manual_seed(1)
chain = A.draw_chain(cutn * B, S, ("Af", "Pe", "Ji", "Er"), generator=g, src_size=S, device="cpu")
segs = A.plan(chain, cutn * B, S, S, sequential=True)
assert len(segs) == 1, [k for k, _ in segs]
prm = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in segs[0][1].items()}
seq = bool(prm.get("seq", 0))
pooled = torch.rand(B, 3, S, S, generator=g).cuda()
gout = torch.randn(cutn * B, (S // P) ** 2, 3 * P * P, generator=g).cuda().half()
mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
noise = torch.randn(cutn * B, 3, S, S, generator=g).cuda()
facs = torch.rand(cutn * B, generator=g).cuda() * 0.1
bwd = lambda: K.augment_bwd(gout, prm["pinv"], prm["ainv"], prm["cmat"], prm["erase"], B, S, cutn, P, std, pooled=pooled,  # noqa: E731
                            coff=prm.get("coff"), cj=prm.get("cj"), seq=seq)
fwd = lambda: K.augment_fwd(pooled, prm["pinv"], prm["ainv"], prm["cmat"], prm["erase"], cutn, P, mean, std, torch.float16, noise=noise,  # noqa: E731
                            facs=facs, coff=prm.get("coff"), cj=prm.get("cj"), seq=seq)
d = bwd()
y = fwd()
torch.cuda.synchronize()
tb, tf = timeit(bwd, iters=10), timeit(fwd, iters=10)
print(f"seq={seq} lib={os.path.basename(os.environ.get('FFVC_LIB', 'default'))} augment_bwd {tb * 1e6:8.1f} us  augment_fwd {tf * 1e6:8.1f} us   "
      f"bwd checksum {float(d.double().sum()):.6f} abs {float(d.double().abs().sum()):.4f}  fwd abs {float(y.double().abs().sum()):.2f}")
