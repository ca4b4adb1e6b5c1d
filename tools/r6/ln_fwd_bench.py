"""LayerNorm forward at cfg2's shape (16384 x 1024 fp32 -> f16): us and TB/s per FFVC_LN_FWD_GRID (read once per process)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

rows, dim = 16384, 1024
x = torch.randn(rows, dim, device="cuda")
g, b = torch.randn(dim, device="cuda"), torch.randn(dim, device="cuda")
for _ in range(5):
    y = K.layernorm_fwd(x, g, b, torch.float16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    y = K.layernorm_fwd(x, g, b, torch.float16)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
yy = y[0] if isinstance(y, tuple) else y
ref = torch.nn.functional.layer_norm(x, (dim,), g, b, 1e-5)
err = (yy.float() - ref).abs().max().item()
print(f"FFVC_LN_FWD_GRID={os.environ.get('FFVC_LN_FWD_GRID', 'default')}: {us:7.1f} us  {rows * dim * 6 / us / 1e6:6.2f} TB/s  max err {err:.2e}")
