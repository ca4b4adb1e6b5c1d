"""Developer probe: cProfile of the host side of the train step (where do the ~140 ms of enqueue time go)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402


class A:
    dtype, dim, depth, cutn, batch, model_type, vq_image_size, augs, grad_wire, keep_cpu_weights = \
        "f16", 1024, 32, 8, 64, "mlp_mixer", 16, "default", "fp32", False
    clip_model, clip_fp8, loss_scale, prefetch_text = "ViT-B/32", False, 4096.0, True


dev = torch.device("cuda:0")
cfg, stepper, _ = bench.build(A, dev)
toks = fmain.synthetic_tokens(64 * 12, seed=1).to(dev)
for i in range(3):
    stepper(toks[i * 64:(i + 1) * 64])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(3, 8):
    stepper(toks[i * 64:(i + 1) * 64])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
