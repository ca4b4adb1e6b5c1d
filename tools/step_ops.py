"""Developer probe: which torch-level ops of one cfg2 train step launch the SMALL device kernels (copies, fills, element-wise) — torch.profiler
table of aten ops by call count, plus the device kernels shorter than 12 us grouped by name.  usage: python tools/step_ops.py"""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402


class A:
    dtype, dim, depth, cutn, batch, model_type, vq_image_size, augs, grad_wire, keep_cpu_weights = \
        "f16", 1024, 32, 8, 64, "mlp_mixer", 16, "default", "fp32", False
    clip_model, clip_fp8, loss_scale, prefetch_text, dec_fp8 = "ViT-B/32", False, 4096.0, True, False
    grad_wire_tail, augment_fused = "fp32", False


dev = torch.device("cuda:0")
cfg, stepper, _ = bench.build(A, dev)
toks = fmain.synthetic_tokens(64 * 8, seed=1).to(dev)
for i in range(3):
    stepper(toks[i * 64:(i + 1) * 64], next_inp=toks[(i + 1) * 64:(i + 2) * 64])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    stepper(toks[3 * 64:4 * 64], next_inp=toks[4 * 64:5 * 64])
    torch.cuda.synchronize()
ev = prof.events()
aten = Counter()
small = Counter()
small_t = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::"):
        aten[e.name] += 1
    if e.device_type == torch.autograd.DeviceType.CUDA:
        us = e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
        if us < 12:
            small[e.name[:70]] += 1
            small_t[e.name[:70]] += us
print("aten ops by count:", aten.most_common(25))
print("device kernels < 12 us: total", sum(small.values()), "launches,", round(sum(small_t.values()) / 1e3, 2), "ms")
for k, n in small.most_common(25):
    print(f"  {n:5d} x {small_t[k] / max(n, 1):5.1f} us  {k}")
# call sites of the copies
sites = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros") and e.stack:
        fr = [s for s in e.stack if "feed_forward_vqgan_clip_amd" in s or "bench.py" in s]
        sites[(e.name, fr[0] if fr else (e.stack[0] if e.stack else "?"))] += 1
for (name, site), n in sites.most_common(30):
    print(f"  {n:4d} {name:12s} {site[-110:]}")
