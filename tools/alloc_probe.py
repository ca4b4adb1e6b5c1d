"""Developer probe: is the caching allocator the slow part of the host side (device mallocs / event polling)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import ops  # noqa: E402


class A:
    dtype, dim, depth, cutn, batch, model_type, vq_image_size, augs, grad_wire, keep_cpu_weights = \
        "bf16", 1024, 32, 8, 64, "mlp_mixer", 16, "default", "fp32", False


dev = torch.device("cuda:0")
cfg, stepper, _ = bench.build(A, dev)
toks = fmain.synthetic_tokens(64 * 30, seed=1).to(dev)


def run(n, off):
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for i in range(n):
        stepper(toks[(off + i) * 64:(off + i + 1) * 64])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    keys = ["num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated"]
    print({k: s1.get(k, 0) - s0.get(k, 0) for k in keys}, f"host {1e3 * (t1 - t0) / n:.1f} ms/step, gpu {1e3 * (t2 - t0) / n:.1f} ms/step",
          f"reserved {s1['reserved_bytes.all.current'] / 2**30:.1f} GiB")


run(3, 0)
run(5, 3)
ops.set_wgrad_side_stream(False)
run(2, 8)
run(5, 10)
