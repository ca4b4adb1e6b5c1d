"""Developer probe: host (Python + ctypes + autograd) CPU time per train step vs the GPU's wall time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    dtype, dim, depth, cutn, batch, model_type, vq_image_size, augs, grad_wire, keep_cpu_weights = \
        "bf16", 1024, 32, 8, 64, "mlp_mixer", 16, "default", "fp32", False


from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402

dev = torch.device("cuda:0")
cfg, stepper, _ = bench.build(A, dev)
toks = fmain.synthetic_tokens(64 * 20, seed=1).to(dev)
for i in range(3):
    stepper(toks[i * 64:(i + 1) * 64])
torch.cuda.synchronize()
c0, t0 = time.process_time(), time.perf_counter()
for i in range(3, 13):
    stepper(toks[i * 64:(i + 1) * 64])
c1, t1 = time.process_time(), time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host CPU time/step {1e3 * (c1 - c0) / 10:.1f} ms | host wall until enqueued {1e3 * (t1 - t0) / 10:.1f} ms | "
      f"GPU wall/step {1e3 * (t2 - t0) / 10:.1f} ms")
