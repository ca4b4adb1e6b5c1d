"""Developer probe: host (Python + ctypes + autograd) CPU time per train step vs the GPU's wall time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    dtype, dim, depth, cutn, batch, model_type, vq_image_size, augs, grad_wire, keep_cpu_weights = \
        "f16", 1024, 32, 8, 64, "mlp_mixer", 16, "default", "fp32", False
    clip_model, clip_fp8, loss_scale, prefetch_text = "ViT-B/32", False, 4096.0, True


if len(sys.argv) > 1 and sys.argv[1] == "cfg3":           # VitGAN 32 x 1024 at a per-GPU batch of 32
    A.model_type, A.batch = "vitgan", 32


from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402

dev = torch.device("cuda:0")
cfg, stepper, _ = bench.build(A, dev)
B = A.batch
toks = fmain.synthetic_tokens(B * 20, seed=1).to(dev)
for i in range(3):
    stepper(toks[i * B:(i + 1) * B])
torch.cuda.synchronize()
c0, t0 = time.process_time(), time.perf_counter()
for i in range(3, 13):
    stepper(toks[i * B:(i + 1) * B])
c1, t1 = time.process_time(), time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host CPU time/step {1e3 * (c1 - c0) / 10:.1f} ms | host wall until enqueued {1e3 * (t1 - t0) / 10:.1f} ms | "
      f"GPU wall/step {1e3 * (t2 - t0) / 10:.1f} ms")

# one step enqueued into an EMPTY queue: no back-pressure from the GPU, so this is the host's own cost per step
one = []
for i in range(13, 18):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stepper(toks[i * B:(i + 1) * B])
    one.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
print("host wall to enqueue ONE step into an empty queue (ms):", " ".join(f"{v:.1f}" for v in one))
