import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from feed_forward_vqgan_clip_amd import main as fmain, ops
class A:
    dtype, dim, depth, cutn, batch, model_type, vq_image_size, augs, grad_wire, keep_cpu_weights = \
        "bf16", 1024, 32, 8, 64, "mlp_mixer", 16, "default", "fp32", False
dev = torch.device("cuda:0")
cfg, stepper, _ = bench.build(A, dev)
toks = fmain.synthetic_tokens(64 * 60, seed=1).to(dev)
B=64
def run(off, n=8, prefetch=True):
    bs=[toks[(off+i)*B:(off+i+1)*B] for i in range(n+4)]
    for i in range(3): stepper(bs[i], next_inp=bs[i+1] if prefetch else None)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(3,3+n): stepper(bs[i], next_inp=bs[i+1] if prefetch else None)
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
print("side on  %.2f" % run(0))
ops.set_wgrad_side_stream(False)
print("side off %.2f" % run(12))
ops.set_wgrad_side_stream(True)
print("side on  %.2f" % run(24))
print("side on, no text prefetch %.2f" % run(36, prefetch=False))
