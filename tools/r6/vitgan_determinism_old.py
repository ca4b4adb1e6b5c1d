"""Round 6: the race hunt of vitgan_determinism2.py against ANOTHER checkout of the package (first argument: its root), e.g. the commit
before the cfg3 launch diet — is the run-to-run spread older than that change?"""
import os
import sys

import torch

root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
from feed_forward_vqgan_clip_amd import ops  # noqa: E402
from feed_forward_vqgan_clip_amd.mappers import Generator  # noqa: E402

F16 = torch.float16


def relrms(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).pow(2).mean() / (b.pow(2).mean() + 1e-300)).sqrt())


def grads(seed=7):
    torch.manual_seed(seed)
    net = Generator(initialize_size=2, out_channels=4, input_dim=64, dim=1024, num_heads=6, blocks=9).cuda().prepare(F16)
    x = torch.randn(32, 64, generator=torch.Generator().manual_seed(seed + 1)).cuda().requires_grad_(True)
    out = net(x)
    r = torch.randn(*out.shape, generator=torch.Generator().manual_seed(seed + 2)).cuda()
    net._ffvc_arena.zero_grad()
    (out * r).sum().backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    g["__dx"] = x.grad.detach().clone()
    return g


print("package:", os.path.dirname(ops.__file__), flush=True)
ATOMIC = ("norm1.", "norm2.", "sln_norm.", ".bias")
ref = grads()
nbad = 0
for it in range(int(os.environ.get("HUNT_N", "60"))):
    g = grads()
    bad = {k: relrms(g[k], ref[k]) for k in g if not any(a in k for a in ATOMIC)}
    if any(v > 1e-6 for v in bad.values()):
        nbad += 1
        d = (g["__dx"] - ref["__dx"]).abs()
        print(it, "perturbed; __dx samples that differ", torch.nonzero(d.amax(dim=1) > 0).flatten().tolist(), flush=True)
print("runs with a perturbed backward:", nbad, flush=True)
