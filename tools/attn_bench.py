import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from feed_forward_vqgan_clip_amd import kernels as K
from tools.gemm_bench import timeit
B, T, H = 512, 50, 12
qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
do = torch.randn(B, T, H * 64, device="cuda").bfloat16()
print("attn fwd %.1f us" % (timeit(lambda: K.attn_small_fwd(qkv, H, 0.125)) * 1e6))
print("attn bwd %.1f us" % (timeit(lambda: K.attn_small_bwd(qkv, do, H, 0.125)) * 1e6))
