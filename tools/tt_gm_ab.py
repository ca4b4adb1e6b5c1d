import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from feed_forward_vqgan_clip_amd import kernels as K
from tools.gemm_bench import timeit
dt = torch.float16
for (M, N, Kd) in [(4096, 1024, 16384), (1024, 4096, 16384), (3072, 768, 25600), (768, 3072, 25600)]:
    xt = torch.randn(Kd, M, device="cuda").to(dt); wt = torch.randn(Kd, N, device="cuda").to(dt)
    y = torch.zeros(M, N, device="cuda")
    K.set_option("gemm2_tile", 512)
    t = timeit(lambda: K.gemm_splitk_accumulate(xt, wt, y, M, N, Kd, 4, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS), iters=10)
    print(f"gm={os.environ.get('FFVC_TILE_GM','dflt')} TT {M}x{N}x{Kd} sk4: {2.0*M*N*Kd/t/1e12:7.1f} TF {t*1e6:7.1f} us")
