import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from feed_forward_vqgan_clip_amd import kernels as K
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, T, D, O = 64, 256, 1024, 1024
NB = 12
Wm = torch.randn(O, T, device=dev).to(dt)
xs = [torch.randn(B, T, D, device=dev).to(dt) for _ in range(NB)]
outs = [torch.empty(B, O, D, device=dev, dtype=dt) for _ in range(NB)]
pres = [torch.empty(B, O, D, device=dev, dtype=dt) for _ in range(NB)]
bias = torch.randn(O, device=dev)
def call(i, mode):
    xn, out, pre = xs[i % NB], outs[i % NB], pres[i % NB]
    if mode == "plain":
        K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, batch=B, wb=(T*D,0), yb=(O*D,0))
    elif mode == "gelu+preact":
        K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, batch=B, wb=(T*D,0), yb=(O*D,0), ab=(O*D,0), bias=bias, act=K.ACT_GELU, aux=pre, ldaux=D, flags=K.F_WRITE_PREACT|K.F_BIAS_ALONG_M)
    else:
        K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, batch=B, wb=(T*D,0), yb=(O*D,0), ab=(O*D,0), act=K.ACT_GELU, aux=pre, ldaux=D, flags=K.F_MUL_ACT_GRAD)
for mode in ("plain", "gelu+preact", "mul_act_grad"):
    for rot in (1, NB):
        for i in range(NB): call(i if rot > 1 else 0, mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(36): call(i if rot > 1 else 0, mode)
        e1.record(); torch.cuda.synchronize()
        print(f"{mode:14s} {'rotating (cold)' if rot > 1 else 'same buffers  '} {e0.elapsed_time(e1)/36*1e3:7.1f} us")
