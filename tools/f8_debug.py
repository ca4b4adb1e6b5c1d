import torch
from feed_forward_vqgan_clip_amd import kernels as K, ops, vqgan as fvq
F16 = torch.float16
g = torch.Generator().manual_seed(21)
vsd = fvq.random_state_dict(fvq.F16_16384, seed=34)
inputs = [torch.randn(4, 16, 16, 256, generator=g).cuda() for _ in range(3)]
gws = [torch.randn(4, 256, 256, 3, generator=g).cuda() for _ in range(3)]
cnt = {"fwd": 0, "bwd": 0, "bwd_only": 0}
_gf, _gb = K.groupnorm_fwd, K.groupnorm_bwd
def gf(*a, **k):
    if k.get("f8") is not None: cnt["fwd"] += 1
    return _gf(*a, **k)
def gb(*a, **k):
    if k.get("f8") is not None:
        cnt["bwd"] += 1
        cnt["bwd_only"] += int(bool(k.get("f8_only")))
    return _gb(*a, **k)
K.groupnorm_fwd, K.groupnorm_bwd = gf, gb
def rel(a, b): return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
res = {}
for name, on in (("off", False), ("off2", False), ("on", True)):
    ops._F8_PRODUCER = on
    model = fvq.VQGAN(vsd, fvq.F16_16384, F16, fp8=True)
    outs = []
    for x, gw in zip(inputs, gws):
        xi = x.clone().requires_grad_(True)
        K.fp8_flush_updates()
        for k in cnt: cnt[k] = 0
        y = model.decode_nhwc(xi.to(F16))
        (y.float() * gw).sum().backward()
        outs.append((y.detach().float(), xi.grad.float()))
        print(name, dict(cnt))
    res[name] = outs
for it in range(3):
    (ya, ga), (yn, gn), (yb, gb_) = res["off"][it], res["off2"][it], res["on"][it]
    print(it, "noise", rel(yn, ya), rel(gn, ga), "on", rel(yb, ya), rel(gb_, ga))
