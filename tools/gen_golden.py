"""Generate tests/golden/*.npz by running the REFERENCE's own Python (build container only).

The reference (/root/reference, read-only) is imported here — never copied — to produce
seeded input/output/gradient vectors for every hot-path piece that is importable
(SURVEY.md §8c).  Third-party modules absent from this image are stubbed with MagicMock so
that `import main` succeeds; only pure functions/classes of main.py that do not touch the
stubs are exercised.  The committed .npz files are data (inputs + expected outputs); this
script is the recipe that made them.  It cannot run on the GPU box (/root/reference is absent
there) and nothing at test/bench time needs it.

Usage:  python tools/gen_golden.py   (writes tests/golden/)
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _import_reference():
    os.environ["USE_HOROVOD"] = "false"
    sys.path.insert(0, REF)
    for name in ["clize", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                 "omegaconf", "kornia", "kornia.augmentation", "torch.utils.tensorboard",
                 "taming", "taming.models", "taming.models.cond_transformer", "taming.models.vqgan",
                 "taming.modules", "taming.modules.losses", "taming.modules.losses.lpips",
                 "clip", "clip.simple_tokenizer", "x_transformers", "horovod", "horovod.torch"]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    import cloob  # noqa
    import main  # noqa
    import mlp_mixer_pytorch  # noqa
    import vitgan  # noqa
    return main, mlp_mixer_pytorch, vitgan, cloob


def bf16_grid_(module_or_tensors):
    """Round weights to the bf16 grid so they can be stored exactly as uint16 (halves fixture size)."""
    ts = module_or_tensors.parameters() if isinstance(module_or_tensors, nn.Module) else module_or_tensors
    with torch.no_grad():
        for p in ts:
            p.copy_(p.bfloat16().float())


def pack_sd(sd, prefix):
    out = {}
    for k, v in sd.items():
        v = v.detach()
        if v.dtype == torch.float32:
            assert torch.equal(v.bfloat16().float(), v), k
            out[f"{prefix}/{k}"] = v.bfloat16().view(torch.int16).numpy().view(np.uint16)
        else:
            out[f"{prefix}/{k}"] = v.numpy()
    return out


def grads_of(module, prefix):
    return {f"{prefix}/{k}": p.grad.detach().numpy() for k, p in module.named_parameters()}


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_main, ref_mixer, ref_vitgan, ref_cloob = _import_reference()

    # ---- Mixer (mlp_mixer_pytorch.py:70-91) ------------------------------------------------
    torch.manual_seed(0)
    cfg = dict(input_dim=24, image_size=4, channels=8, patch_size=1, dim=16, depth=2)
    net = ref_mixer.Mixer(**cfg)
    bf16_grid_(net)
    x = torch.randn(3, 24, requires_grad=True)
    gw = torch.randn(3, 8, 4, 4)
    y = net(x)
    (y * gw).sum().backward()
    np.savez_compressed(os.path.join(OUT, "mixer.npz"), x=x.detach().numpy(), gw=gw.numpy(), y=y.detach().numpy(),
                        dx=x.grad.numpy(), **pack_sd(net.state_dict(), "sd"), **grads_of(net, "grad"))

    # ---- VitGAN Generator / SimpleGenerator (vitgan.py:221-305) ------------------------------
    torch.manual_seed(1)
    g = ref_vitgan.Generator(initialize_size=1, out_channels=8, input_dim=24, dim=12, num_heads=6, blocks=2)
    bf16_grid_(g)
    x = torch.randn(3, 24, requires_grad=True)
    gw = torch.randn(3, 8, 8, 8)
    y = g(x)
    (y * gw).sum().backward()
    np.savez_compressed(os.path.join(OUT, "vitgan.npz"), x=x.detach().numpy(), gw=gw.numpy(), y=y.detach().numpy(),
                        dx=x.grad.numpy(), **pack_sd(g.state_dict(), "sd"), **grads_of(g, "grad"))
    torch.manual_seed(2)
    sg = ref_vitgan.SimpleGenerator(size=4, dim=12, num_heads=6, blocks=2, out_channels=8, input_dim=24)
    bf16_grid_(sg)
    x = torch.randn(3, 24, requires_grad=True)
    gw = torch.randn(3, 8, 4, 4)
    y = sg(x)
    (y * gw).sum().backward()
    np.savez_compressed(os.path.join(OUT, "simple_vitgan.npz"), x=x.detach().numpy(), gw=gw.numpy(),
                        y=y.detach().numpy(), dx=x.grad.numpy(), **pack_sd(sg.state_dict(), "sd"),
                        **grads_of(sg, "grad"))

    # ---- CLIP ViT + text tower (cloob.py:412-553), tiny dims ---------------------------------
    torch.manual_seed(3)
    clip_cfg = dict(embed_dim=32, image_resolution=16, vision_layers=1, vision_width=128, vision_patch_size=8,
                    context_length=12, vocab_size=64, transformer_width=64, transformer_heads=2,
                    transformer_layers=2)
    clip = ref_cloob.CLIP(**clip_cfg).eval()
    bf16_grid_(clip)
    img = torch.randn(4, 3, 16, 16, requires_grad=True)
    tok = torch.zeros(3, 12, dtype=torch.long)
    for i, L in enumerate([3, 7, 10]):
        tok[i, 0] = 62
        tok[i, 1:L] = torch.randint(1, 62, (L - 1,))
        tok[i, L] = 63
    ei = clip.encode_image(img)
    gw = torch.randn_like(ei)
    (ei * gw).sum().backward()
    et = clip.encode_text(tok)
    np.savez_compressed(os.path.join(OUT, "clip.npz"), img=img.detach().numpy(), tok=tok.numpy(),
                        image_embed=ei.detach().numpy(), gw=gw.numpy(), dimg=img.grad.numpy(),
                        text_embed=et.detach().numpy(), **pack_sd(clip.state_dict(), "sd"))

    # ---- glue functions of main.py ---------------------------------------------------------
    torch.manual_seed(4)
    out = {}
    xq = torch.randn(2, 3, 3, 8, requires_grad=True)
    cb = torch.randn(32, 8)
    q = ref_main.vector_quantize(xq, cb)                                     # main.py:134-138
    gq = torch.randn_like(q)
    (q * gq).sum().backward()
    out.update(vq_x=xq.detach().numpy(), vq_codebook=cb.numpy(), vq_out=q.detach().numpy(), vq_g=gq.numpy(),
               vq_dx=xq.grad.numpy())
    xc = (torch.randn(5, 7) * 2).requires_grad_(True)
    yc = ref_main.clamp_with_grad(xc, -1.0, 1.5)                             # main.py:118-132
    gc = torch.randn_like(yc)
    (yc * gc).sum().backward()
    out.update(clamp_x=xc.detach().numpy(), clamp_y=yc.detach().numpy(), clamp_g=gc.numpy(), clamp_dx=xc.grad.numpy())
    a = torch.randn(4, 6)
    b = torch.randn(1, 6, requires_grad=True)
    r = ref_main.replace_grad(a, b)                                          # main.py:105-116
    gr = torch.randn_like(r)
    (r * gr).sum().backward()
    out.update(rg_a=a.numpy(), rg_out=r.detach().numpy(), rg_g=gr.numpy(), rg_db=b.grad.numpy())
    yt = torch.rand(2, 3, 6, 5)
    out.update(tv_x=yt.numpy(), tv=ref_main.tv_loss(yt).numpy())             # main.py:423-428
    mc = ref_main.MakeCutouts(cut_size=8, cutn=3, augs=["R"], pool=True, pool_size=8)   # main.py:154-229
    mc.noise_fac = 0
    xi = torch.rand(2, 3, 20, 20, requires_grad=True)
    co = mc(xi)
    gco = torch.randn_like(co)
    (co * gco).sum().backward()
    out.update(cut_x=xi.detach().numpy(), cut_out=co.detach().numpy(), cut_g=gco.numpy(), cut_dx=xi.grad.numpy())
    mc2 = ref_main.MakeCutouts(cut_size=8, cutn=2, augs=["R"], pool=True, pool_size=10)  # resize 10 -> 8 path
    mc2.noise_fac = 0
    co2 = mc2(xi.detach())
    out.update(cut2_out=co2.numpy())
    np.savez_compressed(os.path.join(OUT, "glue.npz"), **out)

    # ---- composed mini train step (main.py:729-832) -------------------------------------------
    torch.manual_seed(5)
    C, S = 8, 4
    mixer = ref_mixer.Mixer(input_dim=32, image_size=S, channels=C, patch_size=1, dim=16, depth=2)
    bf16_grid_(mixer)

    class FakeVQ:                                   # exposes what synth() touches (main.py:140-143)
        pass

    vq = FakeVQ()
    vq.quantize = types.SimpleNamespace(embedding=types.SimpleNamespace(weight=torch.randn(24, C)))
    dec = nn.Sequential(nn.Conv2d(C, 6, 3, padding=1), nn.SiLU(), nn.Upsample(scale_factor=4, mode="nearest"),
                        nn.Conv2d(6, 3, 3, padding=1)).requires_grad_(False)
    bf16_grid_(dec)
    vq.decode = dec
    cb = vq.quantize.embedding.weight
    z_min = cb.min(dim=0).values[None, :, None, None]                       # main.py:645-646
    z_max = cb.max(dim=0).values[None, :, None, None]
    cutn, cut_size = 3, 16
    mk = ref_main.MakeCutouts(cut_size=cut_size, cutn=cutn, augs=["R"], pool=True, pool_size=cut_size)
    mk.noise_fac = 0
    mean = torch.Tensor(ref_main.CLIP_MEAN).view(1, -1, 1, 1)
    std = torch.Tensor(ref_main.CLIP_STD).view(1, -1, 1, 1)
    perceptor = clip
    inp = tok                                                               # dataset gives (toks, toks) main.py:655
    inp_feats = perceptor.encode_text(inp).float()                          # main.py:733
    out_feats = perceptor.encode_text(inp).float()                          # main.py:737
    z = mixer(inp_feats)                                                    # :754
    z = z.contiguous().view(len(inp), C, S, S)                              # :756-757
    z = ref_main.clamp_with_grad(z, z_min.min(), z_max.max())               # :763
    xr = ref_main.synth(vq, z)                                              # :767
    xcut = mk(xr)                                                           # :796
    xcut = (xcut - mean) / std                                              # :797
    embed = perceptor.encode_image(xcut).float()                            # :799
    clip_dim = 32
    H = out_feats.repeat(cutn, 1).view(cutn, 1, len(inp), clip_dim)         # :801-802
    H = torch.nn.functional.normalize(H, dim=-1).view(-1, clip_dim)         # :803-805
    embed_n = torch.nn.functional.normalize(embed, dim=1)                   # :808
    dists = (H.sub(embed_n).norm(dim=-1).div(2).arcsin().pow(2).mul(2)).mean()   # :811
    dists.backward()                                                        # :832
    np.savez_compressed(
        os.path.join(OUT, "ministep.npz"), tok=inp.numpy(), codebook=cb.numpy(),
        z=z.detach().numpy(), xr=xr.detach().numpy(), embed=embed.detach().numpy(), loss=dists.detach().numpy(),
        cutn=np.int64(cutn), cut_size=np.int64(cut_size),
        **pack_sd(mixer.state_dict(), "mixer_sd"), **grads_of(mixer, "mixer_grad"),
        **pack_sd(dec.state_dict(), "dec_sd"))
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "total bytes", tot)


if __name__ == "__main__":
    main()
