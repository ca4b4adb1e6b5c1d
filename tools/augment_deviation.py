"""How far is the ONE-launch fused augmentation (feed_forward_vqgan_clip_amd/augment.py, the benchmark's default) from kornia 0.5.10's
sequential nn.Sequential (oracle/kornia_aug.py, an independent restatement), on the SAME raw draws, at cfg2's sizes?  CPU only.

  kornia      pooled.repeat(cutn) -> RandomAffine (bilinear, border) -> RandomPerspective (bilinear, zeros) -> ColorJitter -> RandomErasing
  fused       one bilinear interpolation of the composed map  (augment.plan(chain))
  sequential  one launch per warp                              (augment.plan(chain, sequential=True)) — must equal kornia to rounding

Reported: image-space deviation over the N = cutn * B cutouts and the deviation of the spherical CLIP loss (ViT-B/32, random weights,
same text features), for the decoder's own output (random-weight VQGAN: broadband texture, the worst case for a second interpolation)
and for a smooth synthetic image batch.  usage: python tools/augment_deviation.py > profiles/r04_augment_deviation.txt"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import augment as A  # noqa: E402
from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import vqgan as fvq  # noqa: E402
from oracle import clip as oclip  # noqa: E402
from oracle import kornia_aug as ka  # noqa: E402
from oracle import mappers as omap  # noqa: E402
from oracle import step as ostep  # noqa: E402

torch.set_num_threads(8)
B, cutn, S = 4, 8, 224
N = B * cutn
g = torch.Generator().manual_seed(2024)
arch, quick = fmain.clip_arch("ViT-B/32")
clip_sd = fclip.random_state_dict(arch, seed=1234)
vq_sd = fvq.random_state_dict(fvq.F16_16384, seed=1234)
tok = fmain.synthetic_tokens(B, seed=99)
with torch.no_grad():
    feats = oclip.encode_text(clip_sd, tok, 8, quick).float()
    torch.manual_seed(1234)
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=1024, depth=32, dropout=0, cutn=cutn, batch_size=B, repeat=1, nb_noise=None,
                       diversity_coef=0, clip_model="ViT-B/32", model_type="mlp_mixer", vq_image_size=16)
    msd = {k: v.detach() for k, v in fmain.build_model(cfg, 256).state_dict().items()}
    z = omap.mixer_forward(msd, feats, image_size=16, channels=256, depth=32)
    cb = vq_sd["quantize.embedding.weight"]
    xr_dec = ostep.synth(vq_sd, ostep.clamp_with_grad(z, cb.min().item(), cb.max().item()), fvq.F16_16384, None)
    # a smooth batch: low-pass filtered noise stretched to [0, 1]
    lo = F.interpolate(torch.rand(B, 3, 14, 14, generator=g), (256, 256), mode="bicubic", align_corners=False)
    xr_smooth = ((lo - lo.amin((1, 2, 3), keepdim=True)) / (lo.amax((1, 2, 3), keepdim=True) - lo.amin((1, 2, 3), keepdim=True))).clamp(0, 1)

mean = torch.tensor(ostep.CLIP_MEAN).view(1, 3, 1, 1)
std = torch.tensor(ostep.CLIP_STD).view(1, 3, 1, 1)


def loss_of(batch):
    with torch.no_grad():
        e = oclip.encode_image(clip_sd, ((batch - mean) / std).float(), 12, quick).float()
        return float(ostep.spherical_loss(e, feats, cutn))


def run_plan(pooled, segs):
    x, c = pooled, cutn
    for kind, q in segs:
        x = ostep.augment_reference(x, q["pinv"].double(), q["ainv"].double(), q["cmat"].double(), q["erase"], c, coff=q["coff"].double(),
                                    cj=q.get("cj"))
        c = 1
    return x


print(f"# augmentation deviation, cfg2 sizes: B={B} images 256x256 -> pooled {S}x{S}, cutn={cutn} -> N={N} cutouts, default chain {A.DEFAULT}")
print("# columns: image source | draws | rel-rms(fused vs kornia) | 99.9th pct abs | max abs | rel-rms(sequential plan vs kornia) | "
      "loss kornia | loss fused | rel loss dev fused | rel loss dev sequential")
for name, xr in (("decoder output (random-weight VQGAN)", xr_dec), ("smooth synthetic", xr_smooth)):
    pooled = ((F.adaptive_avg_pool2d(xr, S) + F.adaptive_max_pool2d(xr, S)) / 2).double()
    for seed in (1, 2, 3):
        gg = torch.Generator().manual_seed(seed)
        chain = A.draw_chain(N, S, A.DEFAULT, gg)
        want = ka.apply_chain(pooled.repeat(cutn, 1, 1, 1), chain)
        fused = run_plan(pooled, A.plan(chain, N, S))
        seq = run_plan(pooled, A.plan(chain, N, S, sequential=True))
        d = (fused - want).abs()
        rr = float((fused - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        rs = float((seq - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        lk, lf, ls = loss_of(want), loss_of(fused), loss_of(seq)
        print(f"{name:38s} | seed {seed} | {rr:.3e} | {float(d.flatten().kthvalue(int(0.999 * d.numel())).values):.3e} | {float(d.max()):.3e} | "
              f"{rs:.1e} | {lk:.6f} | {lf:.6f} | {abs(lf - lk) / lk:.2e} | {abs(ls - lk) / lk:.1e}", flush=True)
# which operator carries the fused deviation: the same comparison with single operators and pairs
print("# per-operator check (decoder output, seed 1): chain | rel-rms fused vs kornia")
pooled = ((F.adaptive_avg_pool2d(xr_dec, S) + F.adaptive_max_pool2d(xr_dec, S)) / 2).double()
for augs in (("Af",), ("Pe",), ("Ji",), ("Er",), ("Af", "Pe"), ("Af", "Pe", "Ji", "Er")):
    chain = A.draw_chain(N, S, augs, torch.Generator().manual_seed(1))
    want = ka.apply_chain(pooled.repeat(cutn, 1, 1, 1), chain)
    fused = run_plan(pooled, A.plan(chain, N, S))
    print(f"{'+'.join(augs):14s} | {float((fused - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()):.3e}", flush=True)
