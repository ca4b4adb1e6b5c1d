"""Round 6: run-to-run reproducibility of the cfg2 backward pass at full model size (Mixer 32 x 1024, VQGAN f16-16384 decoder, CLIP ViT-B/32,
per-GPU batch B, cutn 8, default augmentations, f16) — the check tools/r6/vitgan_determinism_old.py does for the VitGAN generator, on the
headline path: N forward + backward passes on identical inputs / draws, every gradient tensor against the first pass.
usage (GPU box): python tools/r6/step_determinism.py [B] [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import ops  # noqa: E402
from feed_forward_vqgan_clip_amd import vqgan as fvq  # noqa: E402
from feed_forward_vqgan_clip_amd.optim import FusedAdam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cdt = torch.float16
cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=1024, depth=32, dropout=0, cutn=8, batch_size=B, repeat=1, nb_noise=None,
                   diversity_coef=0, clip_model="ViT-B/32", model_type="mlp_mixer", vq_image_size=16)
torch.manual_seed(7)
net = fmain.build_model(cfg, 256).cuda().prepare(cdt)
vq = fvq.VQGAN(fvq.random_state_dict(fvq.F16_16384, seed=7), fvq.F16_16384, cdt)
perceptor = fclip.CLIP(fclip.random_state_dict(fclip.VIT_B32, seed=7), cdt)
opt = FusedAdam(net.parameters(), lr=cfg.lr)
opt.loss_scale = 4096.0
stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
tok = fmain.synthetic_tokens(B, seed=3).cuda()
g = torch.Generator(device="cuda").manual_seed(5)
facs = torch.rand(8 * B, device="cuda", generator=g) * 0.1
noise = torch.randn(8 * B, 3, 224, 224, device="cuda", generator=g)
prm = stepper.make_cutouts.draw_aug_params(8 * B, "cuda")


def relrms(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).pow(2).mean() / (b.pow(2).mean() + 1e-300)).sqrt())


def one():
    loss, mid = stepper.forward_loss(tok, facs=facs, noise=noise, aug_params=prm)
    opt.zero_grad()
    (loss * opt.loss_scale).backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    return float(loss), mid["indices"].clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


l0, i0, g0 = one()
ATOMIC = ("norm", ".bias")            # LayerNorm parameters and biases are accumulated through fp32 atomics: order-dependent by design
nbad = 0
for it in range(N):
    l, i, gg = one()
    diff = {k: relrms(gg[k], g0[k]) for k in gg if not any(a in k for a in ATOMIC)}
    diff = {k: v for k, v in diff.items() if v > 0}
    codes = int((i != i0).sum())
    if diff or codes or l != l0:
        nbad += 1
        worst = sorted(diff.items(), key=lambda kv: -kv[1])[:3]
        print(it, "loss", l, "vs", l0, "| codes that differ", codes, "| weight-gradient tensors that differ", len(diff), [(k, f"{v:.1e}") for k, v in worst], flush=True)
print(f"cfg2 size, batch {B}: {nbad} of {N} passes differ from the first in loss / codes / a weight gradient", flush=True)
