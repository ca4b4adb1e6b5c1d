"""BASELINE cfg2 at FULL model sizes (Mixer 32x1024, VQGAN f16-16384, CLIP ViT-B/32, 256x256, cutn 8) where the CPU
oracle is too slow to be the checker: size-independent properties of the path instead."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import ops  # noqa: E402
from feed_forward_vqgan_clip_amd import vqgan as fvq  # noqa: E402
from feed_forward_vqgan_clip_amd.optim import FusedAdam  # noqa: E402

B = 8


@pytest.fixture(scope="module")
def full(cuda):
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=1024, depth=32, dropout=0, cutn=8, batch_size=B, repeat=1,
                       nb_noise=None, diversity_coef=0, clip_model="ViT-B/32", model_type="mlp_mixer", vq_image_size=16)
    torch.manual_seed(7)
    net = fmain.build_model(cfg, 256).cuda().prepare(torch.bfloat16)
    vq = fvq.VQGAN(fvq.random_state_dict(fvq.F16_16384, seed=7), fvq.F16_16384, torch.bfloat16)
    perceptor = fclip.CLIP(fclip.random_state_dict(fclip.VIT_B32, seed=7), torch.bfloat16)
    opt = FusedAdam(net.parameters(), lr=cfg.lr)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    tok = fmain.synthetic_tokens(B, seed=3).cuda()
    return cfg, net, vq, perceptor, opt, stepper, tok


def test_step_properties(full):
    cfg, net, vq, perceptor, opt, stepper, tok = full
    loss, mid = stepper.forward_loss(tok)
    # spherical distance loss: 2*asin(d/2)^2 with d in [0, 2]  ->  [0, pi^2/2]   (main.py:801-811)
    assert 0.0 <= loss.item() <= math.pi ** 2 / 2
    assert tuple(mid["z"].shape) == (B, 256, 16, 16) and tuple(mid["xr"].shape) == (B, 256, 256, 3)
    assert mid["xr"].min().item() >= 0.0 and mid["xr"].max().item() <= 1.0          # clamp_with_grad(…, 0, 1)
    assert tuple(mid["embed"].shape) == (8 * B, 512) and torch.isfinite(mid["embed"]).all()
    idx = mid["indices"].view(-1)
    assert idx.min().item() >= 0 and idx.max().item() < 16384
    # VQ idempotence: the chosen code is its own nearest code
    cb = vq.codebook
    zq = cb[idx]
    _, idx2 = ops.vector_quantize(zq, cb, vq.cnorm, torch.float32)
    assert torch.equal(idx2.view(-1), idx)
    opt.zero_grad()
    loss.backward()
    g = net._ffvc_arena.grads
    assert torch.isfinite(g).all() and g.abs().max().item() > 0
    # every parameter tensor received a gradient (token-mix bias in front of a LayerNorm has a zero true gradient)
    dead = [k for k, p in net.named_parameters() if p.grad.abs().max().item() == 0]
    assert len(dead) == 0, dead[:5]


def test_same_inputs_same_loss_and_descent(full):
    cfg, net, vq, perceptor, opt, stepper, tok = full
    g = torch.Generator(device="cuda").manual_seed(5)
    facs = torch.rand(8 * B, device="cuda", generator=g) * 0.1
    noise = torch.randn(8 * B, 3, 224, 224, device="cuda", generator=g)
    prm = stepper.make_cutouts.draw_aug_params(8 * B, "cuda")
    kw = dict(facs=facs, noise=noise, aug_params=prm)
    l0, _ = stepper.forward_loss(tok, **kw)
    l1, _ = stepper.forward_loss(tok, **kw)
    assert abs(l0.item() - l1.item()) < 1e-3 * l0.item()   # repeatable up to the order of the atomic partial sums
    losses = [l0.item()]
    for _ in range(4):
        loss, _ = stepper(tok, **kw)
        losses.append(loss.item())
    final, _ = stepper.forward_loss(tok, **kw)
    assert final.item() < losses[0], losses           # Adam on a fixed batch goes downhill


def test_batch_rows_are_independent(full):
    """Data parallelism shards prompts: a prompt's latent / image must not depend on its batch neighbours."""
    cfg, net, vq, perceptor, opt, stepper, tok = full
    with torch.no_grad():
        _, a = stepper.forward_loss(tok, facs=torch.zeros(8 * B, device="cuda"), noise=torch.zeros(8 * B, 3, 224, 224, device="cuda"),
                                    aug_params=None)
        _, b = stepper.forward_loss(tok[:4], facs=torch.zeros(32, device="cuda"), noise=torch.zeros(32, 3, 224, 224, device="cuda"),
                                    aug_params=None)
    ia, ib = a["indices"].reshape(B, -1)[:4], b["indices"].reshape(4, -1)
    assert (ia == ib).float().mean().item() > 0.999       # a code may flip where two distances tie within bf16 round-off
    assert (a["z"][:4] - b["z"]).abs().max().item() < 1e-5 * a["z"].abs().max().item() + 1e-6
    d = (a["xr"][:4] - b["xr"]).abs()                      # GroupNorm statistics are per image: only bf16 tile-order
    assert d.mean().item() < 5e-3 and d.max().item() < 0.1, (d.mean().item(), d.max().item())   # round-off (and a flipped code) remain
