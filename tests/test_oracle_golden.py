"""Pins oracle/ (the CPU restatement) to golden vectors produced by the reference's own code
(tools/gen_golden.py, run in the build container where /root/reference is importable).
Tolerance: fp32 op-order differences only -> rtol 1e-4 / atol 1e-5 (grads 2e-4/2e-5)."""
import torch
import torch.nn as nn

from golden_util import assert_close, grads, load, t, unpack_sd
from oracle import clip as oclip
from oracle import mappers, step


def _leaf(sd):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}


def _check_mapper(npz_name, fn, **kw):
    z = load(npz_name)
    sd = _leaf(unpack_sd(z, "sd"))
    x = t(z["x"]).requires_grad_(True)
    y = fn(sd, x, **kw)
    assert_close(y, z["y"], what=npz_name + " y")
    (y * t(z["gw"])).sum().backward()
    assert_close(x.grad, z["dx"], 2e-4, 2e-5, npz_name + " dx")
    for k, g in grads(z, "grad").items():
        assert_close(sd[k].grad, g, 2e-4, 2e-5, f"{npz_name} grad {k}")


def test_mixer_matches_reference():
    _check_mapper("mixer.npz", mappers.mixer_forward, image_size=4, channels=8, depth=2)


def test_vitgan_matches_reference():
    _check_mapper("vitgan.npz", mappers.vitgan_forward, initialize_size=1, dim=12, blocks=2, num_heads=6,
                  out_channels=8)


def test_simple_vitgan_matches_reference():
    _check_mapper("simple_vitgan.npz", mappers.simple_vitgan_forward, size=4, dim=12, blocks=2, num_heads=6,
                  out_channels=8)


def test_clip_towers_match_reference():
    z = load("clip.npz")
    sd = unpack_sd(z, "sd")
    img = t(z["img"]).requires_grad_(True)
    e = oclip.encode_image(sd, img)
    assert_close(e, z["image_embed"], what="image_embed")
    (e * t(z["gw"])).sum().backward()
    assert_close(img.grad, z["dimg"], 2e-4, 2e-5, "dimg")
    et = oclip.encode_text(sd, t(z["tok"]), heads=2)
    assert_close(et, z["text_embed"], what="text_embed")


def test_glue_matches_reference():
    z = load("glue.npz")
    x = t(z["vq_x"]).requires_grad_(True)
    q = step.vector_quantize(x, t(z["vq_codebook"]))
    assert torch.equal(q.detach(), t(z["vq_out"]))                      # gather == one-hot GEMM exactly
    (q * t(z["vq_g"])).sum().backward()
    assert_close(x.grad, z["vq_dx"], what="vq dx")
    xc = t(z["clamp_x"]).requires_grad_(True)
    yc = step.clamp_with_grad(xc, -1.0, 1.5)
    assert torch.equal(yc.detach(), t(z["clamp_y"]))
    (yc * t(z["clamp_g"])).sum().backward()
    assert torch.equal(xc.grad, t(z["clamp_dx"]))
    b = torch.zeros(1, 6, requires_grad=True)
    r = step.replace_grad(t(z["rg_a"]), b)
    assert torch.equal(r.detach(), t(z["rg_out"]))
    (r * t(z["rg_g"])).sum().backward()
    assert_close(b.grad, z["rg_db"], what="replace_grad")
    assert_close(step.tv_loss(t(z["tv_x"])), z["tv"], what="tv")
    xi = t(z["cut_x"]).requires_grad_(True)
    co = step.make_cutouts(xi, cut_size=8, cutn=3, pool_size=8)
    assert_close(co, z["cut_out"], what="cutouts")
    (co * t(z["cut_g"])).sum().backward()
    assert_close(xi.grad, z["cut_dx"], what="cutouts dx")
    co2 = step.make_cutouts(xi.detach(), cut_size=8, cutn=2, pool_size=10)
    assert_close(co2, z["cut2_out"], what="cutouts resize")


def test_ministep_matches_reference():
    """Composed step: tokens -> text tower -> Mixer -> clamp -> VQ/STE -> decode -> cutouts -> image tower -> loss."""
    z = load("ministep.npz")
    clip_sd = unpack_sd(load("clip.npz"), "sd")
    msd = _leaf(unpack_sd(z, "mixer_sd"))
    dsd = unpack_sd(z, "dec_sd")
    dec = nn.Sequential(nn.Conv2d(8, 6, 3, padding=1), nn.SiLU(), nn.Upsample(scale_factor=4, mode="nearest"),
                        nn.Conv2d(6, 3, 3, padding=1)).requires_grad_(False)
    dec.load_state_dict(dsd)
    cb = t(z["codebook"])
    loss, mid = step.train_step_loss(
        lambda sd, f: mappers.mixer_forward(sd, f, image_size=4, channels=8, depth=2), msd,
        {"quantize.embedding.weight": cb}, clip_sd, t(z["tok"]), cutn=int(z["cutn"]), cut_size=int(z["cut_size"]),
        z_min=cb.min().item(), z_max=cb.max().item(), clip_heads=(None, 2), decode_fn=dec)
    assert_close(mid["z"], z["z"], what="z")
    assert_close(mid["xr"], z["xr"], what="xr")
    assert_close(mid["embed"], z["embed"], 2e-4, 2e-5, what="embed")
    assert_close(loss, z["loss"], 1e-5, 0, what="loss")
    loss.backward()
    for k, g in grads(z, "mixer_grad").items():
        assert_close(msd[k].grad, g, 5e-4, 1e-7, f"mixer grad {k}")
