"""GPU parity of the assembled HIP modules (mapper, VQGAN decoder, CLIP towers, full train step)
against (a) the golden vectors produced by the reference's own code and (b) the oracle on seeded
inputs.  fp32 ("parity") mode must agree to fp32 round-off; bf16 mode to the stated stage tolerances.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import assert_close, grads, load, t, unpack_sd  # noqa: E402

from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from feed_forward_vqgan_clip_amd import ops, vqgan as fvq  # noqa: E402
from feed_forward_vqgan_clip_amd.mappers import Mixer  # noqa: E402
from feed_forward_vqgan_clip_amd.optim import FusedAdam  # noqa: E402

F32, BF16, F16 = torch.float32, torch.bfloat16, torch.float16
TINY_VQ = dict(ch=64, ch_mult=(1, 1, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32, z_channels=64,
               out_ch=3, embed_dim=64, n_embed=128)      # 3 levels -> image = 4 * S, attention at the first level


def _relrms(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()


# ----------------------------------------------------------------------------- mapper vs reference golden
def test_mixer_fp32_matches_reference_golden(cuda):
    z = load("mixer.npz")
    net = Mixer(input_dim=24, image_size=4, channels=8, patch_size=1, dim=16, depth=2)
    net.load_state_dict(unpack_sd(z, "sd"))
    net = net.cuda().prepare(F32)
    x = t(z["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert y.shape == (3, 8, 4, 4)
    assert_close(y.cpu(), z["y"], 1e-4, 1e-5, "mixer y")
    net._ffvc_arena.zero_grad()
    (y * t(z["gw"]).cuda()).sum().backward()
    assert_close(x.grad.cpu(), z["dx"], 2e-4, 2e-5, "mixer dx")
    sd = dict(net.named_parameters())
    for k, g in grads(z, "grad").items():
        assert_close(sd[k].grad.cpu(), g, 3e-4, 3e-5, f"mixer grad {k}")


@pytest.mark.parametrize("cdt,tol", [(BF16, 2e-2), (F16, 3e-3)])
def test_mixer_16bit_close_to_reference_golden(cuda, cdt, tol):
    z = load("mixer.npz")
    net = Mixer(input_dim=24, image_size=4, channels=8, patch_size=1, dim=16, depth=2)
    net.load_state_dict(unpack_sd(z, "sd"))
    net = net.cuda().prepare(cdt)
    x = t(z["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert _relrms(y, t(z["y"])) < tol             # 16-bit operands, fp32 accumulate / residual / LN
    net._ffvc_arena.zero_grad()
    (y * t(z["gw"]).cuda()).sum().backward()
    assert _relrms(x.grad, t(z["dx"])) < 3 * tol
    sd = dict(net.named_parameters())
    gs = grads(z, "grad")
    gmax = max(g.abs().max().item() for g in gs.values())
    for k, g in gs.items():
        if g.abs().max().item() > 1e-4 * gmax:
            assert _relrms(sd[k].grad, g) < 5 * tol, k


def test_mixer_state_dict_layout():
    net = Mixer(input_dim=24, image_size=4, channels=8, patch_size=1, dim=16, depth=2)
    ref = unpack_sd(load("mixer.npz"), "sd")
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.items()}


# ----------------------------------------------------------------------------- CLIP vs reference golden
@pytest.mark.parametrize("cdt,tol", [(F32, 2e-4), (BF16, 3e-2), (F16, 4e-3)])
def test_clip_matches_reference_golden(cuda, cdt, tol):
    z = load("clip.npz")
    sd = unpack_sd(z, "sd")
    model = fclip.CLIP(sd, cdt, text_heads=2)
    img = t(z["img"]).cuda().requires_grad_(True)
    e = model.encode_image(img)
    if cdt == F32:
        assert_close(e.cpu(), z["image_embed"], tol, 1e-5, "image_embed")
    else:
        assert _relrms(e, t(z["image_embed"])) < tol
    (e * t(z["gw"]).cuda()).sum().backward()
    if cdt == F32:
        assert_close(img.grad.cpu(), z["dimg"], 5e-4, 2e-5, "dimg")
    else:
        assert _relrms(img.grad, t(z["dimg"])) < 2 * tol
    et = model.encode_text(t(z["tok"]).cuda())          # text tower is always exact fp32
    assert_close(et.cpu(), z["text_embed"], 2e-4, 1e-5, "text_embed")


@pytest.mark.parametrize("cdt,tol", [(F32, 2e-4), (BF16, 3e-2), (F16, 4e-3)])
@pytest.mark.parametrize("quick", [True, False])
def test_clip_patch14_long_sequence_matches_oracle(cuda, cdt, tol, quick):
    """ViT-L/14-shaped image tower (patch 14, > 64 tokens -> the flash-style attention kernels in the 16-bit modes, 3*14*14
    = 588-wide patch rows that are not 16-byte multiples) and the erf-GELU MLP of the open_clip architectures
    (main.py:1323-1329), forward + image gradient against the oracle (cloob.py:219-255 math)."""
    from oracle import clip as oclip
    cfg = dict(embed_dim=48, image_resolution=126, vision_layers=2, vision_width=128, vision_patch_size=14,
               context_length=16, vocab_size=96, transformer_width=64, transformer_heads=1, transformer_layers=1)
    sd = fclip.random_state_dict(cfg, seed=21)
    g = torch.Generator().manual_seed(5)
    for k in list(sd):                       # non-trivial biases / LN affine so every epilogue term is exercised
        if k.endswith("bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    model = fclip.CLIP(sd, cdt, quick_gelu=quick)
    assert model.grid == 9 and len(model.vblocks) == 2
    img = torch.randn(3, 3, 126, 126, generator=g)
    gw = torch.randn(3, 48, generator=g)
    x = img.clone().requires_grad_(True)
    ref = oclip.encode_image(sd, x, quick_gelu=quick)
    (ref * gw).sum().backward()
    xi = img.cuda().requires_grad_(True)
    e = model.encode_image(xi)
    (e * gw.cuda()).sum().backward()
    assert _relrms(e, ref.detach()) < tol
    assert _relrms(xi.grad, x.grad) < 2 * tol
    tok = torch.zeros(2, 16, dtype=torch.long)
    tok[:, 0], tok[0, 1:4], tok[0, 4], tok[1, 1:9], tok[1, 9] = 94, torch.tensor([5, 6, 7]), 95, torch.arange(10, 18), 95
    assert _relrms(model.encode_text(tok.cuda()), oclip.encode_text(sd, tok, quick_gelu=quick)) < 2e-4


@pytest.mark.parametrize("cdt", [F16, BF16])
def test_clip_fp8_tower_close_to_16bit_tower(cuda, cdt):
    """cfg5's fp8 MFMA path: the image tower with its four per-block linears in e4m3 (gradients e5m2, per-tensor delayed
    scaling) stays within fp8's error budget of the SAME tower in 16-bit storage, forward and image gradient, and keeps doing
    so on a second call (delayed scales now come from the first call's amax)."""
    cfg = dict(embed_dim=64, image_resolution=112, vision_layers=3, vision_width=256, vision_patch_size=16,
               context_length=16, vocab_size=96, transformer_width=64, transformer_heads=1, transformer_layers=1)
    sd = fclip.random_state_dict(cfg, seed=31)
    g = torch.Generator().manual_seed(6)
    for k in list(sd):
        if k.endswith("bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.1
    ref_model = fclip.CLIP(sd, cdt, quick_gelu=False)
    f8_model = fclip.CLIP(sd, cdt, quick_gelu=False, fp8=True)
    assert all(b.c_fc.fp8 is not None and b.in_proj.fp8 is not None for b in f8_model.vblocks)
    gw = torch.randn(8, 64, generator=g).cuda()
    for it in range(2):
        img = torch.randn(8, 3, 112, 112, generator=g).cuda()
        a, b = img.clone().requires_grad_(True), img.clone().requires_grad_(True)
        ea, eb = ref_model.encode_image(a), f8_model.encode_image(b)
        (ea * gw).sum().backward()
        (eb * gw).sum().backward()
        assert _relrms(eb, ea.detach()) < 0.08, it
        assert _relrms(b.grad, a.grad) < 0.25, it
        cos = torch.nn.functional.cosine_similarity(ea.detach().float(), eb.detach().float(), dim=1)
        assert cos.min().item() > 0.995


@pytest.mark.parametrize("what", ["tower", "decoder"])
def test_fp8_producer_side_quantisation_changes_nothing(cuda, what, monkeypatch):
    """Round 4: LayerNorm / GroupNorm write the fp8 operand of the layer behind them (and GroupNorm's backward the e5m2 operand of
    the dgrad in front) instead of a 16-bit tensor plus a quantisation pass.  The bytes are the same by construction, so the model
    with the switch on must track the model with it off, step after step (the delayed scales evolve identically); what is left is
    the launch-order noise of the GroupNorm moment atomics."""
    g = torch.Generator().manual_seed(21)
    if what == "tower":
        cfg = dict(embed_dim=64, image_resolution=112, vision_layers=3, vision_width=256, vision_patch_size=16,
                   context_length=16, vocab_size=96, transformer_width=64, transformer_heads=1, transformer_layers=1)
        sd = fclip.random_state_dict(cfg, seed=33)
        make = lambda: fclip.CLIP(sd, F16, quick_gelu=False, fp8=True)                       # noqa: E731
        run = lambda m, x: m.encode_image(x)                                                 # noqa: E731
        inputs = [torch.randn(8, 3, 112, 112, generator=g).cuda() for _ in range(3)]
        gws = [torch.randn(8, 64, generator=g).cuda() for _ in range(3)]
        tol = 1e-6
    else:
        vsd = fvq.random_state_dict(fvq.F16_16384, seed=34)
        make = lambda: fvq.VQGAN(vsd, fvq.F16_16384, F16, fp8=True)                          # noqa: E731
        run = lambda m, x: m.decode_nhwc(x.to(F16))                                          # noqa: E731
        inputs = [torch.randn(4, 16, 16, 256, generator=g).cuda() for _ in range(3)]
        gws = [torch.randn(4, 256, 256, 3, generator=g).cuda() for _ in range(3)]
        tol = 5e-3
    results = {}
    for on in (False, True):
        monkeypatch.setattr(ops, "_F8_PRODUCER", on)
        model = make()
        outs = []
        for x, gw in zip(inputs, gws):
            xi = x.clone().requires_grad_(True)
            K.fp8_flush_updates()                     # what TrainStep does at the top of a step
            y = run(model, xi)
            (y.float() * gw).sum().backward()
            outs.append((y.detach().float(), xi.grad.float()))
        results[on] = outs
        del model
    # Forward: same bytes -> same result (the decoder's GroupNorm moments come from deterministic epilogue sums here).  Backward of the
    # decoder: two runs of the SAME path already differ by ~10 % rms (measured, tools/f8_debug.py): the GroupNorm-backward partial sums
    # meet in LDS atomics, and a 1-ulp change that flips an e5m2 rounding (2 mantissa bits) moves that element by 25 %, which the next
    # layer's quantiser amplifies again — the spread IS the e5m2 quantisation noise, so the bound is that noise level plus direction.
    # with the producer-side quantisation on, some 16-bit tensors are NOT written (`f8_only`).  Fill those buffers with
    # NaN (kernels.DEBUG_F8_NAN) and run again: a finite output and gradient prove that nothing reads them
    monkeypatch.setattr(ops, "_F8_PRODUCER", True)
    monkeypatch.setattr(K, "DEBUG_F8_NAN", True)
    model = make()
    for x, gw in zip(inputs, gws):
        xi = x.clone().requires_grad_(True)
        K.fp8_flush_updates()
        y = run(model, xi)
        (y.float() * gw).sum().backward()
        assert torch.isfinite(y).all() and torch.isfinite(xi.grad).all(), "an unwritten (fp8-only) buffer was read"
    del model
    monkeypatch.setattr(K, "DEBUG_F8_NAN", False)
    for it, ((ya, ga), (yb, gb)) in enumerate(zip(results[False], results[True])):
        assert torch.isfinite(yb).all() and torch.isfinite(gb).all()
        ey, eg = _relrms(yb, ya), _relrms(gb, ga)
        cos = torch.nn.functional.cosine_similarity(ga.flatten(), gb.flatten(), dim=0).item()
        print(what, it, "forward", ey, "gradient", eg, "cos", cos)
        assert ey <= tol, (what, it, ey)
        assert (eg <= 1e-6 if what == "tower" else (eg <= 0.25 and cos > 0.97)), (what, it, eg, cos)


def test_decoder_batch_parts_on_streams_change_nothing(cuda, monkeypatch):
    """FFVC_DEC_STREAMS (an experiment kept opt-in: measured slower): the decoder batch in parts on separate HIP streams, forward and
    backward, must give the single-launch result — GroupNorm statistics are per image, the parts only change the launch shapes."""
    g = torch.Generator().manual_seed(41)
    vsd = fvq.random_state_dict(TINY_VQ, seed=42)
    z = torch.randn(8, 6, 6, 64, generator=g).cuda()
    gw = torch.randn(8, 24, 24, 3, generator=g).cuda()
    res = {}
    for n, sizes in ((1, ""), (2, ""), (2, "3,5"), (4, "")):
        monkeypatch.setattr(fvq, "_DEC_STREAMS", n)
        monkeypatch.setattr(fvq, "_DEC_SIZES", sizes)
        model = fvq.VQGAN(vsd, TINY_VQ, F32)
        zi = z.clone().requires_grad_(True)
        y = model.decode_nhwc(zi)
        (y * gw).sum().backward()
        torch.cuda.synchronize()
        res[(n, sizes)] = (y.detach(), zi.grad.detach())
    ya, ga = res[(1, "")]
    for key, (yb, gb) in res.items():
        assert _relrms(yb, ya) < 1e-5 and _relrms(gb, ga) < 1e-5, key


def test_clip_arch_names():
    """main.py:1308-1333: OpenAI names and openclip/<arch>/<pretrained> spellings -> architecture + activation."""
    from feed_forward_vqgan_clip_amd import main as fmain
    assert fmain.clip_arch("ViT-B/32") == (fclip.VIT_B32, True)
    assert fmain.clip_arch("ViT-L/14") == (fclip.VIT_L14, True)
    assert fmain.clip_arch("openclip/ViT-B-32-quickgelu/laion400m_e32") == (fclip.VIT_B32, True)
    assert fmain.clip_arch("openclip/ViT-B-32/laion2b_e16") == (fclip.VIT_B32, False)
    assert fmain.clip_arch("openclip/ViT-L-14/laion2b_s32b_b82k") == (fclip.VIT_L14, False)
    with pytest.raises(ValueError):
        fmain.clip_arch("RN50")


# ----------------------------------------------------------------------------- VQGAN decoder vs oracle
@pytest.mark.parametrize("cdt,tol", [(F32, 1e-4), (BF16, 3e-2), (F16, 4e-3)])
def test_vqgan_decoder_matches_oracle(cuda, cdt, tol):
    from oracle import step as ostep
    sd = fvq.random_state_dict(TINY_VQ, seed=7)
    vq = fvq.VQGAN(sd, TINY_VQ, cdt)
    g = torch.Generator().manual_seed(3)
    zin = torch.randn(2, 64, 4, 4, generator=g)
    z = zin.cuda().requires_grad_(True)
    xr = fvq.synth(vq, z)
    assert xr.shape == (2, 3, 16, 16)
    zo = zin.clone().requires_grad_(True)
    xo = ostep.synth(sd, zo, TINY_VQ)
    gw = torch.randn(2, 3, 16, 16, generator=g)
    (xr * gw.cuda()).sum().backward()
    (xo * gw).sum().backward()
    if cdt == F32:
        assert_close(xr.cpu(), xo.detach(), tol, 1e-5, "xr")
        assert_close(z.grad.cpu(), zo.grad, 1e-3, 1e-5, "dz")
    else:
        assert _relrms(xr, xo.detach()) < tol
        assert _relrms(z.grad, zo.grad) < 3 * tol


def test_vq_and_glue_match_reference_golden(cuda):
    z = load("glue.npz")
    x = t(z["vq_x"]).cuda().requires_grad_(True)
    q = fvq.vector_quantize(x, t(z["vq_codebook"]).cuda())
    assert torch.equal(q.detach().cpu(), t(z["vq_out"]))
    (q * t(z["vq_g"]).cuda()).sum().backward()
    assert_close(x.grad.cpu(), z["vq_dx"], what="vq dx")
    xc = t(z["clamp_x"]).cuda().requires_grad_(True)
    yc = fmain.clamp_with_grad(xc, -1.0, 1.5)
    assert torch.equal(yc.detach().cpu(), t(z["clamp_y"]))
    (yc * t(z["clamp_g"]).cuda()).sum().backward()
    assert torch.equal(xc.grad.cpu(), t(z["clamp_dx"]))
    b = torch.zeros(1, 6, device="cuda", requires_grad=True)
    r = fmain.replace_grad(t(z["rg_a"]).cuda(), b)
    (r * t(z["rg_g"]).cuda()).sum().backward()
    assert_close(b.grad.cpu(), z["rg_db"], what="replace_grad")
    mc = fmain.MakeCutouts(cut_size=8, cutn=3, augs=["R"], pool=True, pool_size=8)
    mc.noise_fac = 0
    xi = t(z["cut_x"]).cuda().requires_grad_(True)
    co = mc(xi)
    assert_close(co.cpu(), z["cut_out"], what="cutouts")
    (co * t(z["cut_g"]).cuda()).sum().backward()
    assert_close(xi.grad.cpu(), z["cut_dx"], what="cutouts dx")


# ----------------------------------------------------------------------------- full train step vs oracle
def _tiny_step(cdt, seed=11):
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=64, depth=2, dropout=0, cutn=4, batch_size=4, repeat=1,
                       nb_noise=None, diversity_coef=0, clip_model="ViT-B/32", clip_dim=32, clip_size=32,
                       model_type="mlp_mixer", vq_image_size=12, augs=["R"])     # 48x48 image -> 32x32 cutouts
    clip_cfg = dict(embed_dim=32, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8,
                    context_length=16, vocab_size=96, transformer_width=64, transformer_heads=1, transformer_layers=2)
    clip_sd = fclip.random_state_dict(clip_cfg, seed)
    vq_sd = fvq.random_state_dict(TINY_VQ, seed + 1)
    torch.manual_seed(seed)
    net = fmain.build_model(cfg, 64).cuda().prepare(cdt)
    vq = fvq.VQGAN(vq_sd, TINY_VQ, cdt)
    perceptor = fclip.CLIP(clip_sd, cdt)
    opt = FusedAdam(net.parameters(), lr=cfg.lr)
    tok = torch.zeros(4, 16, dtype=torch.long)
    g = torch.Generator().manual_seed(seed)
    for i, L in enumerate([3, 6, 9, 12]):
        tok[i, 0] = 94
        tok[i, 1:L] = torch.randint(1, 94, (L - 1,), generator=g)
        tok[i, L] = 95
    facs = torch.rand(16, generator=g) * 0.1
    noise = torch.randn(16, 3, 32, 32, generator=g)
    return cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise


# 16-bit throughput modes: stage tolerances, asserted UNCONDITIONALLY.  The VQ argmin is a discontinuity of the reference
# itself (a code flips when the two nearest codes are closer than the mapper's rounding error), so the end-to-end
# comparison is made twice: free-running (codes may differ: loose bound, agreement rate asserted) and with the
# reference's codes handed to the decoder (`force_idx`), where the north-star tolerance on the loss applies.
# (bf16 `grad`: rounds 2-4 saw the worst tensor — a bias gradient of the tiny model — move between 0.11 and 0.15+ from run to run and
# round 4 widened the bound to 0.25.  Root cause, round 5 (tools/r5/grad_spread.py, profiles/r05_grad_spread.txt): NOT a gradient
# reduction — the decoded image itself differed between runs (3.7e-3 rel-rms in bf16) because the GroupNorm statistics kernels combined
# their row-threads through LDS float atomics (arrival order -> last bits of the fp32 sums -> 1-ulp flips that a deep 16-bit decoder
# amplifies to its rounding-noise level), and the (avg + max) pooling's argmax routed the gradient differently for near-ties: 6e-2
# run-to-run in d(xr), 4e-2 in every mapper gradient.  With the fixed-order combine (csrc/norm.hip) image, embedding, d(xr) and d(z)
# are bit-identical between runs and the worst gradient measures 0.1181 every time: the bound is back at 0.15.)
LOWP_TOL = {BF16: dict(z=3e-2, agree=0.97, xr=3e-2, loss_same_codes=2e-3, loss_free=2e-2, grad=1.5e-1),
            F16: dict(z=4e-3, agree=0.99, xr=4e-3, loss_same_codes=1e-4, loss_free=5e-3, grad=8e-2)}   # worst: bias gradients of the tiny model


@pytest.mark.parametrize("cdt", [F32, BF16, F16])
def test_train_step_matches_oracle(cuda, cdt):
    from oracle import mappers as omap
    from oracle import step as ostep
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _tiny_step(cdt)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss, mid = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())
    # oracle on identical weights / inputs
    osd = {k: v.clone().requires_grad_(True) for k, v in msd.items()}
    oloss, omid = ostep.train_step_loss(
        lambda sd, f: omap.mixer_forward(sd, f, image_size=12, channels=64, depth=2), osd, vq_sd, clip_sd, tok,
        cutn=4, cut_size=32, z_min=vq.z_min, z_max=vq.z_max, facs=facs.view(-1, 1, 1, 1), noise=noise, vq_cfg=TINY_VQ)
    oloss.backward()
    oidx = ostep.vq_indices(omid["z"].detach().movedim(1, 3), vq_sd["quantize.embedding.weight"])
    rel = abs(loss.item() - oloss.item()) / abs(oloss.item())
    agree = (mid["indices"].cpu().view(-1) == oidx.view(-1)).float().mean().item()
    print(f"[{cdt}] loss hip={loss.item():.7f} oracle={oloss.item():.7f} rel={rel:.2e} vq index agreement={agree:.4f}")
    params = dict(net.named_parameters())
    gmax = max(v.grad.abs().max().item() for v in osd.values())
    if cdt == F32:
        opt.zero_grad()
        loss.backward()
        assert_close(mid["z"].cpu(), omid["z"].detach(), 2e-4, 2e-5, "z")
        assert agree == 1.0
        assert_close(mid["xr"].permute(0, 3, 1, 2).cpu(), omid["xr"].detach(), 2e-4, 2e-5, "xr")
        assert_close(mid["embed"].cpu(), omid["embed"].detach(), 5e-4, 5e-5, "embed")
        assert rel < 1e-4                                   # north_star: CLIP loss within 1e-4 rel of the CPU reference
        for k, v in osd.items():
            # parameters whose exact gradient is 0 (token-mix bias in front of a LayerNorm) hold round-off only
            tiny = (params[k].grad.cpu() - v.grad).abs().max().item() < 1e-6 * gmax
            assert tiny or _relrms(params[k].grad, v.grad) < 2e-3, k
        return
    tol = LOWP_TOL[cdt]
    zc = mid["z"].detach().clamp(vq.z_min, vq.z_max)
    assert _relrms(zc, omid["z"].detach()) < tol["z"]
    assert agree >= tol["agree"]
    assert rel < tol["loss_free"]
    # same codes -> the remaining stages are comparable at the stated tolerance
    loss_sc, mid_sc = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(), force_idx=oidx.cuda())
    rel_sc = abs(loss_sc.item() - oloss.item()) / abs(oloss.item())
    xerr = _relrms(mid_sc["xr"].permute(0, 3, 1, 2), omid["xr"].detach())
    print(f"[{cdt}] same codes: loss rel={rel_sc:.2e} xr rel-rms={xerr:.2e}")
    assert xerr < tol["xr"]
    assert rel_sc < tol["loss_same_codes"]
    ls = 8192.0 if cdt == F16 else 1.0                      # loss-scaled backward in f16 (gradients below 6e-5 would go subnormal)
    opt.zero_grad()
    (loss_sc * ls).backward()
    errs = sorted(((_relrms(params[k].grad / ls, v.grad), k) for k, v in osd.items() if v.grad.abs().max().item() > 1e-4 * gmax),
                  reverse=True)
    print(f"[{cdt}] worst param-grad rel-rms (same codes): {errs[:3]}")
    assert errs[0][0] < tol["grad"]


@pytest.mark.parametrize("cdt", [F32, BF16, F16])
def test_train_step_default_augs_matches_oracle(cuda, cdt):
    """Same step with the reference's DEFAULT augmentation set (Af, Pe, Ji, Er; main.py:164-165): ONE set of raw kornia draws.
    The oracle applies them as the reference does — kornia's nn.Sequential, operator after operator (main.py:199,219;
    oracle/kornia_aug.apply_chain) — the HIP step through MakeCutouts' default plan (one resample per warp, round 5)."""
    from feed_forward_vqgan_clip_amd import augment as A
    from oracle import mappers as omap
    from oracle import step as ostep
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _tiny_step(cdt)
    cfg.augs = None
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    assert stepper.make_cutouts.augs == ("Af", "Pe", "Ji", "Er") and stepper.make_cutouts.sequential
    chain = A.draw_chain(16, 32, generator=torch.Generator().manual_seed(5))
    er = dict(chain)["Er"]                                   # a rectangle that is certainly there
    er["on"][:] = True
    er["xs"][:], er["ys"][:], er["widths"][:], er["heights"][:] = 3, 4, 12, 16
    segs = A.plan(chain, 16, 32, sequential=True)
    assert [k for k, _ in segs] == ["fused"] and segs[0][1]["seq"] == 1     # Af, then Pe + Ji + Er: two interpolations, one launch
    msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss, mid = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(), aug_params=A.to_device(segs, "cuda"))
    opt.zero_grad()
    loss.backward()
    osd = {k: v.clone().requires_grad_(True) for k, v in msd.items()}
    oloss, omid = ostep.train_step_loss(
        lambda sd, f: omap.mixer_forward(sd, f, image_size=12, channels=64, depth=2), osd, vq_sd, clip_sd, tok,
        cutn=4, cut_size=32, z_min=vq.z_min, z_max=vq.z_max, facs=facs, noise=noise, vq_cfg=TINY_VQ, aug_chain=chain)
    oloss.backward()
    rel = abs(loss.item() - oloss.item()) / abs(oloss.item())
    print(f"[{cdt}] default augs: loss hip={loss.item():.7f} oracle={oloss.item():.7f} rel={rel:.2e}")
    if cdt == F32:
        assert rel < 1e-4
        params = dict(net.named_parameters())
        gmax = max(v.grad.abs().max().item() for v in osd.values())
        for k, v in osd.items():
            tiny = (params[k].grad.cpu() - v.grad).abs().max().item() < 1e-6 * gmax
            assert tiny or _relrms(params[k].grad, v.grad) < 5e-3, k
    else:
        assert rel < LOWP_TOL[cdt]["loss_free"]
    # a step without explicit parameters draws its own
    loss2, _ = stepper.forward_loss(tok.cuda())
    assert torch.isfinite(loss2)


def test_optimizer_step_matches_torch_adam(cuda):
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _tiny_step(F32)
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in net.named_parameters()}
    ropt = torch.optim.Adam(ref.values(), lr=cfg.lr)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    for _ in range(2):
        loss, _ = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())
        opt.zero_grad()
        loss.backward()
        for k, p in net.named_parameters():
            ref[k].grad = p.grad.detach().clone()
        opt.step()
        ropt.step()
    for k, p in net.named_parameters():
        assert_close(p.detach().cpu(), ref[k].detach().cpu(), 1e-5, 1e-7, f"adam {k}")
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}


# ----------------------------------------------------------------------------- other mapper families
from feed_forward_vqgan_clip_amd.mappers import Generator, SimpleGenerator, XTransformer  # noqa: E402


def _check_mapper_golden(name, net, shape):
    z = load(name)
    net.load_state_dict(unpack_sd(z, "sd"))
    net = net.cuda().prepare(F32)
    x = t(z["x"]).cuda().requires_grad_(True)
    y = net(x)
    assert tuple(y.shape) == shape
    assert_close(y.cpu(), z["y"], 2e-4, 2e-5, name + " y")
    net._ffvc_arena.zero_grad()
    (y * t(z["gw"]).cuda()).sum().backward()
    assert_close(x.grad.cpu(), z["dx"], 5e-4, 5e-5, name + " dx")
    sd = dict(net.named_parameters())
    gmax = max(g.abs().max().item() for g in grads(z, "grad").values())
    for k, g in grads(z, "grad").items():
        assert_close(sd[k].grad.cpu(), g, 1e-3, 1e-5 * gmax + 1e-6, f"{name} grad {k}")


def test_vitgan_fp32_matches_reference_golden(cuda):
    _check_mapper_golden("vitgan.npz", Generator(initialize_size=1, out_channels=8, input_dim=24, dim=12, num_heads=6,
                                                 blocks=2), (3, 8, 8, 8))


def test_simple_vitgan_fp32_matches_reference_golden(cuda):
    _check_mapper_golden("simple_vitgan.npz", SimpleGenerator(size=4, dim=12, num_heads=6, blocks=2, out_channels=8,
                                                              input_dim=24), (3, 8, 4, 4))


@pytest.mark.parametrize("which", ["vitgan", "simple_vitgan"])
def test_vitgan_golden_without_the_one_launch_attention(cuda, which, monkeypatch):
    """The batched-GEMM attention path (what sequences too long for csrc/attn_tiny.hip's LDS panels take), with the zero-padded
    projection rows stripped / restored around it: same golden vectors, forward and every gradient."""
    monkeypatch.setenv("FFVC_ATTN_TINY", "0")
    if which == "vitgan":
        _check_mapper_golden("vitgan.npz", Generator(initialize_size=1, out_channels=8, input_dim=24, dim=12, num_heads=6,
                                                     blocks=2), (3, 8, 8, 8))
    else:
        _check_mapper_golden("simple_vitgan.npz", SimpleGenerator(size=4, dim=12, num_heads=6, blocks=2, out_channels=8,
                                                                  input_dim=24), (3, 8, 4, 4))


@pytest.mark.parametrize("kind", ["vitgan", "simple_vitgan", "xtransformer"])
@pytest.mark.parametrize("cdt", [F32, BF16, F16])
def test_other_mappers_match_oracle(cuda, kind, cdt):
    """Realistic head geometry (dim 120 / 6 heads -> dim_head 20; x-transformer 3 heads x 64) vs the oracle."""
    from oracle import mappers as omap
    torch.manual_seed(5)
    if kind == "vitgan":
        net = Generator(initialize_size=1, out_channels=16, input_dim=32, dim=120, num_heads=6, blocks=2)
        ofn = lambda sd, x: omap.vitgan_forward(sd, x, initialize_size=1, dim=120, blocks=2, num_heads=6, out_channels=16)  # noqa: E731
    elif kind == "simple_vitgan":
        net = SimpleGenerator(size=4, dim=120, num_heads=6, blocks=2, out_channels=16, input_dim=32)
        ofn = lambda sd, x: omap.simple_vitgan_forward(sd, x, size=4, dim=120, blocks=2, num_heads=6, out_channels=16)  # noqa: E731
    else:
        net = XTransformer(input_dim=32, image_size=4, channels=16, dim=64, depth=2, heads=3)
        ofn = lambda sd, x: omap.xtransformer_forward(sd, x, image_size=4, channels=16, dim=64, depth=2, heads=3)  # noqa: E731
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda().prepare(cdt)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 32, generator=g)
    y = net(x.cuda())
    yo = ofn(sd, x)
    gw = torch.randn(*yo.shape, generator=g)
    net._ffvc_arena.zero_grad()
    (y * gw.cuda()).sum().backward()
    (yo * gw).sum().backward()
    tol = {F32: 2e-4, BF16: 4e-2, F16: 5e-3}[cdt]
    assert _relrms(y, yo.detach()) < tol, "forward"
    params = dict(net.named_parameters())
    worst = max(_relrms(params[k].grad, v.grad) for k, v in sd.items() if v.grad.abs().max() > 1e-6)
    assert worst < {F32: 2e-3, BF16: 1.5e-1, F16: 2e-2}[cdt], f"worst param-grad rel-rms {worst}"   # 16-bit operands at dim_head 20


@pytest.mark.parametrize("add_input", [True, False])
def test_xtransformer_without_initial_proj_matches_oracle(cuda, add_input):
    """transformer.py:33-43: initial_proj=False feeds the raw input row to every position (add_input) or as an extra
    leading token that is dropped afterwards."""
    from oracle import mappers as omap
    torch.manual_seed(9)
    net = XTransformer(input_dim=32, image_size=4, channels=16, dim=64, depth=2, heads=3, initial_proj=False,
                       add_input=add_input)
    assert not hasattr(net, "proj") and net.transformer.project_in.weight.shape == (64, 32)
    assert net.transformer.pos_emb.emb.weight.shape[0] == 16 + (0 if add_input else 1)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    net = net.cuda().prepare(F32)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(5, 32, generator=g)
    y = net(x.cuda())
    yo = omap.xtransformer_forward(sd, x, image_size=4, channels=16, dim=64, depth=2, heads=3, initial_proj=False,
                                   add_input=add_input)
    assert tuple(y.shape) == (5, 16, 4, 4)
    gw = torch.randn(*yo.shape, generator=g)
    net._ffvc_arena.zero_grad()
    (y * gw.cuda()).sum().backward()
    (yo * gw).sum().backward()
    assert _relrms(y, yo.detach()) < 2e-4
    params = dict(net.named_parameters())
    worst = max(_relrms(params[k].grad, v.grad) for k, v in sd.items() if v.grad is not None and v.grad.abs().max() > 1e-6)
    assert worst < 2e-3, worst


def test_text_prefetch_matches_inline(cuda):
    """The side-stream text-tower prefetch hands the same features to the step as the inline encode."""
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _tiny_step(F32)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    t = tok.cuda()
    inline = stepper.features(t).clone()
    stepper.prefetch(t)
    assert stepper._prefetched is not None
    got = stepper.features(t)
    assert stepper._prefetched is None
    torch.cuda.synchronize()
    assert torch.equal(got, inline)
    other = t.clone()                       # a different tensor must not pick the stale prefetch up
    stepper.prefetch(t)
    assert torch.equal(stepper.features(other), inline) and stepper._prefetched is not None
    l1, _ = stepper.forward_loss(t, facs=facs.cuda(), noise=noise.cuda())
    stepper._prefetched = None
    l2, _ = stepper.forward_loss(t, facs=facs.cuda(), noise=noise.cuda())
    assert abs(l1.item() - l2.item()) < 1e-6


def test_split_precision_text_tower_is_fp32_grade():
    """CLIP(text_exact=False): the text tower's Linear layers as ONE f16 GEMM of depth 3K over [hi | lo | hi] x [hi | hi | lo]
    (ffvc_split3) — against the exact fp32 MFMA path and the CPU oracle, at the full ViT-B/32 text dimensions (cloob.py:525-538)."""
    from feed_forward_vqgan_clip_amd import clip as fclip
    from feed_forward_vqgan_clip_amd import kernels as K
    from oracle import clip as oclip
    cfg = dict(fclip.VIT_B32, vision_layers=1, transformer_layers=4)
    sd = fclip.random_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(3)
    for k in list(sd):                                                 # non-trivial biases / LayerNorm affine
        if k.startswith("transformer") and (k.endswith("bias") or "ln_" in k):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    tok = torch.zeros(6, 77, dtype=torch.long)
    for i, L in enumerate((3, 9, 20, 40, 60, 76)):
        tok[i, 0] = 49406
        tok[i, 1:L] = torch.randint(1, 49000, (L - 1,), generator=g)
        tok[i, L] = 49407
    exact = fclip.CLIP(sd, torch.float16, text_exact=True).encode_text(tok.cuda()).cpu()
    split = fclip.CLIP(sd, torch.float16, text_exact=False)
    assert not split.text_exact
    got = split.encode_text(tok.cuda()).cpu()
    want = oclip.encode_text(sd, tok)
    assert _relrms(got, exact) < 5e-6, _relrms(got, exact)        # measured 2.3e-6 (the 16-bit epilogue's fast QuickGELU included)
    assert _relrms(got, want) < 1e-5 and _relrms(exact, want) < 1e-5
    # the split itself: hi + lo reproduces fp32 to ~2^-21, layout [hi | lo | hi] / [hi | hi | lo]
    x = torch.randn(7, 64, generator=g).cuda() * 3
    a, w = K.split3(x), K.split3(x, weight_order=True)
    hi, lo = x.half(), (x - x.half().float()).half()
    assert torch.equal(a, torch.cat([hi, lo, hi], 1)) and torch.equal(w, torch.cat([hi, hi, lo], 1))
    assert float(((hi.float() + lo.float()) - x).abs().max() / x.abs().max()) < 1e-6


def test_split_precision_text_tower_handles_out_of_range_checkpoints():
    """f16 segments overflow above 65504.  (1) A checkpoint whose text-tower weights carry entries beyond f16's range
    (compensated by tiny LayerNorm gains so the function stays tame) still gives fp32-grade features: every weight gets a per-tensor
    power-of-two scale that leaves through the GEMM's alpha.  (2) A checkpoint that drives an ACTIVATION out of range is detected
    once at load (probe forward) and the tower falls back to the exact fp32 MFMA instead of producing NaN features."""
    from feed_forward_vqgan_clip_amd import clip as fclip
    cfg = dict(fclip.VIT_B32, vision_layers=1, transformer_layers=2)
    sd = fclip.random_state_dict(cfg, 4)
    big = dict(sd)
    for n in range(2):      # c_fc weight x 2^21 (entries up to ~3e5: beyond f16), ln_2 gain / 2^10, c_fc bias and c_proj weight rescaled to match
        k = f"transformer.resblocks.{n}"
        big[k + ".mlp.c_fc.weight"] = sd[k + ".mlp.c_fc.weight"] * 2.0 ** 21
        big[k + ".ln_2.weight"] = sd[k + ".ln_2.weight"] * 2.0 ** -10
        big[k + ".ln_2.bias"] = sd[k + ".ln_2.bias"] * 2.0 ** -10
        big[k + ".mlp.c_fc.bias"] = sd[k + ".mlp.c_fc.bias"] * 2.0 ** 11
        big[k + ".mlp.c_proj.weight"] = sd[k + ".mlp.c_proj.weight"] * 2.0 ** -11
    assert float(big["transformer.resblocks.0.mlp.c_fc.weight"].abs().max()) > 65504
    tok = torch.zeros(3, 77, dtype=torch.long)
    g = torch.Generator().manual_seed(5)
    for i, L in enumerate((5, 30, 76)):
        tok[i, 0] = 49406
        tok[i, 1:L] = torch.randint(1, 49000, (L - 1,), generator=g)
        tok[i, L] = 49407
    exact = fclip.CLIP(big, torch.float16, text_exact=True).encode_text(tok.cuda()).cpu()
    split = fclip.CLIP(big, torch.float16, text_exact=False)
    assert not split.text_exact
    got = split.encode_text(tok.cuda()).cpu()
    assert torch.isfinite(got).all() and _relrms(got, exact) < 2e-5, _relrms(got, exact)
    hot = dict(sd)                                               # activations: the hidden layer of block 0 blown up by 2^20
    hot["transformer.resblocks.0.mlp.c_fc.weight"] = sd["transformer.resblocks.0.mlp.c_fc.weight"] * 2.0 ** 22
    hot["transformer.resblocks.0.mlp.c_fc.bias"] = sd["transformer.resblocks.0.mlp.c_fc.bias"] * 2.0 ** 22
    hot["transformer.resblocks.0.mlp.c_proj.weight"] = sd["transformer.resblocks.0.mlp.c_proj.weight"] * 2.0 ** -22
    fb = fclip.CLIP(hot, torch.float16, text_exact=False)
    assert fb.text_exact                                          # the probe saw non-finite features -> exact path
    ref = fclip.CLIP(hot, torch.float16, text_exact=True).encode_text(tok.cuda()).cpu()
    assert torch.isfinite(ref).all() and _relrms(fb.encode_text(tok.cuda()).cpu(), ref) < 1e-6


# ----------------------------------------------------------------------------- grouped weight gradients (round 5)
def _mixer_grads(group, depth=10, B=32, seed=3, partial=None):
    """Gradients of a Mixer (dim 1024: the channel MLP's 1024 x 4096 weights qualify for grouping) for loss = sum(z * r)."""
    from feed_forward_vqgan_clip_amd import mappers as fmap
    from feed_forward_vqgan_clip_amd import ops
    old = ops._WGRAD_GROUP
    ops._WGRAD_GROUP = group
    try:
        torch.manual_seed(seed)
        net = fmap.Mixer(input_dim=64, image_size=4, channels=32, patch_size=1, dim=1024, depth=depth).cuda().prepare(F16)
        x = torch.randn(B, 64, generator=torch.Generator().manual_seed(seed + 1)).cuda()
        r = torch.randn(B, 32, 4, 4, generator=torch.Generator().manual_seed(seed + 2)).cuda()
        net._ffvc_arena.zero_grad()
        (net(x) * r).sum().backward()
        torch.cuda.synchronize()
        grouped = sum(1 for b in net._blocks for W in (b[4], b[5]) if W.group is not None)
        return {k: p.grad.detach().clone() for k, p in net.named_parameters()}, grouped
    finally:
        ops._WGRAD_GROUP = old


def test_grouped_weight_gradients_equal_the_per_layer_launches(cuda):
    """ops.WgradGroup: the channel-MLP weight gradients of 4 consecutive Mixer blocks in one launch (blocks 4-7; the first four stay
    on per-layer launches, mappers.Mixer._build_packs), the remaining pair in another (8-9), against one launch per layer.  Same
    products, different fp32 summation order."""
    g4, n4 = _mixer_grads(4)
    g0, n0 = _mixer_grads(0)
    assert n4 == 12 and n0 == 0
    for k in g0:
        assert torch.isfinite(g4[k]).all(), k
        assert _relrms(g4[k], g0[k]) < 2e-5, (k, _relrms(g4[k], g0[k]))


def _vitgan_grads(group, seed=7):
    """Full-width VitGAN generator (vitgan.py:221-260 at cfg3's geometry: dim 1024, 6 heads x 170, 16 tokens, 32 samples = 512 rows),
    9 blocks: gradients of loss = sum(out * r)."""
    from feed_forward_vqgan_clip_amd import ops
    old, old_env = ops._WGRAD_GROUP, os.environ.get("FFVC_VIT_WGRAD_GROUP")
    ops._WGRAD_GROUP = group
    os.environ["FFVC_VIT_WGRAD_GROUP"] = "8"            # mark the packs (off by default); ops._WGRAD_GROUP <= 1 still launches per layer
    try:
        torch.manual_seed(seed)
        net = Generator(initialize_size=2, out_channels=4, input_dim=64, dim=1024, num_heads=6, blocks=9).cuda().prepare(F16)
        x = torch.randn(32, 64, generator=torch.Generator().manual_seed(seed + 1)).cuda().requires_grad_(True)
        out = net(x)
        r = torch.randn(*out.shape, generator=torch.Generator().manual_seed(seed + 2)).cuda()
        net._ffvc_arena.zero_grad()
        (out * r).sum().backward()
        torch.cuda.synchronize()
        grouped = sum(1 for b in net._bp for W in (b[4], b[5]) if W.group is not None)
        g = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        g["__dx"] = x.grad.detach().clone()
        return g, grouped
    finally:
        ops._WGRAD_GROUP = old
        if old_env is None:
            os.environ.pop("FFVC_VIT_WGRAD_GROUP", None)
        else:
            os.environ["FFVC_VIT_WGRAD_GROUP"] = old_env


def test_vitgan_grouped_weight_gradients_equal_the_per_layer_launches(cuda):
    """The MLP weight gradients of 8 consecutive VitGAN blocks in one launch (opt-in: FFVC_VIT_WGRAD_GROUP=8; a ninth block stays alone)
    against one launch per layer: same products.  Tolerance: with these launches a few percent of backward passes come out with ONE
    sample's gradients changed at f16-rounding level (relrms up to 1e-4 in a weight gradient; tools/r6/vitgan_determinism_old.py,
    DESIGN.md section 5) — the reason the grouped form is off by default; anything structural would be orders of magnitude above."""
    g8, n8 = _vitgan_grads(4)
    g0, n0 = _vitgan_grads(0)
    assert n8 == 16 and n0 == 16          # marked either way; FFVC_WGRAD_GROUP <= 1 only stops ops._wgrad from deferring
    for k in g0:
        assert torch.isfinite(g8[k]).all(), k
        # (the one-element SLN scalars are sums of 512 x 1024 products through fp32 atomics: their own run-to-run spread is ~1e-3)
        assert _relrms(g8[k], g0[k]) < (5e-4 if g0[k].numel() > 1 else 1e-2), (k, _relrms(g8[k], g0[k]))


@pytest.mark.parametrize("which", ["vitgan", "simple_vitgan"])
def test_vitgan_golden_with_in_place_sln_gradients(cuda, which, monkeypatch):
    """FFVC_SLN_INPLACE=1 + FFVC_SLN_SHARE=1 (opt-in): ffvc_sln_bwd_acc2 writes the scalar SLN gradients straight to their bucket slots and
    keeps one running sum for the shared modulation input — same golden vectors, forward and every gradient."""
    from feed_forward_vqgan_clip_amd import mappers as fmap
    from feed_forward_vqgan_clip_amd import ops
    monkeypatch.setattr(ops, "_SLN_INPLACE", True)
    monkeypatch.setattr(fmap, "_SLN_SHARE", True)
    if which == "vitgan":
        _check_mapper_golden("vitgan.npz", Generator(initialize_size=1, out_channels=8, input_dim=24, dim=12, num_heads=6,
                                                     blocks=2), (3, 8, 8, 8))
    else:
        _check_mapper_golden("simple_vitgan.npz", SimpleGenerator(size=4, dim=12, num_heads=6, blocks=2, out_channels=8,
                                                                  input_dim=24), (3, 8, 4, 4))


def test_partly_filled_weight_gradient_group_is_flushed(cuda):
    """A group that does not fill (a backward pass that misses some of its layers, or direct calls outside autograd) goes out when
    the side stream is joined: consecutive members as a smaller grouped launch, a gap in the run as per-layer launches."""
    from feed_forward_vqgan_clip_amd import mappers as fmap
    from feed_forward_vqgan_clip_amd import ops
    torch.manual_seed(0)
    net = fmap.Mixer(input_dim=64, image_size=4, channels=32, patch_size=1, dim=1024, depth=4).cuda().prepare(F16)
    Ws = [b[4] for b in net._blocks]                 # fc1 of the four blocks: one group
    assert Ws[0].group is not None and Ws[0].group[0] is Ws[3].group[0]
    rows = 512
    for members in ((0, 1, 2), (0, 1, 3), (2,)):
        net._ffvc_arena.zero_grad()
        ops_in = {}
        for i in members:
            dy = _mk_like((rows, Ws[i].N), 40 + i)
            x = _mk_like((rows, Ws[i].K), 50 + i)
            ops_in[i] = (dy, x)
            ops._wgrad(dy, x, Ws[i], rows)
        assert len(ops._PENDING_GROUPS) == 1        # nothing launched yet: the group is not full
        ops.join_side_stream()
        torch.cuda.synchronize()
        assert not ops._PENDING_GROUPS
        for i in range(4):
            got = Ws[i].weight.grad
            if i in ops_in:
                ref = ops_in[i][0].double().T @ ops_in[i][1].double()
                assert _relrms(got, ref) < 2e-5, (members, i)
                assert _relrms(Ws[i].bias.grad, ops_in[i][0].double().sum(0)) < 2e-5
            else:
                assert float(got.abs().max()) == 0.0


def _mk_like(shape, seed):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * 0.5).to(F16).cuda()
