"""GPU tests of the train-step branches and host surface that round 1 left unexercised:
l2 / tv regularisers (a13), noise conditioning with / without a bank and repeat > 1 (a17), input_loss, normalize_input,
clip_grad_norm, cosine schedule (a14), EMA (n2), `train()` -> checkpoint -> resume -> `load_model` -> `test()` (a16, n3).
The oracle (CPU fp32 restatement) is the checker; tolerances are the fp32-mode ones (1e-4 rel on the loss)."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import ops  # noqa: E402
from feed_forward_vqgan_clip_amd import vqgan as fvq  # noqa: E402
from feed_forward_vqgan_clip_amd.optim import CosineAnnealingLR, FusedAdam  # noqa: E402

F32 = torch.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY_VQ = dict(ch=64, ch_mult=(1, 1, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32, z_channels=64, out_ch=3,
               embed_dim=64, n_embed=128)
TINY_CLIP = dict(embed_dim=32, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8,
                 context_length=16, vocab_size=96, transformer_width=64, transformer_heads=1, transformer_layers=2)


def _relrms(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def _setup(seed=11, cdt=F32, **over):
    kw = dict(lr=1e-3, epochs=1, noise_dim=0, dim=64, depth=2, dropout=0, cutn=4, batch_size=4, repeat=1, nb_noise=None,
              diversity_coef=0, clip_model="ViT-B/32", clip_dim=32, clip_size=32, model_type="mlp_mixer", vq_image_size=12,
              augs=["R"])
    kw.update(over)
    cfg = fmain.Config(**kw)
    clip_sd, vq_sd = fclip.random_state_dict(TINY_CLIP, seed), fvq.random_state_dict(TINY_VQ, seed + 1)
    torch.manual_seed(seed)
    net = fmain.build_model(cfg, 64).cuda().prepare(cdt)
    vq, perceptor = fvq.VQGAN(vq_sd, TINY_VQ, cdt), fclip.CLIP(clip_sd, cdt)
    opt = FusedAdam(net.parameters(), lr=cfg.lr)
    tok = torch.zeros(4, 16, dtype=torch.long)
    g = torch.Generator().manual_seed(seed)
    for i, L in enumerate([3, 6, 9, 12]):
        tok[i, 0] = 94
        tok[i, 1:L] = torch.randint(1, 94, (L - 1,), generator=g)
        tok[i, L] = 95
    n = cfg.cutn * cfg.repeat * 4
    facs = torch.rand(n, generator=g) * 0.1
    noise = torch.randn(n, 3, 32, 32, generator=g)
    return cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise


def _oracle(net_sd, vq, vq_sd, clip_sd, tok, facs, noise, cfg, noise_vec=None, text_feats=None):
    from oracle import clip as oclip
    from oracle import mappers as omap
    from oracle import step as ostep
    osd = {k: v.clone().requires_grad_(True) for k, v in net_sd.items()}
    if text_feats is None:
        with torch.no_grad():
            text_feats = oclip.encode_text(clip_sd, tok, None).float()
    feats = text_feats.repeat(cfg.repeat, 1)

    def mapper(sd, f):
        x = f if noise_vec is None else torch.cat((f, noise_vec), dim=1)
        return omap.mixer_forward(sd, x, image_size=12, channels=64, depth=2)
    oloss, omid = ostep.train_step_loss(mapper, osd, vq_sd, clip_sd, tok, cutn=cfg.cutn, cut_size=32, z_min=vq.z_min,
                                        z_max=vq.z_max, facs=facs.view(-1, 1, 1, 1), noise=noise, vq_cfg=TINY_VQ,
                                        text_feats=feats)
    return oloss, omid, osd, mapper, feats


def _check_grads(net, osd, tol=3e-3):
    params = dict(net.named_parameters())
    gmax = max(v.grad.abs().max().item() for v in osd.values() if v.grad is not None)
    for k, v in osd.items():
        if v.grad is None:
            continue
        tiny = (params[k].grad.cpu() - v.grad).abs().max().item() < 1e-6 * gmax
        assert tiny or _relrms(params[k].grad, v.grad) < tol, k


# ----------------------------------------------------------------------------- kernels
def test_mean_sq_and_tv_kernels(cuda):
    g = torch.Generator().manual_seed(3)
    z = torch.randn(3, 5, 7, 6, generator=g).cuda().requires_grad_(True)
    zr = z.detach().double().cpu().requires_grad_(True)
    (ops.mean_sq(z) * 1.7).backward()
    ((zr ** 2).mean() * 1.7).backward()
    assert abs(ops.mean_sq(z).item() - (zr ** 2).mean().item()) < 1e-6
    assert _relrms(z.grad, zr.grad) < 1e-6
    x = torch.rand(2, 9, 11, 3, generator=g).cuda().requires_grad_(True)           # NHWC
    xr = x.detach().double().cpu().permute(0, 3, 1, 2).requires_grad_(True)         # NCHW reference (main.py:423-428)
    ref = 0.5 * ((xr[:, :, 1:, :] - xr[:, :, :-1, :]).abs().mean() + (xr[:, :, :, 1:] - xr[:, :, :, :-1]).abs().mean())
    tv = ops.tv_loss_nhwc(x)
    assert abs(tv.item() - ref.item()) < 1e-6
    (tv * 0.3).backward()
    (ref * 0.3).backward()
    assert _relrms(x.grad, xr.grad.permute(0, 2, 3, 1)) < 1e-5
    assert abs(fmain.tv_loss(x.detach().permute(0, 3, 1, 2)).item() - ref.item()) < 1e-6     # the reference-shaped entry point


# ----------------------------------------------------------------------------- regularisers in the step
def test_step_with_l2_and_tv_matches_oracle(cuda):
    from oracle import step as ostep
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _setup(l2_coef=0.05, tv_coef=0.5)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss, mid = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())
    opt.zero_grad()
    loss.backward()
    oloss, omid, osd, mapper, feats = _oracle(msd, vq, vq_sd, clip_sd, tok, facs, noise, cfg)
    z_pre = mapper(osd, feats)                                          # l2 is taken BEFORE the clamp (main.py:758-763)
    ol2, otv = (z_pre ** 2).mean(), ostep.tv_loss(omid["xr"])
    total = oloss + 0.05 * ol2 + 0.5 * otv                              # main.py:831
    total.backward()
    assert abs(mid["l2"].item() - ol2.item()) / ol2.item() < 1e-4
    assert abs(mid["tv"].item() - otv.item()) / otv.item() < 1e-4
    assert abs(mid["dists"].item() - oloss.item()) / oloss.item() < 1e-4
    assert abs(loss.item() - total.item()) / total.item() < 1e-4
    _check_grads(net, osd)


# ----------------------------------------------------------------------------- noise conditioning / repeat
@pytest.mark.parametrize("nb_noise,repeat", [(None, 1), (None, 2), (6, 2), (6, 1)])
def test_noise_conditioning_and_repeat_match_oracle(cuda, nb_noise, repeat):
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _setup(noise_dim=8, nb_noise=nb_noise, repeat=repeat)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    assert net.input_dim == 32 + 8
    msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss, mid = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())
    opt.zero_grad()
    loss.backward()
    nv = mid["noise_vec"].detach().cpu()
    assert tuple(nv.shape) == (4 * repeat, 8)
    assert tuple(mid["z"].shape) == (4 * repeat, 64, 12, 12) and tuple(mid["embed"].shape) == (cfg.cutn * repeat * 4, 32)
    if nb_noise:
        bank = stepper.NOISE.cpu()
        assert net.NOISE is stepper.NOISE and tuple(bank.shape) == (nb_noise, 8)
        for r in range(repeat):                                          # repeat-major: rows r*bs .. r*bs+bs-1 share bank row
            rows = nv[r * 4:(r + 1) * 4]
            assert (rows == rows[0]).all() and any(torch.equal(rows[0], b) for b in bank)
        if repeat == 2:
            assert not torch.equal(nv[0], nv[4])                          # distinct bank rows per repeat (shuffle without replacement)
    oloss, omid, osd, _, _ = _oracle(msd, vq, vq_sd, clip_sd, tok, facs, noise, cfg, noise_vec=nv)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) / oloss.item() < 1e-4
    _check_grads(net, osd)
    # pinned conditioning noise reproduces the step exactly
    loss2, _ = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(), noise_vec_in=nv.cuda())
    assert abs(loss2.item() - loss.item()) < 1e-6


# ----------------------------------------------------------------------------- (inp, out) feature pairs
def test_input_loss_and_normalize_input_match_oracle(cuda):
    from oracle import step as ostep
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _setup(input_loss=True, input_loss_coef=0.25,
                                                                            target_loss_coef=0.5, normalize_input=True)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    g = torch.Generator().manual_seed(2)
    inp, out = torch.randn(4, 32, generator=g), torch.randn(4, 32, generator=g)     # pre-computed features (main.py:733 else-branch)
    msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss, mid = stepper.forward_loss(inp.cuda(), out.cuda(), facs=facs.cuda(), noise=noise.cuda())
    opt.zero_grad()
    loss.backward()
    inp_n = torch.nn.functional.normalize(inp, dim=1)                               # main.py:734-735 (only inp is normalised)
    oloss_t, omid, osd, _, _ = _oracle(msd, vq, vq_sd, clip_sd, tok, facs, noise, cfg, text_feats=inp_n)
    total = 0.5 * ostep.spherical_loss(omid["embed"], out, cfg.cutn) + 0.25 * ostep.spherical_loss(omid["embed"], inp_n, cfg.cutn)
    total.backward()
    assert abs(loss.item() - total.item()) / total.item() < 1e-4
    _check_grads(net, osd)


# ----------------------------------------------------------------------------- clip_grad_norm / cosine / EMA
def test_clip_grad_norm_matches_torch(cuda):
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _setup(clip_grad_norm=0.05)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    ref = {k: torch.nn.Parameter(v.detach().clone()) for k, v in net.named_parameters()}
    ropt = torch.optim.Adam(ref.values(), lr=cfg.lr)
    for it in range(2):
        loss, _ = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())
        opt.zero_grad()
        loss.backward()
        for k, p in net.named_parameters():
            ref[k].grad = p.grad.detach().clone()
        total = opt.clip_grad_norm_(cfg.clip_grad_norm)
        rtotal = torch.nn.utils.clip_grad_norm_(ref.values(), cfg.clip_grad_norm)
        assert abs(total.item() - rtotal.item()) / rtotal.item() < 1e-5
        assert rtotal.item() > cfg.clip_grad_norm                       # the clip is active in this test
        opt.step()
        ropt.step()
    for k, p in net.named_parameters():
        assert (p.detach().cpu() - ref[k].detach().cpu()).abs().max().item() < 2e-6, k
    # TrainStep wires it in (main.py:833-834): one call with the config key set must not raise and must clip
    stepper(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())


def test_cosine_schedule_matches_torch_and_resumes(cuda):
    cfg, net, *_ = _setup()
    opt = FusedAdam(net.parameters(), lr=0.01)
    p = torch.nn.Parameter(torch.zeros(1))
    ropt = torch.optim.Adam([p], lr=0.01)
    rs = torch.optim.lr_scheduler.CosineAnnealingLR(ropt, T_max=20, eta_min=0)
    s = CosineAnnealingLR(opt, T_max=20, eta_min=0)
    lrs = []
    for _ in range(12):
        ropt.step()
        rs.step()
        s.step()
        lrs.append(opt.param_groups[0]["lr"])
        assert abs(opt.param_groups[0]["lr"] - ropt.param_groups[0]["lr"]) < 1e-9
    # resume at step 7 from an optimizer whose lr is already decayed (opt.th holds it): same continuation
    opt2 = FusedAdam(net.parameters(), lr=lrs[6])
    s2 = CosineAnnealingLR(opt2, T_max=20, eta_min=0, base_lrs=[0.01], last_epoch=7)
    assert abs(opt2.param_groups[0]["lr"] - lrs[6]) < 1e-12
    for k in range(7, 12):
        s2.step()
        assert abs(opt2.param_groups[0]["lr"] - lrs[k]) < 1e-12


def test_ema_matches_torch_ema_update_rule(cuda):
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _setup()
    opt.enable_ema(0.9)
    shadow = {k: v.detach().clone() for k, v in net.named_parameters()}           # torch_ema: shadow = copy of the params
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    for n in range(1, 4):
        stepper(tok.cuda(), facs=facs.cuda(), noise=noise.cuda())
        decay = min(0.9, (1 + n) / (10 + n))                                        # ExponentialMovingAverage.update()
        for k, p in net.named_parameters():
            shadow[k] -= (1 - decay) * (shadow[k] - p.detach())
    esd = opt.ema_state_dict()
    assert set(esd.keys()) == set(net.state_dict().keys())
    for k in shadow:
        assert (esd[k] - shadow[k]).abs().max().item() < 1e-6, k
        assert (esd[k] - dict(net.named_parameters())[k].detach()).abs().max().item() > 0     # it really lags the weights


# ----------------------------------------------------------------------------- train() -> resume -> load_model -> test()
def _write_cfg(folder, **over):
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = dict(lr=0.001, epochs=50, noise_dim=4, dim=64, depth=2, dropout=0, cutn=2, batch_size=2, repeat=1, nb_noise=5,
               diversity_coef=0, vqgan_config=os.path.join(root, "configs", "vqgan_imagenet_f16_16384.yaml"),
               vqgan_checkpoint="random:1234", clip_model="ViT-B/32", clip_model_path="random:1234", path="synthetic:16",
               folder=str(folder), log_interval=2, model_type="mlp_mixer", vq_image_size=16, compute_dtype="bf16",
               max_steps=3, use_ema=True, scheduler="cosine", clip_grad_norm=1.0, l2_coef=0.01, eval_path="synthetic:4:7")
    cfg.update(over)
    path = os.path.join(str(folder), "cfg.yaml")
    with open(path, "w") as f:
        yaml.safe_dump(cfg, f)
    return path


def test_train_checkpoint_resume_load_model_and_test_cli(cuda, tmp_path):
    from PIL import Image
    path = _write_cfg(tmp_path)
    fmain.train(path)                                                    # steps 0,1,2; logs + checkpoints at 0 and 2
    for f in ("checkpoint.th", "checkpoint_ema.th", "opt.th", "scalars.jsonl", "progress.png", "fixed_batch_progress.png",
              "progress_0000000002.png", "fixed_batch_progress_0000000002.png"):
        assert os.path.exists(tmp_path / f), f
    recs = [json.loads(ln) for ln in open(tmp_path / "scalars.jsonl")]
    assert [r["step"] for r in recs] == [0, 2]
    for r in recs:
        assert set(r) >= {"loss", "dists", "diversity", "l2", "tv", "avg_loss", "lr", "eval_dists", "eval_clip_score"}
        assert math.isfinite(r["loss"]) and r["l2"] > 0 and abs(r["loss"] - (r["dists"] + 0.01 * r["l2"])) < 1e-4
    assert recs[1]["lr"] < recs[0]["lr"]                                  # cosine schedule is live
    assert abs(recs[0]["avg_loss"] - (0.99 + 0.01 * recs[0]["loss"])) < 1e-5      # EMA from 1.0, every step (main.py:861)
    ck = torch.load(tmp_path / "checkpoint.th", weights_only=False)
    assert set(ck) == {"state_dict", "config", "step", "epoch"} and ck["step"] == 2
    assert Image.open(tmp_path / "progress.png").size == (2 * 258 + 2, 258 + 2)   # make_grid(nrow=bs, padding=2) of 256x256
    # resume: picks up at the checkpointed step with the optimizer state and runs on to max_steps
    path2 = _write_cfg(tmp_path, max_steps=6)
    fmain.train(path2)
    recs = [json.loads(ln) for ln in open(tmp_path / "scalars.jsonl")]
    assert [r["step"] for r in recs] == [0, 2, 2, 4]
    ck2 = torch.load(tmp_path / "checkpoint.th", weights_only=False)
    assert ck2["step"] == 4
    opt_sd = torch.load(tmp_path / "opt.th", weights_only=False)
    assert int(float(opt_sd["state"][0]["step"])) == 3 + 3               # 3 steps of the first run + steps 2,3,4 of the second
    # cosine resumed mid-schedule (T_max 6, base 1e-3; the record holds the lr AFTER step 4's scheduler.step(), i.e.
    # schedule position 5) — not a restart from the decayed lr stored in opt.th
    assert abs(recs[-1]["lr"] - 0.001 * (1 + math.cos(math.pi * 5 / 6)) / 2) < 1e-9
    # load_model round trip + forward-only generation (main.py:977-1061)
    net = fmain.load_model(str(tmp_path / "checkpoint.th"))
    assert net.config.dim == 64
    for k, v in ck2["state_dict"].items():
        assert torch.equal(net.state_dict()[k].cpu(), v.cpu()), k
    out = tmp_path / "gen.png"
    xr = fmain.test(str(tmp_path / "checkpoint_ema.th"), "synthetic:3:1", nb_repeats=2, out_path=str(out), seed=0)
    assert tuple(xr.shape) == (6, 3, 256, 256) and float(xr.min()) >= 0 and float(xr.max()) <= 1
    assert Image.open(out).size == (2 * 258 + 2, 3 * 258 + 2)
    assert fmain._cli(["test", str(tmp_path / "checkpoint.th"), "synthetic:2", "--out-path", str(tmp_path / "g2.png")]) == 0
    assert os.path.exists(tmp_path / "g2.png")
    # with a Net2Net prior between the text embedding and the mapper (main.py:1022-1023,1037-1040)
    from feed_forward_vqgan_clip_amd import prior as fprior
    clip_dim = fmain.clip_dim_size(net.config)[0]
    torch.save({"model": fprior.random_state_dict(clip_dim, clip_dim, 16, 32, 2, 2, seed=2), "step": 0, "input_size": clip_dim,
                "output_size": clip_dim, "config": {"model": {"embedding_dim": 16, "hidden_dim": 32, "hidden_depth": 2,
                                                              "n_flows": 2}}}, tmp_path / "prior.th")
    xp = fmain.test(str(tmp_path / "checkpoint.th"), "synthetic:3:1", out_path=str(tmp_path / "g3.png"), seed=0,
                    prior_path=str(tmp_path / "prior.th"))
    assert tuple(xp.shape) == (3, 3, 256, 256) and not torch.equal(xp, xr[:3])


# ----------------------------------------------------------------------------- the wider augmentation set
def test_step_with_wider_augmentation_set_matches_oracle(cuda):
    """augs: ['Ro','Re2','Ji2','Er2','Gn','Cc'] (main.py:166-198) run through the same fused resampling kernel; the drawn
    parameters are handed to the oracle's statement of that formula (kornia itself is unpinned, see augment.py)."""
    from feed_forward_vqgan_clip_amd import augment as A
    augs = ["Ro", "Re2", "Ji2", "Er2", "Gn", "Cc"]
    cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = _setup(augs=augs)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    assert stepper.make_cutouts.augs == tuple(augs)
    prm = A.draw_params(16, 32, augs=tuple(augs), generator=torch.Generator().manual_seed(8))
    assert prm["cj"][:, 0].max() == 1 and prm["gn"].max() == 1 and (prm["erase"][:, 2] > prm["erase"][:, 0]).any()
    msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    loss, mid = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(),
                                     aug_params={k: v.cuda() for k, v in prm.items()})
    opt.zero_grad()
    loss.backward()
    from oracle import mappers as omap
    from oracle import step as ostep
    osd = {k: v.clone().requires_grad_(True) for k, v in msd.items()}
    facs_eff = torch.sqrt(facs * facs + prm["gn"] ** 2)                 # 'Gn' merged with the U(0,.1)*N(0,1) term
    oloss, _ = ostep.train_step_loss(lambda sd, f: omap.mixer_forward(sd, f, image_size=12, channels=64, depth=2), osd, vq_sd,
                                     clip_sd, tok, cutn=4, cut_size=32, z_min=vq.z_min, z_max=vq.z_max,
                                     facs=facs_eff, noise=noise, vq_cfg=TINY_VQ, aug_params=prm)
    oloss.backward()
    assert abs(loss.item() - oloss.item()) / oloss.item() < 1e-4
    _check_grads(net, osd, tol=5e-3)
    # names outside main.py:166-198 fail loudly at construction
    with pytest.raises(NotImplementedError):
        fmain.MakeCutouts(32, 4, augs=["Xx"])


# ----------------------------------------------------------------------------- dropout in the mappers
def test_dropout_kernel_statistics_and_mask_reuse(cuda):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 20, generator=g).cuda()
    res = torch.randn(1 << 20, generator=g).cuda()
    p = 0.3
    y = K.dropout(x, p, 1234)
    keep = y != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 3e-3
    assert torch.allclose(y[keep], x[keep] / (1 - p), rtol=1e-6)
    assert torch.equal(K.dropout(x, p, 1234), y)                              # same seed -> same mask (what backward relies on)
    assert not torch.equal(K.dropout(x, p, 1235) != 0, keep)                  # another seed -> another mask
    yr = K.dropout(x, p, 1234, residual=res)
    assert torch.allclose(yr, y + res, atol=1e-6)
    x16 = x.half()
    y16 = K.dropout(x16, p, 1234)
    assert torch.equal(y16 != 0, keep | (x16 == 0)) or ((y16 != 0) == keep).float().mean() > 0.9999
    assert torch.equal(K.dropout(x, 0.0, 7), x)                               # p = 0 is the identity


@pytest.mark.parametrize("kind", ["mlp_mixer", "vitgan"])
def test_mapper_dropout_train_eval_and_gradient(cuda, kind):
    """dropout > 0 (mlp_mixer_pytorch.py:20-22, vitgan.py:34-41,133): active in train mode (the reference never calls
    .eval() while training), off in eval mode; the backward pass regenerates the forward masks — checked against a
    central finite difference of the (fp32, fixed-seed) forward."""
    from feed_forward_vqgan_clip_amd.mappers import Generator, Mixer
    torch.manual_seed(3)
    if kind == "mlp_mixer":
        mk = lambda p: Mixer(input_dim=16, image_size=4, channels=8, patch_size=1, dim=32, depth=2, dropout=p)  # noqa: E731
    else:
        mk = lambda p: Generator(initialize_size=1, out_channels=8, input_dim=16, dim=24, num_heads=6, blocks=2, dropout=p)  # noqa: E731
    net0 = mk(0.0)
    net = mk(0.25)
    net.load_state_dict(net0.state_dict())
    net0, net = net0.cuda().prepare(F32), net.cuda().prepare(F32)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(6, 16, generator=g).cuda()
    with torch.no_grad():
        y0 = net0(x)
        ya, yb = net(x), net(x)
        assert (ya - yb).abs().max() > 1e-4 and (ya - y0).abs().max() > 1e-4          # train mode: fresh masks each call
        net.eval()
        assert torch.allclose(net(x), y0, atol=1e-5)                                  # eval mode: dropout off
        net.train()
    gw = torch.randn(*y0.shape, generator=g).cuda()
    v = torch.randn(6, 16, generator=g).cuda()

    def f(xx):
        ops._DROP["n"] = 1000                                                        # same masks for every evaluation
        return net(xx)
    xg = x.clone().requires_grad_(True)
    net._ffvc_arena.zero_grad()
    (f(xg) * gw).sum().backward()
    eps = 1e-2
    with torch.no_grad():
        fd = ((f(x + eps * v) - f(x - eps * v)) * gw).sum().item() / (2 * eps)
    an = (xg.grad * v).sum().item()
    assert abs(fd - an) / (abs(fd) + 1e-6) < 2e-2, (fd, an)
    assert all(torch.isfinite(p.grad).all() and p.grad.abs().max() > 0 for p in net.parameters())


# ----------------------------------------------------------------------------- checkpoint compatibility (checkpoint_io)
_LEGACY_WRITER = '''
import sys, types, torch
sys.path.insert(0, sys.argv[2])
out = sys.argv[1]
# a pytorch-lightning-like VQGAN checkpoint: state_dict next to objects of a package the reader does not have
pl = types.ModuleType("pl_fake"); sys.modules["pl_fake"] = pl
class Callback:
    def __init__(self): self.best = 1.0
Callback.__module__ = "pl_fake"; pl.Callback = Callback
from feed_forward_vqgan_clip_amd import vqgan as fvq
cfg = dict(ch=64, ch_mult=(1, 1, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32, z_channels=64, out_ch=3, embed_dim=64, n_embed=128)
sd = fvq.random_state_dict(cfg, 5)
torch.save({"state_dict": sd, "callbacks": [Callback()]}, out + "/vq.ckpt")
gsd = {("quantize.embed.weight" if k == "quantize.embedding.weight" else k): v for k, v in sd.items()}
torch.save({"state_dict": gsd, "callbacks": [Callback()]}, out + "/gumbel.ckpt")
torch.save({"state_dict": {"first_stage_model." + k: v for k, v in sd.items()}, "callbacks": [Callback()]}, out + "/n2n.ckpt")
'''


def _vq_yaml(path, target, nested=False):
    import yaml
    params = dict(embed_dim=64, n_embed=128, ddconfig=dict(ch=64, ch_mult=[1, 1, 2], num_res_blocks=1, attn_resolutions=[8],
                                                            resolution=32, z_channels=64, out_ch=3))
    if nested:
        params = dict(first_stage_config=dict(target="taming.models.vqgan.VQModel", params=params))
    with open(path, "w") as f:
        yaml.safe_dump(dict(model=dict(target=target, params=params)), f)
    return str(path)


def test_load_vqgan_model_reads_lightning_gumbel_and_net2net_checkpoints(cuda, tmp_path):
    """reference main.py:84-103: the three targets, from pytorch-lightning files whose extra objects cannot be unpickled here."""
    import subprocess
    import sys
    subprocess.run([sys.executable, "-c", _LEGACY_WRITER, str(tmp_path), ROOT], check=True)
    with pytest.raises(Exception):
        torch.load(tmp_path / "vq.ckpt", map_location="cpu", weights_only=False)
    ref = fvq.VQGAN(fvq.random_state_dict(TINY_VQ, 5), TINY_VQ, F32)
    z = torch.randn(2, 2, 2, 64, device="cuda")
    want, widx = fvq.synth_nhwc(ref, z)
    for ck, target, nested in (("vq.ckpt", "taming.models.vqgan.VQModel", False), ("gumbel.ckpt", "taming.models.vqgan.GumbelVQ", False),
                               ("n2n.ckpt", "taming.models.cond_transformer.Net2NetTransformer", True)):
        m = fvq.load_vqgan_model(_vq_yaml(tmp_path / "c.yaml", target, nested), str(tmp_path / ck), F32)
        got, gidx = fvq.synth_nhwc(m, z)
        # same weights -> same codes; the decode agrees to the last bits (GroupNorm moments are accumulated with fp64 atomics
        # whose order differs from launch to launch, so bitwise equality of two decodes is not guaranteed)
        assert torch.equal(gidx, widx) and (got - want).abs().max().item() <= 1e-5 * want.abs().max().item(), target
    with pytest.raises(ValueError):
        fvq.load_vqgan_model(_vq_yaml(tmp_path / "c.yaml", "taming.models.other.Thing"), "random:1", F32)


def test_load_model_reads_a_legacy_pickled_module(cuda, tmp_path):
    """reference main.py:568-575, 1281-1289: `model.th` is a pickled module instance of classes this package does not have."""
    import subprocess
    import sys
    cfg = fmain.Config(noise_dim=0, dim=64, depth=2, dropout=0, clip_model="ViT-B/32", clip_dim=32, model_type="mlp_mixer",
                       vq_image_size=12)
    torch.manual_seed(3)
    net = fmain.build_model(cfg, 64)
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    torch.save(sd, tmp_path / "sd.th")
    writer = '''
import sys, types, torch
from torch import nn
sd = torch.load(sys.argv[1] + "/sd.th")
ref = types.ModuleType("mlp_mixer_pytorch_fake"); sys.modules["mlp_mixer_pytorch_fake"] = ref
class Holder(nn.Module):
    pass
Holder.__module__ = "mlp_mixer_pytorch_fake"; ref.Holder = Holder
class Cfg(dict):
    pass
Cfg.__module__ = "mlp_mixer_pytorch_fake"; ref.Cfg = Cfg
def build(prefix_tree):
    m = Holder()
    for name, sub in prefix_tree.items():
        if isinstance(sub, dict): m.add_module(name, build(sub))
        else: m.register_parameter(name, nn.Parameter(sub))
    return m
tree = {}
for k, v in sd.items():
    d = tree
    parts = k.split(".")
    for p in parts[:-1]: d = d.setdefault(p, {})
    d[parts[-1]] = v
root = build(tree)
root.config = Cfg(noise_dim=0, dim=64, depth=2, dropout=0, clip_model="ViT-B/32", clip_dim=32, model_type="mlp_mixer", vq_image_size=12)
torch.save(root, sys.argv[1] + "/model.th")
'''
    subprocess.run([sys.executable, "-c", writer, str(tmp_path)], check=True)
    got = fmain.load_model(str(tmp_path / "model.th"), F32, vq_channels=64)
    assert got.config.model_type == "mlp_mixer" and got.config.depth == 2
    for k, v in sd.items():
        assert torch.equal(got.state_dict()[k].cpu(), v), k


def test_adam_skips_non_finite_gradients_and_backs_the_loss_scale_off(cuda):
    """one overflowing f16 backward must not poison p / m / v / ema.  Elements whose scaled gradient is inf / NaN keep
    their state, the event is counted on the device, `check_overflow` (called where the host synchronises anyway) halves the scale."""
    cfg, net, vq, perceptor, opt, *_ = _setup(cdt=torch.float16)
    opt.loss_scale = 1024.0
    opt.enable_ema(0.9)
    a = opt.arena
    torch.manual_seed(0)
    a.grads.copy_(torch.randn_like(a.grads) * 1024.0)
    bad_idx = torch.tensor([0, 5, a.total // 2, a.total - 1], device="cuda")
    a.grads[bad_idx[0]] = float("inf")
    a.grads[bad_idx[1]] = float("-inf")
    a.grads[bad_idx[2]] = float("nan")
    a.grads[bad_idx[3]] = float("inf")
    p0, g0 = a.params.clone(), a.grads.clone()
    opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(a.params).all() and torch.isfinite(opt._m).all() and torch.isfinite(opt._v).all()
    assert torch.isfinite(opt._ema).all()
    assert torch.equal(a.params[bad_idx], p0[bad_idx]) and float(opt._m[bad_idx].abs().max()) == 0.0   # skipped elements
    ok = torch.ones(a.total, dtype=torch.bool, device="cuda")
    ok[bad_idx] = False
    ref = torch.nn.Parameter(p0.clone())
    ref.grad = torch.where(ok, g0 / 1024.0, torch.zeros_like(g0))
    torch.optim.Adam([ref], lr=cfg.lr).step()
    assert float((a.params[ok] - ref.detach()[ok]).abs().max()) < 1e-6                                  # everything else: plain Adam
    assert opt.check_overflow() >= 1 and opt.loss_scale == 512.0
    assert opt.check_overflow() == 0 and opt.loss_scale == 512.0
    # with global-norm clipping the coefficient is 0 x inf = NaN for every element: the whole step is skipped, nothing is poisoned
    a.grads.copy_(g0)
    p1, e1 = a.params.clone(), opt._ema.clone()
    assert not torch.equal(e1, p1)                    # (the EMA copy lags the parameters: a blend would be visible)
    opt.clip_grad_norm_(1.0)
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(a.params, p1) and torch.isfinite(opt._v).all()
    assert torch.equal(opt._ema, e1)                  # a skipped step leaves the EMA copy alone too
    assert opt.check_overflow() >= 1 and opt.loss_scale == 256.0
    # regrowth after an interval of clean steps
    opt.scale_growth_interval = 2
    a.grads.copy_(torch.randn_like(a.grads))
    opt.step()
    opt.step()
    assert opt.check_overflow() == 0 and opt.loss_scale == 512.0


def test_clock_sample_reports_a_plausible_engine_clock(cuda):
    """ffvc_clock_sample: per-XCD (shader clock, 100 MHz wall clock) pairs.  What is asserted is that the counters tick and their
    ratio is a positive, finite frequency — not a particular value: between the two samples this test leaves the GPU mostly idle, so
    the reading may be anywhere between the idle and the boost clock (bench.py samples around the timed steps: 2.2-2.3 GHz), and a
    box may route the 64 one-thread sampling blocks to a subset of the XCDs."""
    import math
    c0 = K.clock_sample()
    x = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        x = (x @ x).clamp(-1, 1)
    c1 = K.clock_sample()
    torch.cuda.synchronize()
    seen = (c0[:, 1] > 0) & (c1[:, 1] > 0)
    assert int(seen.sum()) >= 1, (c0.tolist(), c1.tolist())
    assert bool(((c1[:, 1] - c0[:, 1])[seen] > 0).all()), (c0.tolist(), c1.tolist())      # the 100 MHz wall clock never stops
    # the shader-clock counter of an XCD that idled (power-gated) before the first sample may restart: at least one XCD must have
    # ticked through, and only those enter the average
    assert int((((c1[:, 0] - c0[:, 0]) > 0) & seen).sum()) >= 1, (c0.tolist(), c1.tolist())
    mhz = K.effective_clock_mhz(c0, c1)
    assert math.isfinite(mhz) and 1.0 < mhz < 10000.0, (mhz, c0.tolist(), c1.tolist())


def _relmax(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


# ----------------------------------------------------------------------------- captured step (hipGraph)
@pytest.mark.parametrize("model_type", ["mlp_mixer", "vitgan"])
def test_captured_step_replays_the_eager_step(cuda, model_type):
    """TrainStep.enable_graph: the step behind the text tower recorded once into a hipGraph and replayed.  With the random draws
    switched off (augs ['R'], noise_fac 0) eager and replayed training must walk the same trajectory: same losses, same
    parameters after the same batches (up to the summation order of the few fp32 atomics), scheduler / Adam bias correction /
    EMA driven through the device-resident scalars."""
    over = dict(model_type=model_type, lr=2e-3)
    if model_type == "vitgan":
        over.update(dim=48, depth=2, num_heads=3, vq_image_size=16)
    toks = []
    for sd in range(5):
        toks.append(_setup(seed=20 + sd, **over)[7].cuda())
    runs = {}
    for mode in ("eager", "graph"):
        cfg, net, vq, perceptor, opt, *_ = _setup(**over)
        # Adam with its default eps turns the rounding noise of near-zero gradients (fp32 atomics order: 1e-5 relative between two
        # identical EAGER runs) into +-lr steps, and three such steps already move the loss by 2e-3 — two eager runs then differ as
        # much as eager and replay.  A large eps makes the update smooth in the gradient, so the comparison below is sharp.
        opt.param_groups[0]["eps"] = 0.1
        opt.enable_ema(0.9)
        sched = fmain.CosineAnnealingLR(opt, T_max=20)
        stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt, scheduler=sched)
        stepper.make_cutouts.noise_fac = 0
        losses = [float(stepper(toks[0])[0])]
        if mode == "graph":
            stepper.enable_graph(4, toks[1])          # the eager warm-up step on the capture stream consumes batch 1
            sched.step()                               # (enable_graph's warm-up runs the body only: scheduler tick by hand)
            losses.append(float(stepper._g_out[0]))
        else:
            losses.append(float(stepper(toks[1])[0]))
        for t in toks[2:]:
            loss, mid = stepper(t)
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert (stepper._graph is not None) == (mode == "graph")
        runs[mode] = (losses, {k: v.detach().clone() for k, v in net.state_dict().items()}, opt._ema.clone(), opt._step,
                      opt.param_groups[0]["lr"])
    (le, pe, ee, se, lre), (lg, pg, eg, sg, lrg) = runs["eager"], runs["graph"]
    assert se == sg == 5 and abs(lre - lrg) < 1e-12
    for a, b in zip(le[:1] + le[2:], lg[:1] + lg[2:]):
        assert abs(a - b) / abs(a) < 1e-4, (le, lg)
    # parameters: Adam turns the rounding noise of ill-conditioned gradients (the token-mix output bias is invisible behind the
    # next LayerNorm: its gradient is pure cancellation noise) into +-lr steps, so two EAGER runs already differ by percents in
    # those entries (fp32 atomics order); the trajectory of the loss above is the sharp check, the bucket-wide rms the coarse one
    flat = lambda d: torch.cat([v.flatten().double() for v in d.values()])      # noqa: E731
    assert float((flat(pg) - flat(pe)).pow(2).mean().sqrt() / flat(pe).pow(2).mean().sqrt()) < 5e-3
    assert float((eg.double() - ee.double()).pow(2).mean().sqrt() / ee.double().pow(2).mean().sqrt()) < 5e-3


def test_captured_step_with_the_default_random_draws(cuda):
    """Default augmentations + cutout noise inside the captured step: fresh draws per replay (augmentation parameters through the
    static device tensors, noise through torch's graph-safe Philox state), finite and decreasing."""
    cfg, net, vq, perceptor, opt, _, _, tok, _, _ = _setup(augs=None, lr=3e-3)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    stepper(tok.cuda())
    stepper.enable_graph(4, tok.cuda())
    seen, losses = set(), []
    for _ in range(12):
        loss, mid = stepper(tok.cuda())
        losses.append(float(loss))
        seen.add(float(mid["embed"].detach().float().sum()))
    assert all(torch.isfinite(torch.tensor(losses))) and len(seen) == 12          # a new augmentation / noise draw every replay
    assert sum(losses[-4:]) < sum(losses[:4])
