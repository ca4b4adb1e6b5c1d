"""GPU tests of the rows either side of the hot path: the Net2Net prior's sampling direction against the
oracle (parity unpinned upstream: net2net is absent, see oracle/prior.py), the text-feature cache and the (text, image)
pair-feature dataset that `train` consumes with `input_loss` (main.py:231-279, 733-737, 812-824), and `test(...,
prior_path=...)` (main.py:1022-1023,1037-1040)."""
import gzip
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import prior as fprior  # noqa: E402


def _relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("C,D,E,H,depth,flows,B", [(16, 24, 8, 32, 2, 3, 5), (512, 512, 64, 256, 2, 4, 7), (6, 4, 4, 8, 1, 1, 1)])
def test_prior_reverse_matches_oracle_and_inverts_forward(cuda, C, D, E, H, depth, flows, B):
    from oracle import prior as oprior
    sd = fprior.random_state_dict(C, D, E, H, depth, flows, seed=3)
    flow = fprior.ConditionalFlatCouplingFlow(sd, in_channels=C, conditioning_dim=D, n_flows=flows)
    g = torch.Generator().manual_seed(9)
    z, cond = torch.randn(B, C, generator=g), torch.randn(B, D, generator=g)
    x = flow.reverse(z, cond.view(B, D, 1, 1))
    assert tuple(x.shape) == (B, C, 1, 1)
    ref = oprior.reverse(sd, z, cond, flows)
    assert _relmax(x.view(B, C), ref) < 2e-4          # exact-fp32 MFMA vs CPU fp32: summation order only
    zz, logdet = flow.forward(x, cond)
    assert _relmax(zz.view(B, C), z) < 1e-3 and tuple(logdet.shape) == (B,) and torch.isfinite(logdet).all()
    s = flow.sample(cond.view(B, D, 1, 1), generator=torch.Generator().manual_seed(1))
    s2 = flow.sample(cond.view(B, D, 1, 1), generator=torch.Generator().manual_seed(1))
    assert torch.equal(s, s2) and tuple(s.shape) == (B, C, 1, 1)


def test_prior_checkpoint_layout_roundtrip(cuda, tmp_path):
    """The dict `train_prior` writes (main.py:1423-1431) loads through load_prior_model (main.py:1447-1451)."""
    sd = fprior.random_state_dict(8, 8, 4, 16, 2, 2, seed=1)
    cfg = {"model": {"embedding_dim": 4, "hidden_dim": 16, "hidden_depth": 2, "n_flows": 2}}
    p = tmp_path / "prior.th"
    torch.save({"model": sd, "step": 7, "input_size": 8, "output_size": 8, "config": cfg}, p)
    flow = fprior.load_prior_model(str(p))
    assert flow.n_flows == 2 and flow.in_channels == 8
    assert tuple(flow.sample(torch.randn(3, 8, 1, 1)).shape) == (3, 8, 1, 1)


def _vocab(tmp_path):
    p = tmp_path / "bpe.txt.gz"
    with gzip.open(p, "wt", encoding="utf-8") as f:
        f.write("\n".join(["#version: test", "c a", "ca t</w>", "d o", "do g</w>", "a t</w>"]) + "\n")
    return str(p)


def test_text_feature_cache_feeds_the_feature_branch(cuda, tmp_path):
    """encode_text writes what encode_text returns; a TrainStep fed those rows (not torch.long -> used as features,
    main.py:733) reproduces the loss of the same step fed the tokens."""
    toks = fmain.synthetic_tokens(6, seed=4)
    tp = tmp_path / "tok.pkl"
    torch.save(toks, tp)
    out = tmp_path / "feat.pkl"
    feats = fmain.encode_text(str(tp), out=str(out), clip_path="random:5", batch_size=4)
    assert feats.dtype == torch.float32 and tuple(feats.shape) == (6, 512)
    loaded = fmain.load_dataset(str(out))
    assert torch.equal(loaded, feats)
    perceptor = fmain.load_clip_model("ViT-B/32", path="random:5", cdt=torch.float32)
    assert _relmax(perceptor.encode_text(toks.cuda()), feats) < 1e-5
    with pytest.raises(TypeError):
        fmain.encode_text(str(out), out=str(tmp_path / "again.pkl"), clip_path="random:5")


def test_encode_text_and_images_pairs(cuda, tmp_path):
    """main.py:231-279 on a folder of (caption, image) files: features equal the towers' outputs on the same inputs, in
    sorted file order, and the saved tuple is what load_dataset hands to train (inputs, targets)."""
    import numpy as np
    from PIL import Image
    vocab = _vocab(tmp_path)
    folder = tmp_path / "pairs"
    folder.mkdir()
    rng = np.random.RandomState(0)
    caps = {"a": "cat", "b": "dog cat", "c": "a dog"}
    for name, cap in caps.items():
        (folder / f"{name}.txt").write_text(cap)
        Image.fromarray(rng.randint(0, 255, (40 + 8 * len(cap), 56, 3), dtype=np.uint8)).save(folder / f"{name}.png")
    out = tmp_path / "features.pkl"
    tf, imf = fmain.encode_text_and_images(str(folder), img_ext="png", out=str(out), clip_path="random:5", bpe_path=vocab,
                                           batch_size=2)
    assert tuple(tf.shape) == (3, 512) and tuple(imf.shape) == (3, 512)
    inp, tgt = fmain.load_dataset(str(out))
    assert torch.equal(inp, tf) and torch.equal(tgt, imf)
    perceptor = fmain.load_clip_model("ViT-B/32", path="random:5")
    from feed_forward_vqgan_clip_amd import tokenizer
    toks = tokenizer.tokenize([caps[k] for k in sorted(caps)], truncate=True, bpe_path=vocab)
    assert _relmax(perceptor.encode_text(toks.cuda()), tf) < 1e-5
    img = fmain.clip_preprocess(Image.open(folder / "b.png"), 224)
    assert tuple(img.shape) == (3, 224, 224)
    with torch.no_grad():
        e = perceptor.encode_image(img[None].cuda()).float().cpu()
    assert _relmax(e, imf[1:2]) < 2e-2                                    # bf16 tower, batch-size independent up to rounding
    with pytest.raises(FileNotFoundError):
        fmain.encode_text_and_images(str(tmp_path), out=str(out), clip_path="random:5", bpe_path=vocab)


# ----------------------------------------------------------------------------- MakeCutouts: the non-default branches
def _cut_oracle(x, mc, prm, facs, noise):
    """main.py:203-229 restated with the oracle's resampling formula: source (pooled or raw) -> chain -> noise -> interpolate.
    prm: the parameter dict of one fused launch, or augment.plan()'s list of fused launches."""
    import torch.nn.functional as F
    from oracle import step as ostep
    B, H = x.shape[0], x.shape[2]
    batch = (F.adaptive_avg_pool2d(x, mc.pool_size) + F.adaptive_max_pool2d(x, mc.pool_size)) / 2 if mc.pool else x
    size = mc.batch_size_px(H)
    segs = [("fused", prm)] if isinstance(prm, dict) else prm
    cutn = mc.cutn
    for i, (kind, q) in enumerate(segs):
        assert kind == "fused"
        last = i == len(segs) - 1
        batch = ostep.augment_reference(batch, q["pinv"], q["ainv"], q["cmat"], q["erase"], cutn, facs=facs if last else None,
                                        noise=noise if last else None, coff=q.get("coff"), out_size=size if last else q.get("out"),
                                        cj=q.get("cj"), seq=bool(q.get("seq", 0)))
        cutn = 1
    if mc.interpolate:
        batch = F.adaptive_avg_pool2d(batch, mc.interp_size)
    return batch


def _to_cuda(prm):
    up = lambda d: {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in d.items()}    # noqa: E731
    return up(prm) if isinstance(prm, dict) else [(k, up(d)) for k, d in prm]


@pytest.mark.parametrize("kw", [
    dict(augs=["R", "Af", "Ji"], pool=False),                                   # raw 48x48 image, bilinear resize to 32
    dict(augs=["Af", "Pe", "Cr"], pool=True, pool_size=40),                     # pooled to 40, random 32-crop at the end
    dict(augs=["Cc", "Ro", "Er2"], pool=False),                                 # centre crop of the raw image
    dict(augs=["Re", "Ji2"], pool=True, pool_size=48),                          # resized crop from a larger pooled image
    dict(augs=["Af", "Pe", "Ji", "Er"], pool=True, interpolate=True, interp_size=16),   # default set, then avg-pool 32 -> 16
    dict(augs=["Af"], pool=True, pool_size=40, interpolate=True, interp_size=32),       # no resize in the chain: 40 -> 32 by pooling
    dict(augs=["R"], pool=True, pool_size=24),                                  # upsampling resize 24 -> 32
])
def test_makecutouts_branches_match_oracle(cuda, kw):
    B, H, cut, cutn = 2, 48, 32, 3
    mc = fmain.MakeCutouts(cut, cutn, **kw)
    mc.generator = torch.Generator().manual_seed(3)
    g = torch.Generator().manual_seed(4)
    x = torch.rand(B, 3, H, H, generator=g)
    n = cutn * B
    size = mc.batch_size_px(H)
    facs = torch.rand(n, generator=g) * 0.1
    noise = torch.randn(n, 3, size, size, generator=g)
    prm = mc.draw_aug_params(n, "cpu", H)
    assert prm is not None
    xo = x.clone().requires_grad_(True)
    ref = _cut_oracle(xo, mc, prm, facs, noise)
    S = mc.out_size_px(H)
    assert tuple(ref.shape) == (n, 3, S, S)
    gw = torch.randn(ref.shape, generator=g)
    (ref * gw).sum().backward()
    xh = x.cuda().requires_grad_(True)
    out = mc(xh, facs=facs.cuda(), noise=noise.cuda(), aug_params=_to_cuda(prm))
    assert tuple(out.shape) == tuple(ref.shape)
    assert _relmax(out, ref.detach()) < 1e-4
    (out * gw.cuda()).sum().backward()
    assert _relmax(xh.grad, xo.grad) < 1e-3


def test_resize_only_cutouts_equal_torch_interpolate(cuda):
    """augs=['R'], pool=False, noise off is literally `Resize(cut_size)(input.repeat(cutn,1,1,1))` (main.py:145-152,216-219):
    pins the 'R' map of the fused resampler to torch's own bilinear interpolate (align_corners=False)."""
    import torch.nn.functional as F
    B, H, cut, cutn = 2, 48, 32, 2
    mc = fmain.MakeCutouts(cut, cutn, augs=["R"], pool=False)
    mc.noise_fac = 0
    x = torch.rand(B, 3, H, H, generator=torch.Generator().manual_seed(8))
    ref = F.interpolate(x.repeat(cutn, 1, 1, 1), (cut, cut), mode="bilinear")
    out = mc(x.cuda())
    assert _relmax(out, ref) < 1e-5
    up = fmain.MakeCutouts(64, 1, augs=["R"], pool=False)
    up.noise_fac = 0
    assert _relmax(up(x.cuda()), F.interpolate(x, (64, 64), mode="bilinear")) < 1e-5


@pytest.mark.parametrize("augs", [["Sh"], ["Af", "Sh", "Pe"], ["Et"], ["Ts"], ["Af", "Pe", "Ji", "Er", "Sh", "Et", "Ts"], ["Ji2", "Ts", "Er2"],
                                  ["Af", "Pe", "Ji", "Er"], ["Af", "Ro", "Ji2"], ["Af", "Pe"]])
def test_sharpness_elastic_tps_match_the_kornia_restatement(cuda, augs):
    """'Sh' / 'Et' / 'Ts' (main.py:169,179,181) run as their own kernels between fused launches.  With one resample per warp
    (`sequential=True`, the default) the whole chain is, operator by operator, what kornia's nn.Sequential computes: checked against
    oracle/kornia_aug.apply_chain on the SAME raw draws, forward and gradient.  The last three lists are the ones whose two warps
    run as ONE launch in the kernel's sequential form (ffvc_augment_seq_fwd / _bwd: the default set among them)."""
    from feed_forward_vqgan_clip_amd import augment as A
    from oracle import kornia_aug as ka
    B, cut, cutn = 2, 32, 3
    n = cutn * B
    mc = fmain.MakeCutouts(cut, cutn, augs=augs, pool=True, sequential=True)
    mc.noise_fac = 0
    g = torch.Generator().manual_seed(21)
    x = torch.rand(B, 3, 64, 64, generator=g)
    chain = A.draw_chain(n, cut, tuple(augs), g, p=0.8)
    segs = A.plan(chain, n, cut, cut, sequential=True)
    if augs[0] == "Af" and augs[1] in ("Pe", "Ro"):
        assert segs[0][1].get("seq") == 1
    xo = x.double().requires_grad_(True)
    import torch.nn.functional as F
    pooled = (F.adaptive_avg_pool2d(xo, cut) + F.adaptive_max_pool2d(xo, cut)) / 2
    ref = ka.apply_chain(pooled.repeat(cutn, 1, 1, 1), chain)
    gw = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    (ref * gw).sum().backward()
    xh = x.cuda().requires_grad_(True)
    out = mc(xh, aug_params=_to_cuda(segs))
    assert tuple(out.shape) == tuple(ref.shape)
    err = (out.double().cpu() - ref.detach()).abs()
    assert (err > 2e-4).float().mean().item() < 2e-3, (err.max().item(), (err > 2e-4).float().mean().item())   # floor() ties in fp32
    (out * gw.float().cuda()).sum().backward()
    assert _relmax(xh.grad, xo.grad.float()) < 2e-2


def test_every_reference_augmentation_name_is_built(cuda):
    """main.py:166-198: the full name list constructs, draws and runs (default fused plan)."""
    for a in ("Ji2", "Ji", "Sh", "Gn", "Pe", "Ro", "Af", "Et", "Ts", "Cr", "Er", "Er2", "Re", "Re2", "Cc", "R"):
        mc = fmain.MakeCutouts(32, 2, augs=[a])
        out = mc(torch.rand(2, 3, 48, 48).cuda())
        assert tuple(out.shape) == (4, 3, 32, 32) and torch.isfinite(out).all()
    with pytest.raises(NotImplementedError):
        fmain.MakeCutouts(32, 2, augs=["Xx"])


def test_sequential_form_in_one_launch_equals_the_two_launches(cuda):
    """ffvc_augment_seq_fwd / _bwd (round 5): the affine as its own interpolation, evaluated lazily inside the launch that does the
    rest, against the same two resamples as two launches (affine alone -> fp32 image batch -> everything else).  Same values up to
    fp32 summation order, forward and gradient, with the colour jitter (non-linear: the backward re-evaluates the forward), an
    erase rectangle, noise and the patch layout."""
    from feed_forward_vqgan_clip_amd import augment as A
    from feed_forward_vqgan_clip_amd import ops
    B, S, cutn, P = 3, 64, 4, 16
    n = B * cutn
    g = torch.Generator().manual_seed(5)
    chain = A.draw_chain(n, S, A.DEFAULT, g, p=0.9)
    one = A.plan(chain, n, S, S, sequential=True)
    assert len(one) == 1 and one[0][1]["seq"] == 1
    # the un-merged plan: the same planner with the merge step switched off
    import feed_forward_vqgan_clip_amd.augment as Amod
    keep, Amod._merge_sequential = Amod._merge_sequential, (lambda segs: segs)
    try:
        two = A.plan(chain, n, S, S, sequential=True)
    finally:
        Amod._merge_sequential = keep
    assert len(two) == 2
    pooled = torch.rand(B, 3, S, S, generator=g)
    facs, noise = torch.rand(n, generator=g) * 0.1, torch.randn(n, 3, S, S, generator=g)
    gw = torch.randn(n, (S // P) ** 2, 3 * P * P, generator=g).cuda()
    mean, std = (0.48, 0.45, 0.40), (0.26, 0.26, 0.27)
    res = []
    for segs in (one, two):
        x = pooled.cuda().requires_grad_(True)
        cur, c = x, cutn
        for i, (_, prm) in enumerate(A.to_device(segs, "cuda")):
            if i == len(segs) - 1:
                out = ops.augment(cur, prm, c, P, mean, std, torch.float32, noise=noise.cuda(), facs=facs.cuda(), out_size=S)
            else:
                cur = ops.augment(cur, prm, c, S, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), torch.float32, out_size=S).view(n, 3, S, S)
                c = 1
        (out * gw).sum().backward()
        res.append((out.detach(), x.grad.detach()))
    assert _relmax(res[0][0], res[1][0]) < 1e-5
    assert _relmax(res[0][1], res[1][1]) < 1e-4
