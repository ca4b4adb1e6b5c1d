"""CPU checks of the augmentation parameter sampler and of the oracle's resampling reference."""
import math

import torch

from feed_forward_vqgan_clip_amd import augment as A
from oracle import step as ostep


def test_parameter_distributions():
    g = torch.Generator().manual_seed(0)
    N, S = 4000, 224
    prm = A.draw_params(N, S, generator=g)
    ident_a = torch.tensor([1.0, 0, 0, 0, 1.0, 0])
    frac_af = 1 - (prm["ainv"] - ident_a).abs().sum(1).eq(0).float().mean().item()
    assert abs(frac_af - 0.7) < 0.03                                   # p = 0.7
    ang = torch.rad2deg(torch.atan2(prm["ainv"][:, 1], prm["ainv"][:, 0]))
    assert ang.abs().max() <= 15.0 + 1e-3 and ang.abs().max() > 13.0   # degrees = 15
    # translate=0.1 is kornia's scalar form: (max_dx, max_dy) = (0, 0.1) -> no horizontal shift of the image centre
    Af = torch.linalg.inv(torch.cat([prm["ainv"].view(N, 2, 3).double(), torch.tensor([[[0.0, 0, 1]]], dtype=torch.float64).repeat(N, 1, 1)], 1))
    c = torch.tensor([(S - 1) / 2, (S - 1) / 2, 1.0], dtype=torch.float64)
    shift = (Af @ c)[:, :2] - c[:2]
    assert shift[:, 0].abs().max() < 1e-4 and 0.08 * S < shift[:, 1].abs().max() <= 0.1 * S * S / (S - 1) + 1e-6
    eye = torch.eye(3).reshape(9)
    frac_pe = 1 - (prm["pinv"] - eye).abs().sum(1).lt(1e-6).float().mean().item()
    assert (prm["cmat"] - eye).abs().max() == 0 and prm["coff"].abs().max() == 0      # the jitter travels as kornia parameters
    cj = prm["cj"]
    frac_ji = cj[:, 0].mean().item()
    assert abs(frac_pe - 0.7) < 0.03 and abs(frac_ji - 0.7) < 0.03
    assert (cj[:, 1] == 1).all() and (cj[:, 2] == 1).all()              # 'Ji': brightness / contrast untouched
    assert cj[:, 3].min() >= 0.9 and cj[:, 3].max() <= 1.1 and cj[:, 4].abs().max() <= 0.1 and cj[:, 4].abs().max() > 0.09
    code = int(cj[0, 5])
    assert sorted((code >> (2 * k)) & 3 for k in range(4)) == [0, 1, 2, 3] and (cj[:, 5] == cj[0, 5]).all()   # one order per batch
    e = prm["erase"]
    assert (e == e[0]).all()                                            # same_on_batch=True
    if e[0, 2] > e[0, 0]:
        area = float((e[0, 2] - e[0, 0]) * (e[0, 3] - e[0, 1])) / (S * S)
        assert 0.05 < area < 0.45


def test_perspective_moves_corners_inwards():
    g = torch.Generator().manual_seed(3)
    S = 64
    prm = A.draw_params(200, S, augs=("Pe",), generator=g, p=1.0)
    H = torch.linalg.inv(prm["pinv"].view(-1, 3, 3).double())
    corners = torch.tensor([[0.0, 0, 1], [S - 1.0, 0, 1], [S - 1.0, S - 1.0, 1], [0, S - 1.0, 1]], dtype=torch.float64)
    q = torch.einsum("nij,kj->nki", H, corners)
    q = q[..., :2] / q[..., 2:]
    assert (q >= -1e-6).all() and (q <= S - 1 + 1e-6).all()
    assert (q[:, 0] <= 0.35 * S + 1.0).all()                             # distortion_scale 0.7 -> at most 0.35 * size (+ the half
                                                                         # pixel of warp_perspective's align_corners=False sampling)


def test_reference_identity_and_gradient():
    S, B, cutn = 16, 2, 2
    g = torch.Generator().manual_seed(1)
    pooled = torch.rand(B, 3, S, S, generator=g, dtype=torch.float64, requires_grad=True)
    prm = A.draw_params(cutn * B, S, augs=(), generator=g)
    out = ostep.augment_reference(pooled, prm["pinv"].double(), prm["ainv"].double(), prm["cmat"].double(), prm["erase"], cutn)
    assert torch.allclose(out, pooled.repeat(cutn, 1, 1, 1))
    out.sum().backward()
    assert torch.allclose(pooled.grad, torch.full_like(pooled, float(cutn)))


def test_wider_augmentation_set_parameters():
    """Ro / Re / Re2 / Cr / Cc / Ji2 / Er2 / Gn (main.py:166-198) as parameters of the same fused resampling kernel."""
    g = torch.Generator().manual_seed(4)
    N, S = 3000, 64
    eye = torch.eye(3).reshape(9)
    prm = A.draw_params(N, S, augs=("Ro",), generator=g)
    Hi = prm["pinv"].view(N, 3, 3)
    on = (prm["pinv"] - eye).abs().sum(1) > 1e-6
    assert abs(on.float().mean().item() - 0.7) < 0.04
    ang = torch.rad2deg(torch.atan2(Hi[:, 1, 0], Hi[:, 0, 0]))[on]
    assert ang.abs().max() <= 15.01 and ang.abs().max() > 13
    c = torch.tensor([(S - 1) / 2, (S - 1) / 2, 1.0])
    assert ((Hi @ c)[:, :2] - c[:2]).abs().max() < 1e-3                   # rotation about the image centre
    assert (prm["ainv"] - torch.tensor([1.0, 0, 0, 0, 1.0, 0])).abs().max() == 0
    # resized crops: axis-aligned, inverse map sends the output corners INTO the source image, area fraction in range
    for name, lo in (("Re", 0.1), ("Re2", 0.9)):
        prm = A.draw_params(N, S, augs=(name,), generator=g)
        Hi = prm["pinv"].view(N, 3, 3).double()
        assert Hi[:, 0, 1].abs().max() < 1e-6 and Hi[:, 1, 0].abs().max() < 1e-6
        w, h = Hi[:, 0, 0] * (S - 1) + 1, Hi[:, 1, 1] * (S - 1) + 1         # crop size in source pixels
        frac = w * h / (S * S)
        assert frac.min() > lo * 0.85 and frac.max() <= 1.0 + 1e-6
        x0, y0 = Hi[:, 0, 2], Hi[:, 1, 2]
        assert x0.min() > -1e-6 and (x0 + w - 1).max() < S - 1 + 1e-3 and y0.min() > -1e-6
    # identities
    prm = A.draw_params(16, S, augs=("Cr", "Cc"), generator=g)
    assert (prm["pinv"] - eye).abs().max() < 1e-6 and prm["coff"].abs().max() == 0
    # Ji2: p = 0.5, brightness / contrast factors U(0.9, 1.1), saturation U(0.95, 1.05), hue U(-0.05, 0.05)
    prm = A.draw_params(N, S, augs=("Ji2",), generator=g)
    cj = prm["cj"]
    assert abs(cj[:, 0].mean().item() - 0.5) < 0.04
    for col, lo, hi in ((1, 0.9, 1.1), (2, 0.9, 1.1), (3, 0.95, 1.05), (4, -0.05, 0.05)):
        assert cj[:, col].min() >= lo - 1e-6 and cj[:, col].max() <= hi + 1e-6 and cj[:, col].max() - cj[:, col].min() > 0.9 * (hi - lo)
    # Er2: independent rectangles, p = 0.7;  Gn: p = 0.5
    prm = A.draw_params(N, S, augs=("Er2", "Gn"), generator=g)
    e = prm["erase"]
    has = (e[:, 2] > e[:, 0])
    assert abs(has.float().mean().item() - 0.7) < 0.04 and not (e[has] == e[has][0]).all()
    assert abs(prm["gn"].mean().item() - 0.5) < 0.04 and set(prm["gn"].unique().tolist()) == {0.0, 1.0}
    # order of composition: a border-padded 'Af' keeps its padding only as the first warp of a launch -> ('Ro', 'Af') is two launches,
    # as is any order that leaves geometry -> colour -> erase; 'Sh' / 'Et' / 'Ts' are their own kernels between fused launches
    import pytest
    with pytest.raises(NotImplementedError):
        A.draw_params(64, S, augs=("Ro", "Af"), generator=g, p=1.0)
    with pytest.raises(NotImplementedError):
        A.draw_params(4, S, augs=("Sh",))
    kinds = lambda augs, **kw: [k for k, _ in A.plan(A.draw_chain(8, S, augs, g), 8, S, **kw)]   # noqa: E731
    assert kinds(("Ro", "Af")) == ["fused", "fused"] and kinds(("Ji", "Af")) == ["fused", "fused"]
    # default = kornia's sequential resampling (one warp per launch); the composed single launch is the opt-in
    # default = kornia's sequential resampling: the affine as its own interpolation.  Where the affine launch is followed by a fused
    # launch on the same image size the two become ONE launch in the kernel's sequential form (`seq`: 16 source taps per pixel)
    seqs = lambda augs, **kw: [(k, int(p.get("seq", 0))) for k, p in A.plan(A.draw_chain(8, S, augs, g), 8, S, **kw)]   # noqa: E731
    assert seqs(A.DEFAULT) == [("fused", 1)] and seqs(A.DEFAULT, sequential=False) == [("fused", 0)]
    assert seqs(("Af", "Ro")) == [("fused", 1)] and seqs(("Ro", "Af")) == [("fused", 0), ("fused", 0)]   # a warp BEHIND the affine merges
    assert kinds(("Sh", "Af", "Et", "Ts", "Er")) == ["fused", "Sh", "fused", "Et", "Ts", "fused"]
    assert kinds(("Af", "Pe", "Sh"), sequential=False) == ["fused", "Sh", "fused"]
    assert seqs(("Af", "Pe", "Sh")) == [("fused", 1), ("Sh", 0), ("fused", 0)]
    assert seqs(("Af", "Re")) == [("fused", 1)]                             # kornia's RandomResizedCrop interpolates too (same size: merged)
    big = lambda augs, **kw: [(k, int(p.get("seq", 0))) for k, p in A.plan(A.draw_chain(8, S, augs, g, src_size=2 * S), 8, S, 2 * S, **kw)]   # noqa: E731
    assert big(("Af", "R")) == [("fused", 0), ("fused", 0)]                 # a resize that changes the image size: two launches
    assert big(("Af", "R"), sequential=False) == [("fused", 0)] and big(("Af", "Cc")) == [("fused", 0)]   # composed / exact integer crop


def test_fused_plan_matches_the_kornia_restatement_where_they_must_agree():
    """augment.plan()'s launches driven by the SAME raw draws as oracle/kornia_aug.apply_chain (the independent sequential
    restatement of kornia 0.5.10).  Where the fused form is exact they agree to rounding: one warp per launch (`sequential=True`)
    with jitter / erase, and the dense operators evaluated with torch's grid_sample.  (The one-launch default differs by its single
    interpolation for Af -> Pe; tools/augment_deviation.py measures that.)"""
    from oracle import kornia_aug as ka
    from oracle import step as ostep
    g = torch.Generator().manual_seed(11)
    N, S = 12, 24
    src = torch.rand(N, 3, S, S, generator=g, dtype=torch.float64)
    for augs in (("Af",), ("Pe",), ("Ro",), ("Ji",), ("Ji2",), ("Af", "Pe", "Ji", "Er"), ("Pe", "Ji2", "Er2")):
        chain = A.draw_chain(N, S, augs, g, p=0.8)
        want = ka.apply_chain(src, chain)
        x = src
        for kind, prm in A.plan(chain, N, S, sequential=True):
            assert kind == "fused"
            x = ostep.augment_reference(x, prm["pinv"].double(), prm["ainv"].double(), prm["cmat"].double(), prm["erase"], 1,
                                        coff=prm["coff"].double(), cj=prm.get("cj"), seq=bool(prm.get("seq", 0)))
        assert (x - want).abs().max() < 2e-5, (augs, (x - want).abs().max())


def test_reference_colour_offset_and_erase_per_sample():
    S, B, cutn = 16, 2, 2
    g = torch.Generator().manual_seed(2)
    pooled = torch.rand(B, 3, S, S, generator=g, dtype=torch.float64)
    prm = A.draw_params(cutn * B, S, augs=(), generator=g)
    prm["coff"][:] = torch.tensor([0.1, -0.2, 0.05])
    prm["erase"][1] = torch.tensor([2, 3, 7, 9], dtype=torch.int32)
    out = ostep.augment_reference(pooled, prm["pinv"].double(), prm["ainv"].double(), prm["cmat"].double(), prm["erase"], cutn,
                                  coff=prm["coff"].double())
    ref = pooled.repeat(cutn, 1, 1, 1) + torch.tensor([0.1, -0.2, 0.05], dtype=torch.float64).view(1, 3, 1, 1)
    ref[1, :, 3:9, 2:7] = 0
    assert torch.allclose(out, ref, atol=1e-6)
