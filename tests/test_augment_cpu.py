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
    eye = torch.eye(3).reshape(9)
    frac_pe = 1 - (prm["pinv"] - eye).abs().sum(1).lt(1e-6).float().mean().item()
    frac_ji = 1 - (prm["cmat"] - eye).abs().sum(1).lt(1e-6).float().mean().item()
    assert abs(frac_pe - 0.7) < 0.03 and abs(frac_ji - 0.7) < 0.03
    grey = prm["cmat"].view(N, 3, 3) @ torch.ones(3)                    # hue / saturation keep greys grey
    assert (grey - 1).abs().max() < 1e-3
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
    assert (q[:, 0] <= 0.35 * S + 1e-6).all()                            # distortion_scale 0.7 -> at most 0.35 * size


def test_reference_identity_and_gradient():
    S, B, cutn = 16, 2, 2
    g = torch.Generator().manual_seed(1)
    pooled = torch.rand(B, 3, S, S, generator=g, dtype=torch.float64, requires_grad=True)
    prm = A.draw_params(cutn * B, S, augs=(), generator=g)
    out = ostep.augment_reference(pooled, prm["pinv"].double(), prm["ainv"].double(), prm["cmat"].double(), prm["erase"], cutn)
    assert torch.allclose(out, pooled.repeat(cutn, 1, 1, 1))
    out.sum().backward()
    assert torch.allclose(pooled.grad, torch.full_like(pooled, float(cutn)))
