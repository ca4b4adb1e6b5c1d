"""Oracle (CPU fp32) restatement of the train-step glue of main.py.  TEST INFRASTRUCTURE ONLY.

Autograd rules, VQ, synth, cutouts, loss and the composed step (main.py:715-837), written as
pure functions.  Randomness of the reference (noise `U(0,.1)*N(0,1)`, main.py:202,223-225) is
made explicit: callers pass `facs` and `noise` tensors.
"""
import torch
import torch.nn.functional as F

from . import clip as oclip
from . import vqgan as ovq

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # main.py:81
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)   # main.py:82


class _ReplaceGrad(torch.autograd.Function):       # main.py:105-116
    @staticmethod
    def forward(ctx, x_forward, x_backward):
        ctx.shape = x_backward.shape
        return x_forward

    @staticmethod
    def backward(ctx, g):
        return None, g.sum_to_size(ctx.shape)


class _ClampWithGrad(torch.autograd.Function):     # main.py:118-132
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.lo, ctx.hi = lo, hi
        ctx.save_for_backward(x)
        return x.clamp(lo, hi)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        # pass the gradient unless it would push an out-of-range value further out
        return g * (g * (x - x.clamp(ctx.lo, ctx.hi)) >= 0), None, None


replace_grad = _ReplaceGrad.apply
clamp_with_grad = _ClampWithGrad.apply


def vq_indices(x, codebook):
    """argmin_j ||x - c_j||^2 as written at main.py:135-136 (first index on ties)."""
    d = x.pow(2).sum(dim=-1, keepdim=True) + codebook.pow(2).sum(dim=1) - 2 * x @ codebook.T
    return d.argmin(-1)


def vector_quantize(x, codebook):
    """main.py:134-138: nearest code, straight-through gradient. x: (..., C)."""
    x_q = codebook[vq_indices(x, codebook)]        # == one_hot(idx) @ codebook (main.py:137)
    return replace_grad(x_q, x)


def synth(vq_sd, z, cfg=ovq.F16_16384, decode_fn=None):
    """main.py:140-143. z: (B,C,S,S) -> RGB (B,3,16S,16S) in [0,1]."""
    z_q = vector_quantize(z.movedim(1, 3), vq_sd["quantize.embedding.weight"]).movedim(3, 1)
    dec = decode_fn(z_q) if decode_fn is not None else ovq.decode(vq_sd, z_q, cfg)
    return clamp_with_grad(dec.add(1).div(2), 0, 1)


def tv_loss(y):                                    # main.py:423-428
    return 0.5 * ((y[:, :, 1:, :] - y[:, :, :-1, :]).abs().mean() + (y[:, :, :, 1:] - y[:, :, :, :-1]).abs().mean())


def make_cutouts(x, *, cut_size, cutn, facs=None, noise=None, pool=True, pool_size=None):
    """main.py:212-229 with augs=['R'] (Resize -> bilinear to cut_size, main.py:145-152,199-200).

    facs: (cutn*B,1,1,1) in [0, 0.1) and noise: like the output — the reference draws them
    unseeded (main.py:223-225); None disables the noise branch (noise_fac = 0).
    """
    pool_size = pool_size or cut_size
    if pool:
        c = (F.adaptive_avg_pool2d(x, pool_size) + F.adaptive_max_pool2d(x, pool_size)) / 2   # :217
        batch = c.repeat(cutn, 1, 1, 1)                                                        # :218
    else:
        batch = x.repeat(cutn, 1, 1, 1)
    batch = F.interpolate(batch, (cut_size, cut_size), mode="bilinear")                        # augs=['R']
    if facs is not None:
        batch = batch + facs * noise                                                           # :224-225
    return batch


def spherical_loss(embed, target_feats, cutn, coef=1.0):
    """main.py:801-811 for repeat=1: mean(2*asin(||H-E||/2)^2)."""
    H = F.normalize(target_feats.repeat(cutn, 1), dim=-1)
    E = F.normalize(embed, dim=1)
    return coef * H.sub(E).norm(dim=-1).div(2).arcsin().pow(2).mul(2).mean()


def train_step_loss(mapper_fn, mapper_sd, vq_sd, clip_sd, tokens, *, cutn, cut_size, z_min, z_max,
                    facs=None, noise=None, vq_cfg=ovq.F16_16384, clip_heads=(None, None),
                    pool_size=None, text_feats=None, decode_fn=None, aug_params=None, quick_gelu=True, aug_chain=None):
    """Forward half of one training step (main.py:729-811,831) with repeat=1, noise_dim=0,
    l2/tv/diversity coefficients 0.  Returns (loss, dict of intermediates).

    Augmentations: `aug_chain` = the raw per-operator draws ([(name, dict)], the layout of kornia_aug.sample_chain) applied
    as the reference applies them — `nn.Sequential(*augment_list)` (main.py:199,219), one kornia operator after the other on
    the previous one's output (kornia_aug.apply_chain).  This is the oracle for MakeCutouts' default path.  `aug_params`
    (composed single-resample parameters -> augment_reference below) states the formula of the opt-in FUSED launch only."""
    if text_feats is None:
        with torch.no_grad():
            text_feats = oclip.encode_text(clip_sd, tokens, clip_heads[1], quick_gelu).float()     # main.py:733,737
    z = mapper_fn(mapper_sd, text_feats).contiguous()                                   # :754-757
    z = clamp_with_grad(z, z_min, z_max)                                                # :763
    xr = synth(vq_sd, z, vq_cfg, decode_fn)                                             # :767
    if aug_chain is not None:                                                           # main.py:212-225 with default augs
        from . import kornia_aug
        pooled = (F.adaptive_avg_pool2d(xr, cut_size) + F.adaptive_max_pool2d(xr, cut_size)) / 2   # :213-215
        x = kornia_aug.apply_chain(pooled.repeat(cutn, 1, 1, 1), aug_chain)                       # :218-219
        if facs is not None:
            x = x + facs.view(-1, 1, 1, 1) * noise                                                 # :222-225
    elif aug_params is None:
        x = make_cutouts(xr, cut_size=cut_size, cutn=cutn, facs=facs, noise=noise, pool_size=pool_size)  # :796
    else:                                                                               # default augs, explicit parameters
        pooled = (F.adaptive_avg_pool2d(xr, cut_size) + F.adaptive_max_pool2d(xr, cut_size)) / 2   # :217
        x = augment_reference(pooled, aug_params["pinv"], aug_params["ainv"], aug_params["cmat"], aug_params["erase"], cutn,
                              facs, noise, coff=aug_params.get("coff"), cj=aug_params.get("cj"))
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype, device=x.device).view(1, -1, 1, 1)
    x = (x - mean) / std                                                                # :797
    embed = oclip.encode_image(clip_sd, x, clip_heads[0], quick_gelu).float()           # :799
    loss = spherical_loss(embed, text_feats, cutn)                                      # :801-811
    return loss, {"text_feats": text_feats, "z": z, "xr": xr, "embed": embed}


def adam_step(params, grads, state, lr, step, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (main.py:591) on lists of tensors, in place. `step` is 1-based."""
    b1, b2 = betas
    for p, g, (m, v) in zip(params, grads, state):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def augment_reference(pooled, pinv, ainv, cmat, erase, cutn, facs=None, noise=None, coff=None, out_size=None, cj=None, seq=False):
    """Plain-PyTorch statement of ffvc_augment_fwd (the fused Af -> Pe -> Ji -> Er chain of main.py:164-198 with
    explicit per-cutout parameters): returns (cutn*B, 3, S, S) BEFORE mean/std normalisation.  This pins the HIP kernel to
    its documented single-resample formula; kornia's own sequential form is restated in oracle/kornia_aug.py, and
    tools/augment_deviation.py measures one against the other.  cj (N,8): ColorJitter parameters as augment.plan() lays them
    out ([on, brightness, contrast, saturation, hue, order code]), applied with oracle/kornia_aug.color_jitter."""
    if seq:
        # the kernel's sequential form (ffvc_augment_seq_fwd): the affine slot as its own resample (an intermediate image at the
        # integer pixels), then everything else sampling that image with an identity affine
        N0 = cutn * pooled.shape[0]
        eye9 = torch.eye(3, dtype=pinv.dtype).reshape(1, 9).repeat(N0, 1)
        ident6 = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=ainv.dtype).repeat(N0, 1)
        inter = augment_reference(pooled, eye9, ainv, eye9.to(cmat.dtype), torch.zeros(N0, 4, dtype=erase.dtype), cutn)
        return augment_reference(inter, pinv, ident6, cmat, erase, 1, facs, noise, coff, out_size, cj)
    B, _, S, _ = pooled.shape                      # S: side of the source frame; So: side of the cutouts (a resize / crop in
    So = out_size or S                             # the chain is part of pinv)
    N = cutn * B
    ys, xs = torch.meshgrid(torch.arange(So, dtype=pooled.dtype), torch.arange(So, dtype=pooled.dtype), indexing="ij")
    x2, y2 = xs[None].expand(N, So, So), ys[None].expand(N, So, So)
    P, A = pinv.view(N, 9).to(pooled.dtype), ainv.view(N, 6).to(pooled.dtype)
    pe = lambda i: P[:, i].view(N, 1, 1)  # noqa: E731
    ae = lambda i: A[:, i].view(N, 1, 1)  # noqa: E731
    w = pe(6) * x2 + pe(7) * y2 + pe(8)
    w = torch.where(w.abs() > 1e-8, w, torch.full_like(w, 1e-8))
    x1 = (pe(0) * x2 + pe(1) * y2 + pe(2)) / w
    y1 = (pe(3) * x2 + pe(4) * y2 + pe(5)) / w
    # zero padding as grid_sample applies it: every tap outside the frame is zero -> linear fade over one pixel
    m = ((x1 + 1).clamp(0, 1) * (S - x1).clamp(0, 1)) * ((y1 + 1).clamp(0, 1) * (S - y1).clamp(0, 1))
    # a homography slot that only scales / shifts (resize, crop, identity) holds no zero-padded warp: coordinates are clamped only
    zp = ((P[:, 1] != 0) | (P[:, 3] != 0) | (P[:, 6] != 0) | (P[:, 7] != 0)).view(N, 1, 1)
    m = torch.where(zp, m, torch.ones_like(m))
    x0 = (ae(0) * x1 + ae(1) * y1 + ae(2)).clamp(0, S - 1)
    y0 = (ae(3) * x1 + ae(4) * y1 + ae(5)).clamp(0, S - 1)
    xi = x0.floor().clamp(max=max(S - 2, 0)).long()
    yi = y0.floor().clamp(max=max(S - 2, 0)).long()
    wx, wy = x0 - xi, y0 - yi
    src = pooled.repeat(cutn, 1, 1, 1).reshape(N, 3, S * S)
    def tap(dy, dx):
        idx = ((yi + dy) * S + (xi + dx)).view(N, 1, So * So).expand(N, 3, So * So)
        return src.gather(2, idx).view(N, 3, So, So)
    val = (1 - wy)[:, None] * ((1 - wx)[:, None] * tap(0, 0) + wx[:, None] * tap(0, 1)) + \
        wy[:, None] * ((1 - wx)[:, None] * tap(1, 0) + wx[:, None] * tap(1, 1))
    val = val * m[:, None]
    out = torch.einsum("nij,njhw->nihw", cmat.view(N, 3, 3).to(pooled.dtype), val)
    if coff is not None:
        out = out + coff.view(N, 3, 1, 1).to(pooled.dtype)
    if cj is not None:
        from . import kornia_aug as ka
        c = cj.view(N, 8).to(pooled.dtype)
        on = c[:, 0] != 0
        if bool(on.any()):
            idx = on.nonzero().squeeze(1)
            code = int(c[idx[0], 5])
            order = [(code >> (2 * k)) & 3 for k in range(4)]
            out = out.index_copy(0, idx, ka.color_jitter(out[idx], c[idx, 1], c[idx, 2], c[idx, 3], c[idx, 4], order))
    e = erase.view(N, 4)
    er = (xs[None] >= e[:, 0].view(N, 1, 1)) & (xs[None] < e[:, 2].view(N, 1, 1)) & \
         (ys[None] >= e[:, 1].view(N, 1, 1)) & (ys[None] < e[:, 3].view(N, 1, 1))
    out = out * (~er)[:, None].to(pooled.dtype)
    if facs is not None:
        out = out + facs.view(N, 1, 1, 1) * noise
    return out
