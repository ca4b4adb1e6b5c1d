"""GPU parity of the norm / softmax / glue kernels against plain PyTorch fp32/fp64 math of the same op.
Tolerances: fp32 storage 2e-5 rel (op order only); bf16 storage adds one 2^-8 rounding of the output."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

DT = [torch.bfloat16, torch.float16, torch.float32]


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def _mk(shape, dtype, dev, seed, scale=1.0, shift=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).to(dtype).to(dev)


def _tol(dt):
    return {torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3}.get(dt, 3e-5)


@pytest.mark.parametrize("xdt", DT)
@pytest.mark.parametrize("ydt", DT)
@pytest.mark.parametrize("rows,dim", [(515, 1024), (70, 768), (33, 12), (9, 256)])
def test_layernorm(cuda, xdt, ydt, rows, dim):
    x = _mk((rows, dim), xdt, cuda, 1, 2.0, 0.5)
    g, b = _mk((dim,), torch.float32, cuda, 2), _mk((dim,), torch.float32, cuda, 3)
    y, mean, rstd = K.layernorm_fwd(x, g, b, ydt)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.layer_norm(xd, (dim,), gd, bd, 1e-5)
    assert _rel(y, ref) < _tol(ydt)
    dy = _mk((rows, dim), ydt, cuda, 4)
    dres = _mk((rows, dim), xdt, cuda, 5)
    dx, dg, db = K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_param_grads=True)
    ref.backward(dy.double())
    assert _rel(dx, xd.grad + dres.double()) < _tol(xdt)
    assert _rel(dg, gd.grad) < 3e-5 and _rel(db, bd.grad) < 3e-5
    dx2, _, _ = K.layernorm_bwd(dy, x, g, mean, rstd)
    assert _rel(dx2, xd.grad) < _tol(xdt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("swish", [True, False])
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 512), (3, 7, 9, 128), (1, 64, 64, 256), (2, 4, 4, 32)])
def test_groupnorm(cuda, dt, swish, B, H, W, C):
    x = _mk((B, H, W, C), dt, cuda, 1, 1.5, 0.3)
    g, b = _mk((C,), torch.float32, cuda, 2, 0.5, 1.0), _mk((C,), torch.float32, cuda, 3, 0.2)
    y, mean, rstd = K.groupnorm_fwd(x, g, b, swish=swish)
    xd = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.group_norm(xd, 32, g.double(), b.double(), 1e-6)
    if swish:
        ref = ref * torch.sigmoid(ref)
    assert _rel(y, ref.permute(0, 2, 3, 1)) < _tol(dt)
    dy, dres = _mk((B, H, W, C), dt, cuda, 4), _mk((B, H, W, C), dt, cuda, 5)
    dx = K.groupnorm_bwd(dy, x, g, b, mean, rstd, dres=dres, swish=swish)
    ref.backward(dy.double().permute(0, 3, 1, 2))
    assert _rel(dx, xd.grad.permute(0, 2, 3, 1) + dres.double()) < _tol(dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("T,ld", [(50, 64), (77, 80), (256, 256), (1000, 1024)])
def test_softmax(cuda, dt, causal, T, ld):
    R = 3 * T
    s = _mk((R, ld), torch.float32, cuda, 1, 3.0)
    p = torch.full((R, ld), 7.0, dtype=dt, device=cuda)
    K.softmax_fwd(s, p, R, T, ld, ld, scale=0.37, causal=causal, q_len=T)
    sd = (s[:, :T].double() * 0.37).requires_grad_(True)
    sm = sd
    if causal:
        q = torch.arange(R, device=cuda) % T
        mask = torch.arange(T, device=cuda)[None, :] > q[:, None]
        sm = sd.masked_fill(mask, float("-inf"))
    ref = sm.softmax(-1)
    assert _rel(p[:, :T], ref) < _tol(dt)
    assert (p[:, T:] == 0).all()
    dp = _mk((R, ld), torch.float32, cuda, 2)
    ds = torch.empty_like(p)
    K.softmax_bwd(p, dp, ds, R, T, ld, ld, scale=0.37)
    pd = p[:, :T].double()
    dref = 0.37 * pd * (dp[:, :T].double() - (pd * dp[:, :T].double()).sum(-1, keepdim=True))
    assert _rel(ds[:, :T], dref) < _tol(dt)


def test_cast_transpose_colsum(cuda):
    x = _mk((5, 70, 130), torch.float32, cuda, 1)
    xb = K.cast(x, torch.bfloat16)
    assert torch.equal(xb, x.bfloat16())
    assert torch.equal(K.cast(xb, torch.float32), xb.float())
    odd = _mk((1027,), torch.float32, cuda, 2)
    assert torch.equal(K.cast(odd, torch.bfloat16), odd.bfloat16())
    t = K.transpose(x, torch.bfloat16)
    assert torch.equal(t, x.transpose(1, 2).contiguous().bfloat16())
    t2 = K.transpose(xb)
    assert torch.equal(t2, xb.transpose(1, 2).contiguous())
    for dt in DT:
        m = _mk((3000, 200), dt, cuda, 3)
        out = torch.ones(200, dtype=torch.float32, device=cuda)
        K.colsum(m, out, accumulate=True)
        assert _rel(out, m.double().sum(0) + 1.0) < 3e-5
        K.colsum(m, out)
        assert _rel(out, m.double().sum(0)) < 3e-5


def test_clamp_with_grad(cuda):
    x = _mk((4, 37, 11), torch.float32, cuda, 1, 2.0)
    g = _mk((4, 37, 11), torch.float32, cuda, 2)
    y = K.clamp_fwd(x, torch.float32, 0.5, 0.5, 0.0, 1.0)
    u = x * 0.5 + 0.5
    assert torch.allclose(y, u.clamp(0, 1))
    dx = K.clamp_bwd(x, g, 0.5, 0.5, 0.0, 1.0)
    ref = 0.5 * g * ((g * (u - u.clamp(0, 1))) >= 0)
    assert torch.allclose(dx, ref)


@pytest.mark.parametrize("dt", DT)
def test_sumpool(cuda, dt):
    x = _mk((2, 6, 10, 64), dt, cuda, 1)
    y = K.sumpool2x2(x)
    ref = F.avg_pool2d(x.double().permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1) * 4
    assert _rel(y, ref) < _tol(dt)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,codes,C", [(300, 1000, 64), (1000, 260, 24), (4096, 16384, 256), (513, 16384, 256)])
def test_vq_argmin_inside_the_distance_gemm(cuda, dt, rows, codes, C):
    """FFVC_F_VQ_ARGMIN (main.py:133-139): the split-precision distance GEMM keeps the argmin itself — every index must equal the
    two-launch form (fp32 distance matrix + ffvc_vq_argmin) bit for bit, ragged row / column tiles and exact ties included
    (duplicated codebook rows: the first one wins, as torch.argmin)."""
    x, cb = _mk((rows, C), torch.float32, cuda, 1), _mk((codes, C), torch.float32, cuda, 2)
    cb[codes // 2] = cb[3]
    cb[codes - 1] = cb[0]
    x[5] = cb[3]                                  # row 5 sits exactly on a duplicated code
    xn, cn = K.rownorm_sq(x), K.rownorm_sq(cb)
    x3, cb3 = K.split3(x, dt), K.split3(cb, dt, weight_order=True)
    assert K.vq_fused_ok(dt, codes, 3 * C)
    dot = torch.empty(rows, codes, dtype=torch.float32, device=cuda)
    K.gemm(x3, cb3, dot, rows, codes, 3 * C, ldx=3 * C, ldw=3 * C)
    ref = K.vq_argmin(dot, xn, cn)
    idx = K.vq_argmin_fused(x3, cb3, xn, cn)
    assert idx.dtype == torch.int64 and idx.shape == ref.shape
    assert int(idx.min()) >= 0 and int(idx.max()) < codes
    assert torch.equal(idx, ref), f"{int((idx != ref).sum())} of {rows} indices differ"
    assert int(idx[5]) == 3 and (idx != codes // 2).all() and (idx != codes - 1).all()
    d = x.double().pow(2).sum(-1, keepdim=True) + cb.double().pow(2).sum(1) - 2 * x.double() @ cb.double().T
    assert (idx == d.argmin(-1)).float().mean() > 0.995


def test_vq_argmin_flag_is_refused_where_no_kernel_implements_it(cuda):
    """fp32 operands (or an odd reduction depth) do not take the 256x256 LDS-DMA kernel: the launch must fail loudly, not store nothing."""
    x, cb = _mk((64, 32), torch.float32, cuda, 1), _mk((128, 32), torch.float32, cuda, 2)
    packed = torch.full((64,), -1, dtype=torch.int64, device=cuda)
    with pytest.raises(RuntimeError):
        K.gemm(x, cb, packed, 64, 128, 32, ldx=32, ldw=32, vq=(K.rownorm_sq(x), K.rownorm_sq(cb), packed))
    assert (packed == -1).all()


def test_vq_and_gather(cuda):
    x, cb = _mk((300, 64), torch.float32, cuda, 1), _mk((1000, 64), torch.float32, cuda, 2)
    xn, cn = K.rownorm_sq(x), K.rownorm_sq(cb)
    assert _rel(xn, x.double().pow(2).sum(-1)) < 1e-6
    dot = torch.empty(300, 1000, dtype=torch.float32, device=cuda)
    K.gemm(x, cb, dot, 300, 1000, 64, ldx=64, ldw=64)
    idx = K.vq_argmin(dot, xn, cn)
    d = x.double().pow(2).sum(-1, keepdim=True) + cb.double().pow(2).sum(1) - 2 * x.double() @ cb.double().T
    assert (idx == d.argmin(-1)).float().mean() > 0.995
    # exact tie -> first index
    cb2 = cb.clone()
    cb2[500] = cb2[3]
    K.gemm(x, cb2, dot, 300, 1000, 64, ldx=64, ldw=64)
    idx2 = K.vq_argmin(dot, xn, K.rownorm_sq(cb2))
    assert (idx2 != 500).all()
    zq = K.gather_rows(cb, idx, torch.float32)
    assert torch.equal(zq, cb[idx])
    pos = _mk((7, 64), torch.float32, cuda, 3)
    tok = torch.randint(0, 1000, (6, 7), device=cuda)
    e = K.gather_rows(cb, tok, torch.bfloat16, pos=pos, period=7)
    assert torch.equal(e, (cb[tok] + pos).bfloat16())
    tok[:, 0] = 999
    tok[2, 5] = 999  # tie: first wins
    tok[3, 0] = 5
    tok[3, 4] = 999
    xx = _mk((6, 7, 64), torch.float32, cuda, 4)
    out = K.eot_gather(xx, tok)
    assert torch.equal(out, xx[torch.arange(6, device=cuda), tok.argmax(-1)])


@pytest.mark.parametrize("odt", DT)
@pytest.mark.parametrize("H,cut,P", [(256, 224, 32), (40, 32, 8), (64, 64, 16)])
def test_cutouts(cuda, odt, H, cut, P):
    B, cutn = 2, 3
    g = torch.Generator().manual_seed(1)
    xr = torch.rand(B, H, H, 3, generator=g).to(cuda)
    noise = torch.randn(cutn * B, 3, cut, cut, generator=g).to(cuda)
    facs = (torch.rand(cutn * B, generator=g) * 0.1).to(cuda)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    out = K.cutouts_fwd(xr, cut, cutn, P, mean, std, odt, noise=noise, facs=facs)
    xd = xr.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    c = (F.adaptive_avg_pool2d(xd, cut) + F.adaptive_max_pool2d(xd, cut)) / 2
    batch = c.repeat(cutn, 1, 1, 1) + facs.double().view(-1, 1, 1, 1) * noise.double()
    m = torch.tensor(mean, device=cuda, dtype=torch.float64).view(1, 3, 1, 1)
    s = torch.tensor(std, device=cuda, dtype=torch.float64).view(1, 3, 1, 1)
    ref = (batch - m) / s
    gw = cut // P
    ref_p = ref.view(cutn * B, 3, gw, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(cutn * B, gw * gw, 3 * P * P)
    assert _rel(out, ref_p) < _tol(odt)
    gout = _mk(tuple(out.shape), odt, cuda, 5)
    dxr = K.cutouts_bwd(xr, gout, cut, cutn, P, std)
    ref_p.backward(gout.double())
    assert _rel(dxr, xd.grad.permute(0, 2, 3, 1)) < 3e-5


def test_spherical_loss(cuda):
    N, B, D = 24, 8, 512
    e = _mk((N, D), torch.float32, cuda, 1)
    f = _mk((B, D), torch.float32, cuda, 2)
    loss, de = K.spherical_loss(e, f, coef=1.0)
    ed = e.double().requires_grad_(True)
    Hn = F.normalize(f.double().repeat(N // B, 1), dim=-1)
    En = F.normalize(ed, dim=1)
    ref = Hn.sub(En).norm(dim=-1).div(2).arcsin().pow(2).mul(2).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) / ref.item() < 1e-6
    assert _rel(de, ed.grad) < 1e-5


@pytest.mark.parametrize("sdt", [None, torch.bfloat16, torch.float16])
def test_adam(cuda, sdt):
    n = 10007
    p, g = _mk((n,), torch.float32, cuda, 1), _mk((n,), torch.float32, cuda, 2)
    pp = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([pp], lr=1e-2)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sh = torch.empty(n, dtype=sdt, device=cuda) if sdt else None
    for step in range(1, 4):
        gg = g * step
        pp.grad = gg.clone()
        opt.step()
        K.adam(p, gg, m, v, sh, 1e-2, 0.9, 0.999, 1e-8, step)
        assert _rel(p, pp.data) < 1e-6
    if sdt:
        assert torch.equal(sh, p.to(sdt))
    ss = torch.zeros(1, dtype=torch.float32, device=cuda)
    K.sumsq(g, ss)
    assert abs(ss.item() - g.double().pow(2).sum().item()) / ss.item() < 1e-5
    y = torch.ones_like(g)
    K.axpby(g, y, 2.0, 3.0)
    assert torch.allclose(y, 2 * g + 3)


@pytest.mark.parametrize("odt", DT)
def test_augment_matches_oracle_reference(cuda, odt):
    """Fused default-augmentation kernel (fwd + bwd) vs the plain-torch statement of the same resampling math."""
    from feed_forward_vqgan_clip_amd import augment as A
    from oracle import step as ostep
    B, S, cutn, P = 3, 32, 4, 8
    g = torch.Generator().manual_seed(7)
    prm = A.draw_params(cutn * B, S, generator=g)
    prm["erase"][:] = torch.tensor([5, 9, 17, 20], dtype=torch.int32)           # force an erased rectangle
    pooled = torch.rand(B, 3, S, S, generator=g)
    noise = torch.randn(cutn * B, 3, S, S, generator=g)
    facs = torch.rand(cutn * B, generator=g) * 0.1
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    dev = {k: v.to(cuda) for k, v in prm.items()}
    assert prm["cj"][:, 0].sum() > 0                                            # the kornia ColorJitter path is exercised
    out = K.augment_fwd(pooled.to(cuda), dev["pinv"], dev["ainv"], dev["cmat"], dev["erase"], cutn, P, mean, std, odt,
                        noise=noise.to(cuda), facs=facs.to(cuda), coff=dev["coff"], cj=dev["cj"])
    pd = pooled.double().requires_grad_(True)
    ref = ostep.augment_reference(pd, prm["pinv"].double(), prm["ainv"].double(), prm["cmat"].double(), prm["erase"], cutn,
                                  facs.double(), noise.double(), coff=prm["coff"].double(), cj=prm["cj"])
    m = torch.tensor(mean, dtype=torch.float64).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float64).view(1, 3, 1, 1)
    refn = (ref - m) / s
    gw = S // P
    ref_p = refn.view(cutn * B, 3, gw, P, gw, P).permute(0, 2, 4, 1, 3, 5).reshape(cutn * B, gw * gw, 3 * P * P)
    # a handful of pixels sit exactly on a floor()/mask boundary where fp32 and fp64 coordinates may disagree
    err = (out.double().cpu() - ref_p).abs()
    tol = {torch.bfloat16: 3e-2, torch.float16: 4e-3}.get(odt, 1e-3)
    assert (err > tol).float().mean().item() < 2e-3, f"mismatching fraction {(err > tol).float().mean().item()}"
    gout = _mk(tuple(out.shape), odt, cuda, 5)
    dp = K.augment_bwd(gout, dev["pinv"], dev["ainv"], dev["cmat"], dev["erase"], B, S, cutn, P, std, pooled=pooled.to(cuda),
                       coff=dev["coff"], cj=dev["cj"])
    ref_p.backward(gout.double().cpu())
    rel = ((dp.double().cpu() - pd.grad).abs().max() / pd.grad.abs().max()).item()
    assert rel < 2e-2, rel


def test_augment_identity_params_equal_plain_cutouts(cuda):
    """With every augmentation switched off the fused path must reproduce the 'R' path exactly."""
    from feed_forward_vqgan_clip_amd import augment as A
    B, H, cut, cutn, P = 2, 40, 32, 3, 8
    g = torch.Generator().manual_seed(1)
    xr = torch.rand(B, H, H, 3, generator=g).to(cuda)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    plain = K.cutouts_fwd(xr, cut, cutn, P, mean, std, torch.float32)
    prm = {k: v.to(cuda) for k, v in A.draw_params(cutn * B, cut, augs=(), generator=g).items()}
    pooled = K.cutouts_fwd(xr, cut, 1, cut, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), torch.float32).view(B, 3, cut, cut)
    fused = K.augment_fwd(pooled, prm["pinv"], prm["ainv"], prm["cmat"], prm["erase"], cutn, P, mean, std, torch.float32)
    assert _rel(fused, plain) < 1e-6


@pytest.mark.parametrize("ldt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,heads", [(3, 50, 12), (2, 64, 2), (5, 17, 1), (2, 33, 3)])
def test_attention_small_fused(cuda, B, T, heads, ldt):
    """Fused short-sequence attention (one wave per (item, head)) vs fp64 math on the same bf16 inputs, fwd + bwd,
    and vs the GEMM + softmax path it replaces."""
    from feed_forward_vqgan_clip_amd import ops
    D = heads * 64
    qkv = _mk((B, T, 3 * D), ldt, cuda, 1, 0.7)
    do = _mk((B, T, D), ldt, cuda, 2)
    scale = 64 ** -0.5
    f = 1.0 if ldt == torch.bfloat16 else 0.125          # f16 rounds 8x finer
    assert K.attn_small_ok(qkv, heads, False)
    o = K.attn_small_fwd(qkv, heads, scale)
    dqkv = K.attn_small_bwd(qkv, do, heads, scale)
    x = qkv.double().requires_grad_(True)
    q, k, v = [t.view(B, T, heads, 64).transpose(1, 2) for t in x.split(D, dim=-1)]
    p = (q @ k.transpose(-1, -2) * scale).softmax(-1)
    ref = (p @ v).transpose(1, 2).reshape(B, T, D)
    ref.backward(do.double())
    assert _rel(o, ref) < 1.2e-2 * f
    assert _rel(dqkv, x.grad) < 2.5e-2 * f
    # the unfused path on the same inputs
    import os
    os.environ["FFVC_ATTN_SMALL"] = "0"
    try:
        x2 = qkv.clone().requires_grad_(True)
        o2 = ops.attention(x2, heads, scale)
        o2.backward(do)
    finally:
        os.environ.pop("FFVC_ATTN_SMALL")
    assert _rel(o, o2) < 1.2e-2 * f and _rel(dqkv, x2.grad) < 2.5e-2 * f
    x3 = qkv.clone().requires_grad_(True)
    o3 = ops.attention(x3, heads, scale)
    o3.backward(do)
    assert torch.equal(o3, o) and torch.equal(x3.grad, dqkv)


@pytest.mark.parametrize("ldt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("B,T,heads", [(2, 257, 3), (1, 1024, 2), (3, 100, 1), (2, 64, 2), (1, 65, 1), (2, 7, 2)])
def test_attention_flash(cuda, B, T, heads, causal, ldt):
    """Flash-style attention (online softmax over 64-key blocks, no score matrix in HBM) vs fp64 math on the same 16-bit
    inputs, forward + backward, causal and not; and vs the GEMM + softmax path it replaces (transformer.py:11-20)."""
    import os
    from feed_forward_vqgan_clip_amd import ops
    D = heads * 64
    qkv = _mk((B, T, 3 * D), ldt, cuda, 1, 0.7)
    do = _mk((B, T, D), ldt, cuda, 2)
    scale = 64 ** -0.5
    f = 1.0 if ldt == torch.bfloat16 else 0.125
    assert K.attn_flash_ok(qkv, heads)
    o, lse = K.attn_flash_fwd(qkv, heads, scale, causal)
    dqkv = K.attn_flash_bwd(qkv, o, do, lse, heads, scale, causal)
    x = qkv.double().requires_grad_(True)
    q, k, v = [t.view(B, T, heads, 64).transpose(1, 2) for t in x.split(D, dim=-1)]
    s = q @ k.transpose(-1, -2) * scale
    if causal:
        s = s.masked_fill(torch.ones(T, T, dtype=torch.bool, device=cuda).triu(1), float("-inf"))
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, T, D)
    ref.backward(do.double())
    assert _rel(o, ref) < 1.2e-2 * f
    assert _rel(dqkv, x.grad) < 2.5e-2 * f
    ref_lse = torch.logsumexp(s, -1).reshape(B * heads, T) / math.log(2.0)
    assert (lse.double() - ref_lse).abs().max().item() < 2e-2 * f + 1e-3
    os.environ["FFVC_ATTN_FLASH"] = "0"
    os.environ["FFVC_ATTN_SMALL"] = "0"
    try:
        x2 = qkv.clone().requires_grad_(True)
        o2 = ops.attention(x2, heads, scale, causal)
        o2.backward(do)
    finally:
        os.environ.pop("FFVC_ATTN_FLASH")
        os.environ.pop("FFVC_ATTN_SMALL")
    assert _rel(o, o2) < 1.2e-2 * f and _rel(dqkv, x2.grad) < 2.5e-2 * f
    if T > 64 or causal:          # what ops.attention dispatches to for these shapes
        x3 = qkv.clone().requires_grad_(True)
        o3 = ops.attention(x3, heads, scale, causal)
        o3.backward(do)
        assert torch.equal(o3, o) and torch.equal(x3.grad, dqkv)


def test_bad_arguments_fail_loudly(cuda):
    """Every entry point validates its arguments and reports through ffvc_last_error (no silent fallback)."""
    from feed_forward_vqgan_clip_amd._lib import FFVCError
    x = _mk((4, 65, 3 * 64), torch.bfloat16, cuda, 1)                       # T = 65 > 64: outside the fused attention
    assert not K.attn_small_ok(x, 1, False)
    with pytest.raises(FFVCError, match="T <= 64"):
        K.attn_small_fwd(x, 1, 0.125)
    with pytest.raises(FFVCError):
        K.layernorm_fwd(_mk((4, 5000), torch.float32, cuda, 1), _mk((5000,), torch.float32, cuda, 2),
                        _mk((5000,), torch.float32, cuda, 3), torch.bfloat16)     # dim > 64 * LN_MAXE
    with pytest.raises((FFVCError, TypeError)):
        K.colsum(_mk((8, 8), torch.float32, cuda, 1), torch.zeros(8, dtype=torch.bfloat16, device=cuda))
    with pytest.raises((FFVCError, RuntimeError, TypeError)):
        K.cast(torch.zeros(4), torch.bfloat16)                                # CPU tensor: there is no CPU path


def test_degenerate_sizes(cuda):
    """Smallest legal problems: one row, one token, one image."""
    y, mean, rstd = K.layernorm_fwd(_mk((1, 8), torch.float32, cuda, 1), torch.ones(8, device=cuda), torch.zeros(8, device=cuda),
                                    torch.float32)
    assert abs(y.mean().item()) < 1e-5
    qkv = _mk((1, 1, 192), torch.bfloat16, cuda, 2)                           # one token attends to itself: o == v
    o = K.attn_small_fwd(qkv, 1, 0.125)
    assert torch.equal(o.view(-1), qkv.view(-1)[128:])
    dq = K.attn_small_bwd(qkv, o, 1, 0.125)
    assert dq[..., :128].abs().max().item() == 0 and torch.equal(dq.view(-1)[128:], o.view(-1))
    m = _mk((1, 4), torch.bfloat16, cuda, 3)
    out = torch.zeros(4, device=cuda)
    K.colsum(m, out)
    assert torch.equal(out, m.float().view(-1))


@pytest.mark.parametrize("dt", DT)
def test_transpose_multi(cuda, dt):
    shapes = [(1024, 256), (70, 130), (1, 5), (64, 64), (129, 63)]
    pairs = [(_mk(s, dt, cuda, i), torch.zeros(s[1], s[0], dtype=dt, device=cuda)) for i, s in enumerate(shapes)]
    plan = K.TransposePlan(pairs)
    plan.run()
    for src, dst in pairs:
        assert torch.equal(dst, src.t().contiguous())


# ----------------------------------------------------------------------------- fused token-mixing MLP
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,D,O", [(3, 256, 320, 1024), (2, 128, 64, 512), (2, 256, 1024, 1024), (1, 128, 288, 160)])
def test_tokmix_fused_kernels_vs_fp64(cuda, dt, B, T, D, O):
    """ffvc_tokmix_fwd / ffvc_tokmix_bwd_hidden (mlp_mixer_pytorch.py:28,34) vs fp64 math on the same 16-bit operands,
    incl. a D that is not a multiple of the 256-column workgroup tile and a hidden size that is not a multiple of 64."""
    import torch.nn.functional as F
    tol = 1e-2 if dt == torch.bfloat16 else 1.3e-3
    xn = _mk((B, T, D), dt, cuda, 1)
    w1, w2 = _mk((O, T), dt, cuda, 2, T ** -0.5), _mk((T, O), dt, cuda, 3, O ** -0.5)
    b1, b2 = _mk((O,), torch.float32, cuda, 4, 0.3), _mk((T,), torch.float32, cuda, 5, 0.3)
    res = _mk((B, T, D), torch.float32, cuda, 6)
    assert K.tokmix_supported(dt, T, D, O)
    y = K.tokmix_fwd(xn, w1, b1, w2, b2, res)
    pre = w1.double() @ xn.double() + b1.double()[None, :, None]
    h = F.gelu(pre)
    hr = h.to(dt).double()                                   # the kernel rounds the hidden activation to the storage dtype
    ref = w2.double() @ hr + b2.double()[None, :, None] + res.double()
    assert _rel(y, ref) < tol / 8                             # fp32 output: only the boundary flips of the h rounding remain
    dy = _mk((B, T, D), dt, cuda, 7)
    hk, dhk = K.tokmix_bwd_hidden(xn, dy, w1, b1, w2.t().contiguous())
    p = pre.clone().requires_grad_(True)
    F.gelu(p).sum().backward()
    dref = (w2.double().t() @ dy.double()) * p.grad
    assert _rel(hk, h) < tol and _rel(dhk, dref) < tol
    # the forward that saves h and act'(pre) for the backward: same y and h to the bit, act' vs autograd's derivative
    ys, hs, gs = K.tokmix_fwd_save(xn, w1, b1, w2, b2, res)
    assert torch.equal(ys, y) and torch.equal(hs, hk)
    assert _rel(gs, p.grad) < tol
    assert torch.equal(K.tokmix_fwd_save(xn, w1, b1, w2, b2, res)[2], gs)     # staging race check: reproducible


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("save", ["1", "0"])
def test_tokmix_fused_autograd_matches_unfused(cuda, dt, save, monkeypatch):
    """ops.token_mlp with the fused kernels (forward recompute in backward) vs the two-GEMM path: output, dx and every
    parameter gradient."""
    import os
    from feed_forward_vqgan_clip_amd import ops
    from feed_forward_vqgan_clip_amd.mappers import Mixer
    torch.manual_seed(3)
    monkeypatch.setattr(ops, "_TM_SAVE", save == "1")      # fused forward with saved h / act' | recompute in the backward
    outs = {}
    for mode in ("1", "0"):
        os.environ["FFVC_TOKMIX"] = mode
        try:
            torch.manual_seed(3)
            net = Mixer(input_dim=32, image_size=16, channels=16, patch_size=1, dim=64, depth=1).cuda().prepare(dt)
            (n1, t1, t2, n2, c1, c2) = net._blocks[0]
            g = torch.Generator().manual_seed(1)
            x = torch.randn(3, 256, 64, generator=g).cuda().requires_grad_(True)
            hn, hid = ops.layernorm_fork(x, n1.weight, n1.bias, dt)
            y = ops.token_mlp(hn, t1, t2, residual=hid, out_dtype=torch.float32)
            gw = torch.randn(3, 256, 64, generator=g).cuda()
            net._ffvc_arena.zero_grad()
            (y * gw).sum().backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            outs[mode] = (y.detach().clone(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()
                                                               if "mixer.2.0" in k})
        finally:
            os.environ.pop("FFVC_TOKMIX")
    tol = 2e-2 if dt == torch.bfloat16 else 3e-3
    (y1, dx1, g1), (y0, dx0, g0) = outs["1"], outs["0"]
    assert _rel(y1, y0) < tol / 4 and _rel(dx1, dx0) < tol
    assert len(g1) == 6
    for k in g1:
        assert _rel(g1[k], g0[k]) < tol, k


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols,period", [(64 * 256, 1024, 256), (16 * 96, 72, 96), (40, 24, 8), (7 * 5, 16, 5)])
def test_rowsum_grouped_and_row_forms(cuda, dt, rows, cols, period):
    """ffvc_rowsum: out[r % period] (+)= sum_c x[r, c] (token-mixing bias gradients, mlp_mixer_pytorch.py:28) — the grouped form
    (one wave per output and row chunk, 16-bit inputs) and the row-per-wave form, with and without accumulate."""
    g = torch.Generator().manual_seed(rows + cols)
    x = torch.randn(rows, cols, generator=g).to(dt).cuda()
    ref = x.double().sum(1).view(-1, period).sum(0)
    out = torch.full((period,), 3.0, device="cuda")
    K.rowsum(x, out, period, accumulate=True)
    tol = 2e-5 * max(1.0, float(ref.abs().max())) * (cols * rows / period) ** 0.5
    assert float((out.double().cpu() - (ref.cpu() + 3.0)).abs().max()) < tol
    K.rowsum(x, out, period, accumulate=False)
    assert float((out.double().cpu() - ref.cpu()).abs().max()) < tol


@pytest.mark.parametrize("nslab", [1, 2, 3, 4, 5, 8])
def test_slab_reduce_every_specialisation(cuda, nslab):
    g = torch.Generator().manual_seed(nslab)
    slabs = torch.randn(nslab, 300, 44, generator=g).cuda()
    y = torch.randn(300, 44, generator=g).cuda()
    want = y.double() + slabs.double().sum(0)
    from feed_forward_vqgan_clip_amd.kernels import _call, stream_ptr
    _call("ffvc_slab_reduce", slabs.data_ptr(), y.data_ptr(), 300 * 44, nslab, 1, stream_ptr())
    assert float((y.double() - want).abs().max()) < 1e-5
    _call("ffvc_slab_reduce", slabs.data_ptr(), y.data_ptr(), 300 * 44, nslab, 0, stream_ptr())
    assert float((y.double() - slabs.double().sum(0)).abs().max()) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,T,H,dh,layout,pad", [(3, 16, 6, 170, "dkh", 0), (2, 16, 6, 170, "dkh", 4), (2, 9, 3, 40, "khd", 8),
                                                 (1, 32, 2, 96, "dkh", 0)])
def test_attention_tiny_vs_fp64(cuda, dt, B, T, H, dh, layout, pad):
    """One-workgroup-per-(sample, head) attention (csrc/attn_tiny.hip) against fp64 autograd, reading q/k/v in the reference's
    '(d k h)' column order (vitgan.py:81-82) or '(k h d)', with and without padded rows."""
    from feed_forward_vqgan_clip_amd import kernels as K
    assert K.attn_tiny_ok(T, dh)
    g = torch.Generator().manual_seed(5)
    row, out_ld = 3 * H * dh + pad, H * dh + pad
    qkv = torch.randn(B, T, row, generator=g).to(dt).cuda()
    do = torch.randn(B, T, out_ld, generator=g).to(dt).cuda()
    scale = 0.11
    o = K.attn_tiny_fwd(qkv, H, dh, scale, layout, out_ld)
    dqkv = K.attn_tiny_bwd(qkv, do, H, dh, scale, layout)
    x = qkv.double()[:, :, :3 * H * dh].detach().requires_grad_(True)
    if layout == "dkh":
        q, k, v = x.view(B, T, dh, 3, H).permute(3, 0, 4, 1, 2)          # 'b t (d k h) -> k b h t d'
    else:
        q, k, v = x.view(B, T, 3, H, dh).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * scale, dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, T, H * dh)              # 'b h t d -> b t (h d)'
    ref.backward(do.double()[:, :, :H * dh])
    tol = {torch.float32: 2e-5, torch.float16: 2e-3, torch.bfloat16: 1.5e-2}[dt]
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()
    assert rel(o[:, :, :H * dh], ref.detach()) < tol
    assert rel(dqkv[:, :, :3 * H * dh], x.grad) < tol
    if pad:
        assert torch.count_nonzero(o[:, :, H * dh:]) == 0 and torch.count_nonzero(dqkv[:, :, 3 * H * dh:]) == 0


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("B,T,heads", [(3, 77, 8), (2, 128, 2), (1, 5, 1), (4, 16, 3)])
def test_attention_text_fp32_one_launch(cuda, B, T, heads, causal):
    """ffvc_attn_text_fwd: exact-fp32 attention of a short sequence in one launch (the frozen CLIP text tower: 77 causal tokens) against fp64."""
    g = torch.Generator().manual_seed(23)
    qkv = (torch.randn(B, T, 3 * heads * 64, generator=g) * 0.8).cuda()
    o = K.attn_text_fwd(qkv, heads, 0.125, causal)
    q, k, v = [t.double().view(B, T, heads, 64).permute(0, 2, 1, 3) for t in qkv.split(heads * 64, dim=2)]
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s.masked_fill(torch.ones(T, T, dtype=torch.bool, device=s.device).triu(1), float("-inf"))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, heads * 64)
    assert ((o.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
