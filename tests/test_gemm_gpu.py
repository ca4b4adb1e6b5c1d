"""GPU parity of ffvc_gemm (every operand mode / dtype / epilogue) against fp64 torch math.

Tolerances: with fp32 output the only error is accumulation order (<= 2e-5 rel for bf16
inputs rounded beforehand, 2e-5 for fp32 MFMA); with bf16 output one extra rounding (2^-8).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


def _mk(shape, dtype, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


DT = [torch.bfloat16, torch.float16, torch.float32]
LOTOL = {torch.bfloat16: 1e-2, torch.float16: 1.3e-3, torch.float32: 2e-5}     # one rounding of the stored output


def test_probe_tr16(cuda):
    t = K.probe_tr16()
    # documented semantic: lane 16g+c, elem j <- source lane 16g + 4j + (c>>2), elem (c&3); source lane s holds 4s..4s+3
    exp = torch.empty(64, 4, dtype=torch.int16)
    for lane in range(64):
        g, c = lane // 16, lane % 16
        for j in range(4):
            src_lane = 16 * g + 4 * j + (c >> 2)
            exp[lane, j] = 4 * src_lane + (c & 3)
    print("tr16 probe lanes 0..19:\n", t[:20])
    assert torch.equal(t, exp), f"ds_read_b64_tr_b16 semantic differs:\n{t}"


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K_", [(300, 200, 136), (128, 128, 64), (1024, 512, 1024), (65, 3, 72)])
def test_nt_bias_f32out(cuda, dt, M, N, K_):
    x, w = _mk((M, K_), dt, cuda, 1), _mk((N, K_), dt, cuda, 2)
    b = _mk((N,), torch.float32, cuda, 3)
    y = torch.empty(M, N, dtype=torch.float32, device=cuda)
    K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_, bias=b)
    ref = x.double() @ w.double().T + b.double()
    assert _rel(y, ref) < 2e-5


@pytest.mark.parametrize("dt", DT)
def test_nt_act_preact_residual(cuda, dt):
    M, N, K_ = 384, 256, 192
    x, w = _mk((M, K_), dt, cuda, 1, 0.5), _mk((N, K_), dt, cuda, 2, 0.2)
    b = _mk((N,), torch.float32, cuda, 3)
    res = _mk((M, N), torch.float32, cuda, 4)
    for act, fn in [(K.ACT_GELU, lambda t: F.gelu(t)), (K.ACT_QUICKGELU, lambda t: t * torch.sigmoid(1.702 * t))]:
        y = torch.empty(M, N, dtype=dt, device=cuda)
        aux = torch.empty(M, N, dtype=dt, device=cuda)
        K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_, bias=b, residual=res, aux=aux, ldaux=N, act=act,
               flags=K.F_WRITE_PREACT)
        pre = x.double() @ w.double().T + b.double()
        ref = fn(pre) + res.double()
        tol = LOTOL[dt]
        assert _rel(aux, pre) < tol
        assert _rel(y, ref) < tol
        # backward-of-activation epilogue: dy @ w2 * act'(pre)
        dy, w2 = _mk((M, K_), dt, cuda, 5), _mk((N, K_), dt, cuda, 6, 0.2)
        g = torch.empty(M, N, dtype=dt, device=cuda)
        K.gemm(dy, w2, g, M, N, K_, ldx=K_, ldw=K_, aux=aux, ldaux=N, act=act, flags=K.F_MUL_ACT_GRAD)
        p = aux.double().requires_grad_(True)
        fn(p).sum().backward()
        gref = (dy.double() @ w2.double().T) * p.grad
        assert _rel(g, gref) < tol


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("safe", [False, True])
@pytest.mark.parametrize("M,N,K_", [(256, 384, 1000), (128, 128, 64), (200, 72, 130)])
def test_tn_wgrad(cuda, dt, safe, M, N, K_):
    # y[m,n] = sum_k xt[k,m] wt[k,n]  (both operands stored reduction-major)
    xt, wt = _mk((K_, M), dt, cuda, 1), _mk((K_, N), dt, cuda, 2)
    y = torch.zeros(M, N, dtype=torch.float32, device=cuda)
    K.gemm(xt, wt, y, M, N, K_, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS,
           flags=K.F_TR_SAFE if safe else 0)
    ref = xt.double().T @ wt.double()
    assert _rel(y, ref) < 2e-5
    # split-K with atomics accumulates on top of existing contents
    y2 = torch.ones(M, N, dtype=torch.float32, device=cuda)
    K.gemm(xt, wt, y2, M, N, K_, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS,
           flags=K.F_ATOMIC_OUT | (K.F_TR_SAFE if safe else 0), split_k=4)
    assert _rel(y2, ref + 1.0) < 2e-5


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("safe", [False, True])
def test_nn_and_tk(cuda, dt, safe):
    M, N, K_ = 192, 320, 200
    fl = K.F_TR_SAFE if safe else 0
    x, wt = _mk((M, K_), dt, cuda, 1), _mk((K_, N), dt, cuda, 2)
    y = torch.empty(M, N, dtype=torch.float32, device=cuda)
    K.gemm(x, wt, y, M, N, K_, ldx=K_, ldw=N, w_mode=K.OP_TRANS, flags=fl)
    assert _rel(y, x.double() @ wt.double()) < 2e-5
    xt, w = _mk((K_, M), dt, cuda, 3), _mk((N, K_), dt, cuda, 4)
    K.gemm(xt, w, y, M, N, K_, ldx=M, ldw=K_, x_mode=K.OP_TRANS, flags=fl)
    assert _rel(y, xt.double().T @ w.double().T) < 2e-5


@pytest.mark.parametrize("dt", DT)
def test_batched_maps_kseg(cuda, dt):
    # token-mix form: out[b][o, d] = sum_t W[o,t] xn[b][t,d] + bias[o] + res[b][o,d]
    B, T, D, O = 3, 64, 96, 160
    Wm, xn = _mk((O, T), dt, cuda, 1), _mk((B, T, D), dt, cuda, 2)
    bias = _mk((O,), torch.float32, cuda, 3)
    res = _mk((B, O, D), torch.float32, cuda, 4)
    out = torch.empty(B, O, D, dtype=torch.float32, device=cuda)
    K.gemm(Wm, xn, out, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, bias=bias, flags=K.F_BIAS_ALONG_M,
           residual=res, batch=B, wb=(T * D, 0), yb=(O * D, 0), rb=(O * D, 0))
    ref = torch.einsum("ot,btd->bod", Wm.double(), xn.double()) + bias.double()[None, :, None] + res.double()
    assert _rel(out, ref) < 2e-5
    # segmented-K wgrad of the same op: dW[o,t] = sum_{b,d} dy[b][o,d] xn[b][t,d]
    bk = 64 if dt != torch.float32 else 32
    D2 = 2 * bk
    dy, x2 = _mk((B, O, D2), dt, cuda, 5), _mk((B, T, D2), dt, cuda, 6)
    dW = torch.zeros(O, T, dtype=torch.float32, device=cuda)
    K.gemm(dy, x2, dW, O, T, B * D2, ldx=D2, ldw=D2, kseg=D2, xkso=O * D2, wkso=T * D2,
           flags=K.F_ATOMIC_OUT, split_k=3)
    assert _rel(dW, torch.einsum("bod,btd->ot", dy.double(), x2.double())) < 2e-5
    # two-level batch + output row map (attention-like head split): y[b, t, h, :] = q[b,t,h,:] @ k[b,s,h,:]^T
    Bq, H, Tq, dh = 2, 3, 50, 64
    qkv = _mk((Bq, Tq, 3 * H * dh), dt, cuda, 7)
    S = torch.empty(Bq * H, Tq, 64, dtype=torch.float32, device=cuda).fill_(-7.0)
    q = qkv[:, :, :H * dh]
    kk = qkv[:, :, H * dh:2 * H * dh]
    K.gemm(q, kk, S, Tq, Tq, dh, ldx=3 * H * dh, ldw=3 * H * dh, batch=Bq * H, batch_inner=H,
           xb=(Tq * 3 * H * dh, dh), wb=(Tq * 3 * H * dh, dh), yb=(H * Tq * 64, Tq * 64), y_map=(0, 0, 64),
           alpha=0.125)
    ref = torch.einsum("bthd,bshd->bhts", q.reshape(Bq, Tq, H, dh).double(), kk.reshape(Bq, Tq, H, dh).double()) * 0.125
    assert _rel(S.view(Bq, H, Tq, 64)[..., :Tq], ref) < 2e-5
    assert (S.view(Bq, H, Tq, 64)[..., Tq:] == -7.0).all()
    # y_map with split rows: m -> (m // mi) * so + (m % mi) * sm  (ViT patch rows into [n, 1+p, :])
    n_img, P, Wd = 4, 9, 128
    xp, wp = _mk((n_img * P, 64), dt, cuda, 8), _mk((Wd, 64), dt, cuda, 9)
    pos = _mk((P + 1, Wd), torch.float32, cuda, 10)
    tok = torch.zeros(n_img, P + 1, Wd, dtype=torch.float32, device=cuda)
    K.gemm(xp, wp, tok[:, 1:], n_img * P, Wd, 64, ldx=64, ldw=64, y_map=(P, (P + 1) * Wd, Wd),
           residual=pos[1:], r_map=(P, 0, Wd))
    ref = (xp.double() @ wp.double().T).view(n_img, P, Wd) + pos[1:].double()
    assert _rel(tok[:, 1:], ref) < 2e-5
    assert (tok[:, 0] == 0).all()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("ups", [False, True])
def test_conv3x3(cuda, dt, ups):
    B, Hin, Win, Cin, Cout = 2, 6, 10, 128, 192
    x = _mk((B, Hin, Win, Cin), dt, cuda, 1)
    w = _mk((Cout, 3, 3, Cin), dt, cuda, 2, 0.05)
    b = _mk((Cout,), torch.float32, cuda, 3)
    H, W = (2 * Hin, 2 * Win) if ups else (Hin, Win)
    res = _mk((B, H, W, Cout), dt, cuda, 4)
    y = torch.empty(B, H, W, Cout, dtype=dt, device=cuda)
    K.gemm(x, w, y, B * H * W, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, bias=b, residual=res,
           conv=(H, W, Cin), flags=K.F_UPSAMPLE2X if ups else 0)
    xn = x.double().permute(0, 3, 1, 2)
    if ups:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xn, w.double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    assert _rel(y, ref) < LOTOL[dt]


@pytest.mark.parametrize("W,Cin,Cout,ups,out32", [(64, 128, 3, False, True), (128, 64, 3, False, False), (256, 128, 3, False, True),
                                                  (128, 128, 5, True, True), (512, 64, 16, False, False)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_conv_narrow_output_kernel(cuda, W, Cin, Cout, ups, out32, dt):
    """conv_row_n16_kernel (the decoder's conv_out, 128 -> 3; taming Decoder.conv_out): <= 16 output channels on the haloed row
    tile with a 16-row filter tile, vs F.conv2d incl. borders, bias, fp32 / 16-bit output and the fused 2x upsample; the generic
    path (FFVC_CONV_N16=0 is process-wide, so compare against the reference instead) must agree too."""
    B, H = (1, min(W, 128)) if W >= 128 else (2, 64)
    Hin, Win = (H // 2, W // 2) if ups else (H, W)
    x = _mk((B, Hin, Win, Cin), dt, cuda, 1)
    w = _mk((Cout, 3, 3, Cin), dt, cuda, 2, 0.05)
    b = _mk((Cout,), torch.float32, cuda, 3)
    y = torch.full((B, H, W, Cout), float("nan"), dtype=torch.float32 if out32 else dt, device=cuda)
    K.gemm(x, w, y, B * H * W, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, bias=b, conv=(H, W, Cin),
           flags=K.F_UPSAMPLE2X if ups else 0)
    xn = x.double().permute(0, 3, 1, 2)
    if ups:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xn, w.double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
    assert torch.isfinite(y).all()
    assert _rel(y, ref) < (2e-5 if out32 else LOTOL[dt]) * (10 if out32 and dt == torch.bfloat16 else 1) + (0 if out32 else 0)


def test_bad_args_raise(cuda):
    x = torch.zeros(8, 8, dtype=torch.bfloat16, device=cuda)
    y = torch.zeros(8, 8, dtype=torch.bfloat16, device=cuda)
    from feed_forward_vqgan_clip_amd._lib import FFVCError
    with pytest.raises(FFVCError):
        K.gemm(x, x, y, 8, 8, 8, ldx=7, ldw=8)
    with pytest.raises(FFVCError):
        K.gemm(x, x, y, 8, 8, 8, ldx=8, ldw=8, split_k=2)


@pytest.mark.parametrize("dt", DT)
def test_dword_aligned_leading_dims(cuda, dt):
    """VitGAN's dim_head = 170 gives K = 1020 / N = 3060: rows are only 4-/8-byte aligned."""
    M, N, K_ = 96, 3060, 1020
    x, w = _mk((M, K_), dt, cuda, 1), _mk((N, K_), dt, cuda, 2)
    y = torch.empty(M, N, dtype=torch.float32, device=cuda)
    K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_)
    assert _rel(y, x.double() @ w.double().T) < 2e-5
    dy = _mk((M, N), dt, cuda, 3)
    wg = torch.zeros(N, K_, dtype=torch.float32, device=cuda)
    K.gemm(dy, x, wg, N, K_, M, ldx=N, ldw=K_, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS, flags=K.F_ACCUM_OUT)
    assert _rel(wg, dy.double().T @ x.double()) < 2e-5
    # head blocks at 340-byte offsets (170 bf16): q_h . k_h^T
    H, dh, T = 6, 170, 16
    qkv = _mk((2, T, 3 * H * dh), dt, cuda, 4)
    S = torch.empty(2 * H, T, T, dtype=torch.float32, device=cuda)
    D3 = 3 * H * dh
    K.gemm(qkv, qkv.view(-1)[H * dh:], S, T, T, dh, ldx=D3, ldw=D3, batch=2 * H, batch_inner=H, xb=(T * D3, dh),
           wb=(T * D3, dh), yb=(H * T * T, T * T), y_map=(0, 0, T))
    q = qkv[:, :, :H * dh].reshape(2, T, H, dh).double()
    k = qkv[:, :, H * dh:2 * H * dh].reshape(2, T, H, dh).double()
    assert _rel(S.view(2, H, T, T), torch.einsum("bthd,bshd->bhts", q, k)) < 2e-5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,ldx,ldw", [(3060, 1024, 3064, 1024), (1024, 1020, 1024, 1024), (3060, 1020, 3064, 1024), (3060, 1024, 3060, 1024)])
def test_padded_reduction_major_operands(cuda, dt, M, N, ldx, ldw):
    """The VitGAN blocks' qkv / w_out weight gradients (vitgan.py:62-97: 6 heads x 170): un-padded fp32 gradients [3060, 1024] /
    [1024, 1020] from operands whose ROWS are padded to 3064 / 1024.  The LDS-DMA kernel reads the last 8-column chunk in full
    (pad columns hold garbage here: NaN, to prove they reach no stored output) when FFVC_TT_PAD=1 (off by default: every such launch
    then takes the register-staged kernel, as without room in the row stride).  Either way the result must agree with fp64."""
    rows = 512
    dy = _mk((rows, ldx), dt, cuda, 1)
    x = _mk((rows, ldw), dt, cuda, 2)
    if ldx > M:
        dy[:, M:] = float("nan")
    if ldw > N:
        x[:, N:] = float("nan")
    wg = torch.ones(M, N, dtype=torch.float32, device=cuda)
    K.gemm(dy, x, wg, M, N, rows, ldx=ldx, ldw=ldw, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS, flags=K.F_ACCUM_OUT)
    ref = dy[:, :M].double().T @ x[:, :N].double() + 1.0
    assert torch.isfinite(wg).all()
    assert _rel(wg, ref) < 2e-5


def test_padded_reduction_major_operands_on_the_lds_dma_kernel(cuda):
    """FFVC_TT_PAD=1 (opt-in, read once per process -> a child process): the same launches on the LDS-DMA kernel, which reads the last
    8-column chunk of a padded row in full — 10 repetitions x 2 dtypes x 4 shapes with NaN in the pad columns and a NaN-poisoned
    allocator cache (tools/r6/tt_pad_stress.py), every result against fp64."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVC_TT_PAD="1", STRESS_REPS="10")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "r6", "tt_pad_stress.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "mismatches: 0" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("dt", DT)
def test_splitk_slabs(cuda, dt):
    M, N, K_ = 256, 384, 4096
    xt, wt = _mk((K_, M), dt, cuda, 1), _mk((K_, N), dt, cuda, 2)
    out = torch.ones(M, N, dtype=torch.float32, device=cuda)
    K.gemm_splitk_accumulate(xt, wt, out, M, N, K_, 5, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS)
    assert _rel(out, xt.double().T @ wt.double() + 1.0) < 2e-5
    out2 = torch.ones(M, N, dtype=torch.float32, device=cuda)
    K.gemm_splitk_accumulate(xt, wt, out2, M, N, K_, 1, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS)
    assert _rel(out2, xt.double().T @ wt.double() + 1.0) < 2e-5


@pytest.mark.parametrize("W,Cin,Cout,ups", [(64, 128, 128, False), (128, 64, 256, False), (256, 128, 128, False),
                                           (64, 128, 128, True), (128, 256, 128, True), (512, 64, 128, False),
                                           (512, 128, 128, True), (768, 64, 128, False)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_conv_row_tile_kernel(cuda, W, Cin, Cout, ups, dt):
    """Haloed row-tile conv fast path (forced) vs F.conv2d, incl. image borders and the fused 2x upsample."""
    B, H = (1, min(W, 128)) if W >= 128 else (2, 64)        # wide images (W > 256): a tile is a 256-pixel row segment
    Hin, Win = (H // 2, W // 2) if ups else (H, W)
    x = _mk((B, Hin, Win, Cin), dt, cuda, 1)
    w = _mk((Cout, 3, 3, Cin), dt, cuda, 2, 0.05)
    b = _mk((Cout,), torch.float32, cuda, 3)
    res = _mk((B, H, W, Cout), dt, cuda, 4)
    y = torch.empty(B, H, W, Cout, dtype=dt, device=cuda)
    K.set_option("conv_row", 2)
    try:
        K.gemm(x, w, y, B * H * W, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, bias=b, residual=res,
               conv=(H, W, Cin), flags=K.F_UPSAMPLE2X if ups else 0)
    finally:
        K.set_option("conv_row", 1)
    xn = x.double().permute(0, 3, 1, 2)
    if ups:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xn, w.double().permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    assert _rel(y, ref) < LOTOL[dt]


@pytest.mark.parametrize("kind", ["conv_row", "conv_generic", "linear_residual"])
def test_gn_sums_from_epilogue(cuda, kind):
    """FFVC_F_GN_SUMS: per-(image, group) sum / sum of squares of the STORED output, accumulated by the epilogue."""
    torch.manual_seed(0)
    B, G = 3, 32
    if kind == "linear_residual":
        HW, Kd, C = 256, 64, 128
        x = torch.randn(B * HW, Kd, device=cuda).bfloat16()
        w = (torch.randn(C, Kd, device=cuda) * 0.1).bfloat16()
        res = torch.randn(B * HW, C, device=cuda).bfloat16()
        y = torch.empty(B * HW, C, device=cuda, dtype=torch.bfloat16)
        sums = K.gn_sums_buffer(B, G, cuda)
        K.gemm(x, w, y, B * HW, C, Kd, ldx=Kd, ldw=Kd, residual=res, gn_sums=(sums, HW, C // G))
    else:
        H, Cin, C = (64, 128, 128) if kind == "conv_row" else (16, 64, 256)
        HW = H * H
        x = torch.randn(B, H, H, Cin, device=cuda).bfloat16()
        w = (torch.randn(C, 3, 3, Cin, device=cuda) * 0.05).bfloat16()
        bias = torch.randn(C, device=cuda)
        y = torch.empty(B, H, H, C, device=cuda, dtype=torch.bfloat16)
        sums = K.gn_sums_buffer(B, G, cuda)
        K.gemm(x, w, y, B * HW, C, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, bias=bias, conv=(H, H, Cin),
               gn_sums=(sums, HW, C // G))
    yd = y.double().view(B, HW, G, C // G)
    ref = torch.stack([yd.sum(dim=(1, 3)), (yd * yd).sum(dim=(1, 3))], dim=-1)
    # the kernel sums the fp32 values BEFORE the bf16 rounding of y: per element the difference is <= 2^-9 |x|, random sign
    n = HW * (C // G)
    noise = 4.0 * (n ** 0.5) * 2.0 ** -9 * (ref[..., 1] / n).sqrt()
    assert ((sums[..., 0] - ref[..., 0]).abs() <= noise + 1e-3).all()
    assert ((sums[..., 1] - ref[..., 1]).abs() / ref[..., 1]).max().item() < 1e-3
    # and GroupNorm from those sums == GroupNorm with its own statistics pass
    gamma, beta = torch.randn(C, device=cuda), torch.randn(C, device=cuda)
    yv = y.view(B, -1, 1, C) if kind == "linear_residual" else y
    a = K.groupnorm_fwd(yv, gamma, beta, swish=True)
    b = K.groupnorm_fwd(yv, gamma, beta, swish=True, sums=sums)
    assert _rel(b[0], a[0]) < 1e-2 and _rel(b[1], a[1]) < 2e-3 and _rel(b[2], a[2]) < 2e-3


@pytest.mark.parametrize("dt,Kred,sk", [(torch.bfloat16, 65536, 48), (torch.float32, 65536, 48), (torch.bfloat16, 1000, 7),
                                        (torch.float16, 65536, 48)])
def test_splitk_slabs_all_written(cuda, dt, Kred, sk):
    """Regression: ceil(K / ceil(ksteps/split)*BK) can be < split_k (48 -> 47 at K=65536), and the
    trailing slab used to stay unwritten while ffvc_slab_reduce summed it.  Poison the caching allocator with NaNs so
    a recycled, unwritten slab cannot hide behind freshly zeroed VRAM."""
    M, N = 256, 128
    x, w = _mk((M, Kred), dt, cuda, 1, 0.1), _mk((N, Kred), dt, cuda, 2, 0.1)
    out = _mk((M, N), torch.float32, cuda, 3)
    ref = out.double() + x.double() @ w.double().T
    poison = torch.full((sk, M, N), float("nan"), dtype=torch.float32, device=cuda)
    del poison                                   # the next torch.empty(sk, M, N) recycles this block
    K.gemm_splitk_accumulate(x, w, out, M, N, Kred, sk, ldx=Kred, ldw=Kred)
    assert torch.isfinite(out).all(), "an unwritten split-K slab leaked into the reduction"
    assert _rel(out, ref) < 1e-4           # fp32 accumulation order over up to 65536 terms


# ----------------------------------------------------------------------------- every LDS-DMA tile configuration, forced
@pytest.mark.parametrize("tile", [128, 256, 512])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_forced_tiles_nt_and_conv(cuda, tile, dt):
    """Each tile configuration of the LDS-DMA path (512 = the 8-phase 256x256 kernel) on interior and ragged shapes,
    K tails, fused epilogues, split-K slabs, batches and the generic implicit-GEMM conv; every launch is repeated and
    must reproduce bit for bit (a staging race shows up as run-to-run differences long before it shows in a tolerance)."""
    K.set_option("gemm2_tile", tile)
    K.set_option("gemm8", 1)            # 256x256 launches take the 8-phase kernel wherever it is eligible
    try:
        for (M, N, K_) in [(512, 512, 64), (512, 256, 128), (768, 512, 1024), (1000, 520, 328), (256, 256, 4096),
                           (2048, 1024, 192), (300, 264, 72)]:
            x, w = _mk((M, K_), dt, cuda, 1, 0.5), _mk((N, K_), dt, cuda, 2, 0.5)
            b = _mk((N,), torch.float32, cuda, 3)
            res = _mk((M, N), torch.float32, cuda, 4)
            ref = x.double() @ w.double().T + b.double() + res.double()
            outs = []
            for _ in range(3):
                y = torch.empty(M, N, dtype=torch.float32, device=cuda)
                K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_, bias=b, residual=res)
                outs.append(y)
            assert _rel(outs[0], ref) < 2e-5, (M, N, K_)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (M, N, K_)
            # 16-bit output + GELU + pre-activation
            y16 = torch.empty(M, N, dtype=dt, device=cuda)
            aux = torch.empty(M, N, dtype=dt, device=cuda)
            K.gemm(x, w, y16, M, N, K_, ldx=K_, ldw=K_, bias=b, aux=aux, ldaux=N, act=K.ACT_GELU, flags=K.F_WRITE_PREACT)
            pre = x.double() @ w.double().T + b.double()
            assert _rel(aux, pre) < LOTOL[dt] and _rel(y16, F.gelu(pre)) < LOTOL[dt], (M, N, K_)
        # split-K slabs + batch
        M, N, K_ = 512, 512, 2048
        x, w = _mk((M, K_), dt, cuda, 5, 0.3), _mk((N, K_), dt, cuda, 6, 0.3)
        out = torch.zeros(M, N, dtype=torch.float32, device=cuda)
        K.gemm_splitk_accumulate(x, w, out, M, N, K_, 4, ldx=K_, ldw=K_)
        assert _rel(out, x.double() @ w.double().T) < 2e-5
        xb, wb = _mk((3, 256, 320), dt, cuda, 7), _mk((3, 512, 320), dt, cuda, 8)
        yb = torch.empty(3, 256, 512, dtype=torch.float32, device=cuda)
        K.gemm(xb, wb, yb, 256, 512, 320, ldx=320, ldw=320, batch=3, xb=(256 * 320, 0), wb=(512 * 320, 0), yb=(256 * 512, 0))
        assert _rel(yb, torch.einsum("bmk,bnk->bmn", xb.double(), wb.double())) < 2e-5
        # generic implicit-GEMM conv (row-tile kernel off), with and without the fused upsample
        K.set_option("conv_row", 0)
        for (B, Hin, Cin, Cout, ups) in [(2, 16, 128, 256, False), (1, 16, 64, 512, True), (3, 12, 128, 320, False)]:
            H = 2 * Hin if ups else Hin
            xc = _mk((B, Hin, Hin, Cin), dt, cuda, 9)
            wc = _mk((Cout, 3, 3, Cin), dt, cuda, 10, 0.05)
            bc = _mk((Cout,), torch.float32, cuda, 11)
            yc = torch.empty(B, H, H, Cout, dtype=dt, device=cuda)
            K.gemm(xc, wc, yc, B * H * H, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, bias=bc, conv=(H, H, Cin),
                   flags=K.F_UPSAMPLE2X if ups else 0)
            xn = xc.double().permute(0, 3, 1, 2)
            if ups:
                xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
            refc = F.conv2d(xn, wc.double().permute(0, 3, 1, 2), bc.double(), padding=1).permute(0, 2, 3, 1)
            assert _rel(yc, refc) < LOTOL[dt], (B, Hin, Cin, Cout, ups)
    finally:
        K.set_option("gemm2_tile", 1)
        K.set_option("gemm8", 0)
        K.set_option("conv_row", 1)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_forced_gemm8_with_auto_split_k_shape(cuda, dt):
    """A 256x256-tile launch with 24..100 tiles and K >= 2048 gets its K split inside the launch (ticket + combine, gemm2_kernel
    only).  With the 8-phase kernel forced the same launch must not end up on a kernel without the combine (every K slice would
    store its partial tile straight into y): 6400x768x3072 = 75 tiles, split 2."""
    M, N, K_ = 6400, 768, 3072
    x, w = _mk((M, K_), dt, cuda, 21, 0.3), _mk((N, K_), dt, cuda, 22, 0.3)
    b = _mk((N,), torch.float32, cuda, 23)
    ref = x.double() @ w.double().T + b.double()
    K.set_option("gemm2_tile", 512)
    K.set_option("gemm8", 1)
    try:
        outs = []
        for _ in range(2):
            y = torch.empty(M, N, dtype=torch.float32, device=cuda)
            K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_, bias=b)
            outs.append(y)
    finally:
        K.set_option("gemm2_tile", 1)
        K.set_option("gemm8", 0)
    assert _rel(outs[0], ref) < 2e-5
    assert torch.equal(outs[0], outs[1])


# ----------------------------------------------------------------------------- fp8 path (cfg5)
def _f8_ref(t8, fmt):
    """uint8 fp8 bytes -> fp32 values through torch's own OCP float8 dtypes (the independent decoder)."""
    return t8.view(torch.float8_e4m3fn if fmt == K.E4M3 else torch.float8_e5m2).float()


@pytest.mark.parametrize("fmt", [0, 1])
def test_fp8_quant_matches_torch_float8(cuda, fmt):
    """ffvc_fp8_quant = saturating round-to-nearest-even to OCP e4m3fn / e5m2 of x * scale, scale = fmt_max / (amax * margin)."""
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(64, 264, generator=g) * 3).to(torch.float16).cuda()
    sc = K.Fp8Scale(fmt, x.device)
    q = K.fp8_quant(x, sc)
    st = sc.state.cpu()
    fmax = 448.0 if fmt == 0 else 57344.0
    assert abs(st[0].item() - fmax / (x.float().abs().max().item() * K.FP8_MARGIN)) < 1e-3 * st[0].item()
    assert abs(st[0].item() * st[2].item() - 1) < 1e-6 and st[1].item() == x.float().abs().max().item()
    ref = (x.float() * st[0]).clamp(-fmax, fmax).to(torch.float8_e4m3fn if fmt == 0 else torch.float8_e5m2)
    assert torch.equal(q.cpu(), ref.view(torch.uint8).cpu())
    sat = K.fp8_quant(x * 100, sc)                       # stale (delayed) scale -> saturates at the format maximum, no NaN / inf
    assert torch.isfinite(_f8_ref(sat, fmt)).all() and _f8_ref(sat, fmt).abs().max().item() == fmax
    K.fp8_next_scale(sc)                                 # marks the stream; the update itself is enqueued by the flush (or the next producer)
    K.fp8_flush_updates()
    assert abs(sc.state[0].item() - fmax / ((x * 100).float().abs().max().item() * K.FP8_MARGIN)) < 1e-2 * sc.state[0].item()
    xn = x.clone()
    xn[3, 7] = float("nan")                              # a NaN must stay a NaN (not be clamped to -max and hidden)
    qn = _f8_ref(K.fp8_quant(xn, K.Fp8Scale(fmt, x.device)), fmt)
    assert torch.isnan(qn[3, 7]) and torch.isnan(qn).sum().item() == 1


@pytest.mark.parametrize("ldt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("xfmt", [0, 1])
@pytest.mark.parametrize("M,N,Kd,tile", [(512, 512, 1024, 0), (300, 264, 80, 0), (1024, 256, 256, 512), (520, 384, 272, 256),
                                         (256, 128, 128, 128), (16448, 1024, 1024, 0)])
def test_fp8_gemm_matches_dequantised_reference(cuda, M, N, Kd, tile, xfmt, ldt, monkeypatch):
    """ffvc_gemm_fp8 (v_mfma_f32_32x32x64_f8f6f4, fp32 accumulate) against fp64 math on the SAME fp8 values decoded by
    torch's float8 dtypes, for every tile configuration, ragged M / N, K tails, e4m3 and e5m2 activations."""
    if tile:
        monkeypatch.setenv("FFVC_FP8_BM", str(tile))      # read once per process: only the first forced value sticks,
    g = torch.Generator().manual_seed(7)                  # the heuristic covers the rest
    x = torch.randn(M, Kd, generator=g).to(ldt).cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).cuda()
    sx, sw = K.Fp8Scale(xfmt, x.device), K.Fp8Scale(K.E4M3, x.device)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    K.gemm_fp8(x8, w8, y, M, N, Kd, sx, sw, lo_dtype=ldt, bias=bias, residual=res)
    ref = (_f8_ref(x8, xfmt).double() @ _f8_ref(w8, K.E4M3).double().t()) * (sx.state[2].double() * sw.state[2].double()) + \
        bias.double() + res.double()
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 1e-4      # fp32 accumulation order / MFMA adder tree
    # close to the unquantised product as well (3 mantissa bits on the activation, per-tensor scales)
    full = x.double() @ w.double().t() + bias.double() + res.double()
    assert ((y.double() - full).pow(2).mean().sqrt() / full.pow(2).mean().sqrt()).item() < (0.06 if xfmt == 0 else 0.12)
    # 16-bit output + GELU + pre-activation write, then the activation-gradient epilogue on the stored pre-activation
    h, pre = torch.empty(M, N, dtype=ldt, device=x.device), torch.empty(M, N, dtype=ldt, device=x.device)
    K.gemm_fp8(x8, w8, h, M, N, Kd, sx, sw, lo_dtype=ldt, bias=bias, act=K.ACT_GELU, aux=pre, ldaux=N, flags=K.F_WRITE_PREACT)
    pref = ref - res.double()
    tol = 2e-3 if ldt == torch.float16 else 1.6e-2
    assert ((pre.double() - pref).abs().max() / pref.abs().max()).item() < tol
    assert ((h.double() - torch.nn.functional.gelu(pref)).abs().max() / pref.abs().max()).item() < tol
    dh = torch.empty(M, N, dtype=ldt, device=x.device)
    K.gemm_fp8(x8, w8, dh, M, N, Kd, sx, sw, lo_dtype=ldt, act=K.ACT_GELU, aux=pre, ldaux=N, flags=K.F_MUL_ACT_GRAD)
    p = pre.double()
    gg = 0.5 * (1 + torch.erf(p / 2 ** 0.5)) + p * torch.exp(-0.5 * p * p) / (2 * torch.pi) ** 0.5
    want = (ref - res.double() - bias.double()) * gg
    assert ((dh.double() - want).abs().max() / want.abs().max()).item() < tol


def test_fp8_gemm_rejects_bad_shapes(cuda):
    from feed_forward_vqgan_clip_amd._lib import FFVCError
    x8 = torch.zeros(64, 72, dtype=torch.uint8, device=cuda)
    w8 = torch.zeros(64, 72, dtype=torch.uint8, device=cuda)
    y = torch.empty(64, 64, device=cuda)
    sc = K.Fp8Scale(K.E4M3, x8.device)
    with pytest.raises(FFVCError, match="multiples of 16"):
        K.gemm_fp8(x8, w8, y, 64, 64, 72, sc, sc, lo_dtype=torch.float16)


@pytest.mark.parametrize("ldt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu", "quickgelu"])
@pytest.mark.parametrize("M,N,Kd", [(1024, 1024, 256), (200, 72, 40), (512, 256, 128)])
def test_gemm_actgrad_storage(cuda, M, N, Kd, act, ldt):
    """FFVC_F_AUX_ACTGRAD: the forward leaves act'(pre) in aux (specialised epilogue on 256x256 tiles, pre-activation +
    conversion pass elsewhere) and the backward epilogue multiplies by it — same numbers as the textbook pair
    (pre-activation stored, derivative evaluated in the backward epilogue)."""
    code = K.ACT_GELU if act == "gelu" else K.ACT_QUICKGELU
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, Kd, generator=g).to(ldt).cuda()
    w = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(ldt).cuda()
    b = torch.randn(N, generator=g).cuda()
    dy = torch.randn(M, Kd, generator=g).to(ldt).cuda()
    wt = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(ldt).cuda()
    h0, pre = torch.empty(M, N, dtype=ldt, device=cuda), torch.empty(M, N, dtype=ldt, device=cuda)
    h1, gd = torch.empty_like(h0), torch.empty_like(h0)
    K.gemm(x, w, h0, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=code, aux=pre, ldaux=N, flags=K.F_WRITE_PREACT)
    K.gemm(x, w, h1, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=code, aux=gd, ldaux=N, flags=K.F_WRITE_PREACT | K.F_AUX_ACTGRAD)
    assert torch.equal(h0, h1)
    p = (x.double() @ w.double().t() + b.double())
    if act == "gelu":
        want = 0.5 * (1 + torch.erf(p / 2 ** 0.5)) + p * torch.exp(-0.5 * p * p) / (2 * torch.pi) ** 0.5
    else:
        s = torch.sigmoid(1.702 * p)
        want = s * (1 + 1.702 * p * (1 - s))
    tol = 2e-3 if ldt == torch.float16 else 1.6e-2
    assert (gd.double() - want).abs().max().item() < tol * max(1.0, want.abs().max().item()) + \
        (2e-2 if ldt == torch.bfloat16 else 3e-3)          # conversion-pass path differentiates the ROUNDED pre-activation
    d0, d1 = torch.empty_like(h0), torch.empty_like(h0)
    K.gemm(dy, wt, d0, M, N, Kd, ldx=Kd, ldw=Kd, act=code, aux=pre, ldaux=N, flags=K.F_MUL_ACT_GRAD)
    K.gemm(dy, wt, d1, M, N, Kd, ldx=Kd, ldw=Kd, act=code, aux=gd, ldaux=N, flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD)
    ref = (dy.double() @ wt.double().t()) * want
    scale = ref.abs().max().item()
    assert (d1.double() - ref).abs().max().item() < 3 * tol * scale and (d0.double() - ref).abs().max().item() < 3 * tol * scale


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K_", [(512, 1024, 4096), (512, 1024, 1024), (300, 200, 2048), (616, 768, 9216), (64, 128, 8192),
                                    (6400, 768, 3072), (4928, 512, 6144), (3200, 768, 2048)])      # the last three: 256x256 tiles, 2-4 slices
def test_underfilled_grid_in_kernel_split_k(cuda, dt, M, N, K_):
    """Few output tiles x long reduction (VitGAN / x-transformer linears at a per-GPU batch of 16-32 samples): the launch is split
    along K inside the kernel and the last workgroup per tile runs the fused epilogue on the summed tile.  Same answers as the
    unsplit launch (bias + GELU + pre-activation write + fp32 residual), repeatable bit for bit, also from two streams at once."""
    x, w = _mk((M, K_), dt, cuda, 1, 0.5), _mk((N, K_), dt, cuda, 2, 0.05)
    b = _mk((N,), torch.float32, cuda, 3)
    res = _mk((M, N), torch.float32, cuda, 4)
    pre = x.double() @ w.double().T + b.double()
    ref = F.gelu(pre) + res.double()

    def run():
        y = torch.empty(M, N, dtype=torch.float32, device=cuda)
        aux = torch.empty(M, N, dtype=dt, device=cuda)
        K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_, bias=b, residual=res, aux=aux, ldaux=N, act=K.ACT_GELU, flags=K.F_WRITE_PREACT)
        return y, aux

    y, aux = run()
    assert _rel(aux, pre) < LOTOL[dt]
    assert _rel(y, ref) < 3e-5
    y2, aux2 = run()
    assert torch.equal(y, y2) and torch.equal(aux, aux2)           # slice order, not arrival order
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(run())
    torch.cuda.synchronize()
    for yy, aa in outs:
        assert torch.equal(yy, y) and torch.equal(aa, aux)
    # plain 16-bit output without any epilogue (the lean kernel class)
    yl = torch.empty(M, N, dtype=dt, device=cuda)
    K.gemm(x, w, yl, M, N, K_, ldx=K_, ldw=K_)
    assert _rel(yl, x.double() @ w.double().T) < LOTOL[dt]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K_", [(16448, 1024, 2048), (25600, 768, 2304), (16448, 1024, 4096), (66000, 256, 2112), (16448, 1024, 1024)])
def test_partly_filled_last_round_is_split_along_k(cuda, dt, M, N, K_):
    """A few 256x256 tiles more than a multiple of the CU count (260 / 300 on 256 CUs): the tiles of the last round are cut along
    K inside the launch (tail mode of the in-kernel split-K).  Same answers as fp64 math through the fused epilogues, repeatable
    bit for bit, and identical to the unsplit launch where no slice boundary changes the fp32 summation order."""
    x, w = _mk((M, K_), dt, cuda, 1, 0.5), _mk((N, K_), dt, cuda, 2, 0.05)
    b = _mk((N,), torch.float32, cuda, 3)
    res = _mk((M, N), torch.float32, cuda, 4)
    y = torch.empty(M, N, dtype=torch.float32, device=cuda)
    K.gemm(x, w, y, M, N, K_, ldx=K_, ldw=K_, bias=b, residual=res)
    ref = x.double() @ w.double().T + b.double() + res.double()
    assert _rel(y, ref) < 3e-5
    y2 = torch.empty_like(y)
    K.gemm(x, w, y2, M, N, K_, ldx=K_, ldw=K_, bias=b, residual=res)
    assert torch.equal(y, y2)
    h = torch.empty(M, N, dtype=dt, device=cuda)
    aux = torch.empty(M, N, dtype=dt, device=cuda)
    K.gemm(x, w, h, M, N, K_, ldx=K_, ldw=K_, bias=b, aux=aux, ldaux=N, act=K.ACT_GELU, flags=K.F_WRITE_PREACT)
    pre = x.double() @ w.double().T + b.double()
    assert _rel(aux, pre) < LOTOL[dt] and _rel(h, F.gelu(pre)) < LOTOL[dt]
    # the tail rows (last tile row) against the head rows of the same launch: both must be right
    assert _rel(y[-64:], ref[-64:]) < 3e-5 and _rel(y[:256], ref[:256]) < 3e-5


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,Kred,sk", [(1024, 4096, 16384, 4), (4096, 1024, 16384, 4), (1024, 1024, 8192, 3), (2048, 1024, 32768, 8)])
def test_weight_gradient_split_k_combined_inside_the_launch(cuda, dt, M, N, Kred, sk):
    """FFVC_F_SPLITK_INKERNEL on the 256x256-tile weight-gradient kernel (both operands reduction-major): out += dy^T x with the K
    slices combined by the last workgroup per tile — against fp64, against the slab + reduce form, and bit-repeatable."""
    dy, x = _mk((Kred, M), dt, cuda, 1, 0.5), _mk((Kred, N), dt, cuda, 2, 0.5)
    base = _mk((M, N), torch.float32, cuda, 3)
    ref = base.double() + dy.double().T @ x.double()

    def run(in_kernel):
        out = base.clone()
        K.gemm_splitk_accumulate(dy, x, out, M, N, Kred, sk, in_kernel=in_kernel, ldx=M, ldw=N, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS)
        return out

    a, b = run(True), run(False)
    assert _rel(a, ref) < 2e-5 and _rel(b, ref) < 2e-5
    assert torch.equal(a, run(True))
    # a shape that does not take the 256x256 kernel refuses the flag instead of silently splitting some other way
    with pytest.raises(RuntimeError):
        small = torch.zeros(128, 128, dtype=torch.float32, device=cuda)
        K.gemm_splitk_accumulate(dy[:512, :128].contiguous(), x[:512, :128].contiguous(), small, 128, 128, 512, 2, in_kernel=True, ldx=128,
                                 ldw=128, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS)


@pytest.mark.parametrize("ldt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("act", ["gelu", "quickgelu"])
def test_fp8_gemm_writes_the_next_operand_itself(cuda, ldt, act):
    """fp8 OUTPUT of the two frozen-MLP kinds (include/ffvc.h y8_state): the activation forward leaves the hidden activation as
    e4m3 bytes (+ act'(pre) in aux), the aux-multiply backward leaves the hidden gradient as e5m2 bytes — each equal to what
    the 16-bit output followed by ffvc_fp8_quant would hold up to the double rounding of that path (<= one fp8 step on a small
    fraction of the elements), with the running amax updated the same way."""
    code = K.ACT_GELU if act == "gelu" else K.ACT_QUICKGELU
    M, N, Kd = 1024, 1024, 256
    g = torch.Generator().manual_seed(13)
    x = torch.randn(M, Kd, generator=g).to(ldt).cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    sx, sw = K.Fp8Scale(K.E4M3, x.device), K.Fp8Scale(K.E4M3, x.device)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    flags = K.F_WRITE_PREACT | K.F_AUX_ACTGRAD
    # reference path: 16-bit output, then the separate quantisation pass (which also initialises the output scale)
    h, dact = torch.empty(M, N, dtype=ldt, device=x.device), torch.empty(M, N, dtype=ldt, device=x.device)
    K.gemm_fp8(x8, w8, h, M, N, Kd, sx, sw, lo_dtype=ldt, bias=bias, act=code, aux=dact, ldaux=N, flags=flags)
    so = K.Fp8Scale(K.E4M3, x.device)
    h8_ref = K.fp8_quant(h, so)
    amax_ref = so.state[1].item()
    so.state[1] = 0.0
    # fused path
    h8, dact2 = torch.empty(M, N, dtype=torch.uint8, device=x.device), torch.empty(M, N, dtype=ldt, device=x.device)
    K.gemm_fp8(x8, w8, h8, M, N, Kd, sx, sw, lo_dtype=ldt, bias=bias, act=code, aux=dact2, ldaux=N, flags=flags, out_scale=so)
    # (the reference launch may have taken a kernel without the specialised epilogue: pre-activation stored in 16 bits, converted
    #  to act' by a second pass — same quantity, one more rounding)
    assert _rel(dact2, dact.double()) < (2e-3 if ldt == torch.float16 else 1.6e-2)
    a, b = _f8_ref(h8, K.E4M3), _f8_ref(h8_ref, K.E4M3)
    differ = (a != b).float().mean().item()
    assert differ < 0.05 and (a - b).abs().max().item() <= 0.13 * b.abs().max().item()      # one e4m3 step at the top binade = 1/8
    assert abs(so.state[1].item() - amax_ref) <= 2e-2 * amax_ref
    # backward kind: dy (e5m2) x W^T * act'(pre) -> e5m2
    dy = (torch.randn(M, Kd, generator=g) * 1e-3).to(ldt).cuda()
    sg = K.Fp8Scale(K.E5M2, x.device)
    dy8 = K.fp8_quant(dy, sg)
    dh = torch.empty(M, N, dtype=ldt, device=x.device)
    K.gemm_fp8(dy8, w8, dh, M, N, Kd, sg, sw, lo_dtype=ldt, act=code, aux=dact, ldaux=N, flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD)
    sd = K.Fp8Scale(K.E5M2, x.device)
    dh8_ref = K.fp8_quant(dh, sd)
    amax_ref = sd.state[1].item()
    sd.state[1] = 0.0
    dh8 = torch.empty(M, N, dtype=torch.uint8, device=x.device)
    K.gemm_fp8(dy8, w8, dh8, M, N, Kd, sg, sw, lo_dtype=ldt, act=code, aux=dact, ldaux=N, flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD,
               out_scale=sd)
    a, b = _f8_ref(dh8, K.E5M2), _f8_ref(dh8_ref, K.E5M2)
    assert (a != b).float().mean().item() < 0.05 and (a - b).abs().max().item() <= 0.26 * b.abs().max().item()
    assert abs(sd.state[1].item() - amax_ref) <= 2e-2 * amax_ref
    # anything but those two kinds refuses an fp8 output
    from feed_forward_vqgan_clip_amd._lib import FFVCError
    with pytest.raises(FFVCError, match="fp8 output"):
        K.gemm_fp8(x8, w8, h8, M, N, Kd, sx, sw, lo_dtype=ldt, out_scale=so)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Hin,Cin,Cout,ups,res,gn", [(8, 128, 128, 128, False, True, True), (16, 64, 256, 128, False, False, False),
                                                      (4, 64, 128, 256, True, False, True), (2, 256, 128, 128, False, False, False)])
def test_fp8_conv3x3_forward_and_dgrad(cuda, dt, B, Hin, Cin, Cout, ups, res, gn):
    """The frozen decoder's 3x3 convolution on the fp8 row kernel (e4m3 x e4m3 forward with bias / residual / GroupNorm moments /
    fused nearest-2x upsample, e5m2 x e4m3 dgrad) against fp64 math on the DECODED fp8 operands (torch's own float8 dtypes): the
    kernel's arithmetic is exact up to fp32 accumulation; and against the 16-bit convolution of the unquantised operands within the
    quantisation budget."""
    from feed_forward_vqgan_clip_amd import ops
    H = 2 * Hin if ups else Hin
    g = torch.Generator().manual_seed(B + Cin + Cout)
    x = (torch.randn(B, Hin, Hin, Cin, generator=g) * 0.7).to(dt).cuda()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, generator=g) * 0.1
    r = (torch.randn(B, H, H, Cout, generator=g) * 0.5).to(dt).cuda() if res else None
    P8, P16 = ops.ConvWeights(w, b, dt, fp8=True), ops.ConvWeights(w, b, dt)
    assert P8.fp8 is not None and K.conv_fp8_ok(B, H, H, Cin, Cout)
    xg = x.clone().requires_grad_(True)
    y8 = ops.conv3x3(xg, P8, residual=r, upsample=ups, gn=gn)
    sums8 = getattr(y8, "_ffvc_gn", None)
    y16 = ops.conv3x3(x, P16, residual=r, upsample=ups)
    # fp64 reference on the decoded fp8 bytes
    f = P8.fp8
    sx, sw = f["x"].state.cpu(), f["w"].state.cpu()
    x8 = _f8_ref(K.fp8_quant(x, f["x"]).cpu(), K.E4M3).double() * float(sx[2])
    w8 = _f8_ref(f["w8"].cpu(), K.E4M3).double().view(Cout, 3, 3, Cin) * float(sw[2])
    xn = x8.permute(0, 3, 1, 2)
    if ups:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xn, w8.permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1)
    if res:
        ref = ref + r.double().cpu()
    assert _rel(y8.detach().cpu(), ref) < (2e-3 if dt == torch.float16 else 1e-2)   # storage rounding of y only
    assert _rel(y8.detach(), y16.double()) < 5e-2                                    # e4m3 quantisation of x and w
    if gn:
        assert sums8 is not None
        m = ref.view(B, H * H, 32, Cout // 32)
        want = torch.stack([m.sum((1, 3)), (m * m).sum((1, 3))], -1)
        assert _rel(sums8.cpu(), want) < 2e-3
    gy = (torch.randn(B, H, H, Cout, generator=g) * 1e-2).to(dt).cuda()
    y8.backward(gy)
    x16 = x.clone().requires_grad_(True)
    ops.conv3x3(x16, P16, residual=r, upsample=ups).backward(gy)
    assert _rel(xg.grad, x16.grad.double()) < 8e-2                                   # e5m2 gradients (2 mantissa bits) x e4m3 filter


# ----------------------------------------------------------------------------- producer-side quantisation (round 4)
def _ready_scale(t, fmt):
    """An initialised scale that has seen `t` once, its update folded in (so the next producer quantises with amax(t))."""
    sc = K.Fp8Scale(fmt, t.device)
    K.fp8_quant(t, sc)
    K.fp8_next_scale(sc)
    K.fp8_flush_updates()
    return sc


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,C,use_sums", [(2, 32, 128, False), (3, 16, 256, True), (1, 64, 512, False)])
def test_groupnorm_writes_the_fp8_operand_of_its_convolution(cuda, dt, B, H, C, use_sums):
    """ffvc_groupnorm_fwd_f8 / _bwd_f8: the fp8 bytes written by the normalisation pass ARE fp8_quant of the 16-bit tensor the plain
    kernels write (same scale, same rounding), the running amax is the tensor's, and the 16-bit output may be skipped."""
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(B, H, H, C, generator=g) * 2 + 0.3).to(dt).cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.2 * torch.randn(C, generator=g)).cuda()
    # the moments: given (as the producing GEMM's epilogue hands them over) -> every launch sees the same statistics and the comparison is
    # exact; measured by the statistics pass -> its LDS atomics add in a different order per launch, compare with a flip allowance
    xg = x.double().view(B, H * H, 32, C // 32)
    sums = torch.stack([xg.sum((1, 3)), (xg * xg).sum((1, 3))], -1).contiguous()
    if not use_sums:
        y0, _, _ = K.groupnorm_fwd(x, gamma, beta, 32, 1e-6, True)
        sc0 = _ready_scale(y0, K.E4M3)
        q0 = K.fp8_quant(y0, sc0)
        _, _, _, y8 = K.groupnorm_fwd(x, gamma, beta, 32, 1e-6, True, f8=sc0, f8_only=True)
        assert (y8 != q0).float().mean().item() < 1e-3
    y_ref, mean, rstd = K.groupnorm_fwd(x, gamma, beta, 32, 1e-6, True, sums=sums)
    sc = _ready_scale(y_ref, K.E4M3)
    q_ref = K.fp8_quant(y_ref, sc)
    amax_ref = sc.state[1].item()
    assert amax_ref == y_ref.float().abs().max().item()
    for only in (False, True):
        sc.state[1] = 0.0
        y, m2, r2, y8 = K.groupnorm_fwd(x, gamma, beta, 32, 1e-6, True, sums=sums, f8=sc, f8_only=only)
        assert torch.equal(y8, q_ref) and sc.state[1].item() == amax_ref
        assert torch.equal(m2, mean) and torch.equal(r2, rstd)
        if not only:
            assert torch.equal(y, y_ref)
    # backward: dx (+ dres) as e5m2.  The statistics pass combines its partial sums with LDS atomics (order differs per launch), so
    # the bytes are checked exactly against the 16-bit dx of the SAME launch and with a flip allowance against another launch's
    dy = (torch.randn(B, H, H, C, generator=g) * 1e-3).to(dt).cuda()
    dres = (torch.randn(B, H, H, C, generator=g) * 1e-3).to(dt).cuda()
    for r in (None, dres):
        dx_ref = K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=r, G=32, swish=True)
        sg = _ready_scale(dx_ref, K.E5M2)
        q_ref = K.fp8_quant(dx_ref, sg)
        sg.state[1] = 0.0
        dx, dx8 = K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=r, G=32, swish=True, f8=sg)
        assert sg.state[1].item() == dx.float().abs().max().item()
        twin = K.Fp8Scale(K.E5M2, x.device)
        twin.state.copy_(sg.state)
        twin.ready = True
        assert torch.equal(dx8, K.fp8_quant(dx, twin))
        assert (dx.float() - dx_ref.float()).abs().max().item() <= 2e-2 * dx_ref.float().abs().max().item()
        _, dx8o = K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=r, G=32, swish=True, f8=sg, f8_only=True)
        assert (dx8o != q_ref).float().mean().item() < 2e-3


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,dim,xdt", [(514, 1024, torch.float32), (100, 768, torch.float32), (64, 512, None)])
def test_layernorm_writes_the_fp8_operand_of_its_linear(cuda, dt, rows, dim, xdt):
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(rows, dim, generator=g) * 1.5 + 0.1).to(xdt or dt).cuda()
    gamma = (1 + 0.2 * torch.randn(dim, generator=g)).cuda()
    beta = (0.2 * torch.randn(dim, generator=g)).cuda()
    y_ref, mean, rstd = K.layernorm_fwd(x, gamma, beta, dt)
    sc = _ready_scale(y_ref, K.E4M3)
    q_ref = K.fp8_quant(y_ref, sc)
    amax_ref = sc.state[1].item()
    for only in (False, True):
        sc.state[1] = 0.0
        y, m2, r2, y8 = K.layernorm_fwd(x, gamma, beta, dt, f8=sc, f8_only=only)
        assert torch.equal(y8, q_ref) and sc.state[1].item() == amax_ref
        assert torch.equal(m2, mean) and torch.equal(r2, rstd)
        if not only:
            assert torch.equal(y, y_ref)
    with pytest.raises(TypeError):
        K.layernorm_fwd(x, gamma, beta, torch.float32, f8=sc)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,dim,with_res", [(514, 1024, True), (100, 768, False), (70, 1280, True)])
def test_layernorm_backward_writes_the_fp8_operand_of_the_dgrad(cuda, dt, rows, dim, with_res):
    """ffvc_layernorm_bwd_f8 (frozen layer, fp32 residual stream): dx is the plain kernel's dx bit for bit, and the fp8 bytes are
    fp8_quant of the 16-bit copy the plain kernel writes next to it."""
    g = torch.Generator().manual_seed(14)
    x = (torch.randn(rows, dim, generator=g) * 1.5 + 0.1).cuda()
    gamma = (1 + 0.2 * torch.randn(dim, generator=g)).cuda()
    beta = (0.2 * torch.randn(dim, generator=g)).cuda()
    _, mean, rstd = K.layernorm_fwd(x, gamma, beta, dt)
    dy = (torch.randn(rows, dim, generator=g) * 1e-3).to(dt).cuda()
    dres = (torch.randn(rows, dim, generator=g) * 1e-3).cuda() if with_res else None
    dx_ref, _, _ = K.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres, want_lo=True)
    lo = dx_ref._ffvc_lo
    assert lo.dtype == dt
    sg = _ready_scale(lo, K.E5M2)
    q_ref = K.fp8_quant(lo, sg)
    amax_ref = sg.state[1].item()
    sg.state[1] = 0.0
    dx, dx8 = K.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres, f8=sg)
    assert torch.equal(dx, dx_ref) and torch.equal(dx8, q_ref) and sg.state[1].item() == amax_ref
    with pytest.raises(TypeError):
        K.layernorm_bwd(dy, x, gamma, mean, rstd, want_param_grads=True, f8=sg)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,T,heads,causal", [(4, 257, 4, False), (2, 130, 2, False), (2, 600, 2, True)])
def test_flash_attention_writes_the_fp8_operand_of_out_proj(cuda, dt, B, T, heads, causal):
    g = torch.Generator().manual_seed(15)
    qkv = (torch.randn(B, T, 3 * heads * 64, generator=g) * 0.7).to(dt).cuda()
    o_ref, lse_ref = K.attn_flash_fwd(qkv, heads, 0.125, causal)
    sc = _ready_scale(o_ref, K.E4M3)
    q_ref = K.fp8_quant(o_ref, sc)
    amax_ref = sc.state[1].item()
    sc.state[1] = 0.0
    o, lse, o8 = K.attn_flash_fwd(qkv, heads, 0.125, causal, f8=sc)
    assert torch.equal(o, o_ref) and torch.equal(lse, lse_ref)
    assert torch.equal(o8, q_ref) and sc.state[1].item() == amax_ref


@pytest.mark.parametrize("ydt,rdt", [(torch.float32, torch.float32), (torch.float16, torch.float16), (torch.bfloat16, torch.float32),
                                     (torch.float16, None)])
@pytest.mark.parametrize("xfmt", [0, 1])
@pytest.mark.parametrize("M,N,Kd", [(64, 1024, 4096), (37, 96, 512), (1, 32, 1024), (64, 3072, 1024)])
def test_fp8_skinny_gemm_matches_dequantised_reference(cuda, M, N, Kd, xfmt, ydt, rdt):
    """ffvc_gemm_fp8_skinny (K split across the eight waves of a workgroup, operands straight from global memory) against fp64 math on the
    fp8 values decoded by torch's float8 dtypes: ragged M, every output / residual dtype, e4m3 and e5m2 activations."""
    g = torch.Generator().manual_seed(17)
    x = torch.randn(M, Kd, generator=g).half().cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).cuda()
    sx, sw = K.Fp8Scale(xfmt, x.device), K.Fp8Scale(K.E4M3, x.device)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    bias = torch.randn(N, generator=g).cuda()
    res = None if rdt is None else torch.randn(M, N, generator=g).to(rdt).cuda()
    y = torch.full((M, N), float("nan"), dtype=ydt, device=x.device)
    K.gemm_fp8_skinny(x8, w8, y, M, N, Kd, sx, sw, bias=bias, residual=res)
    ref = (_f8_ref(x8, xfmt).double() @ _f8_ref(w8, K.E4M3).double().t()) * (sx.state[2].double() * sw.state[2].double()) + bias.double()
    if res is not None:
        ref = ref + res.double()
    tol = 1e-4 if ydt == torch.float32 else (2e-3 if ydt == torch.float16 else 1.6e-2)      # fp32: accumulation order / MFMA adder tree
    assert torch.isfinite(y).all() and _rel(y, ref) < tol


def test_fp8_gemm_row_split_uses_the_skinny_tail(cuda, monkeypatch):
    """64 x 257 rows: the 16384 + 64 split with the skinny tail equals the single launch up to fp32 summation order."""
    g = torch.Generator().manual_seed(18)
    M, N, Kd = 16448, 1024, 1024
    x = torch.randn(M, Kd, generator=g).half().cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).cuda()
    sx, sw = K.Fp8Scale(K.E4M3, x.device), K.Fp8Scale(K.E4M3, x.device)
    x8, w8 = K.fp8_quant(x, sx), K.fp8_quant(w, sw, frozen=True)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    outs = {}
    for split in (False, True):
        monkeypatch.setattr(K, "_FP8_ROWSPLIT", split)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        K.gemm_fp8(x8, w8, y, M, N, Kd, sx, sw, lo_dtype=torch.float16, bias=bias, residual=res)
        outs[split] = y
    assert _rel(outs[True], outs[False]) < 1e-4


@pytest.mark.parametrize("dt,ydt,rdt", [(torch.float16, torch.float32, torch.float32), (torch.float16, torch.float16, torch.float16),
                                        (torch.bfloat16, torch.bfloat16, torch.float32), (torch.bfloat16, torch.float32, None)])
@pytest.mark.parametrize("M,N,Kd", [(64, 1024, 4096), (37, 96, 256), (1, 32, 1024), (64, 3072, 1024)])
def test_skinny_gemm_matches_fp64(cuda, M, N, Kd, dt, ydt, rdt):
    """ffvc_gemm_skinny (16-bit operands, K split across the eight waves of a workgroup) against fp64 math on the same 16-bit values."""
    g = torch.Generator().manual_seed(19)
    x = torch.randn(M, Kd, generator=g).to(dt).cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).to(dt).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = None if rdt is None else torch.randn(M, N, generator=g).to(rdt).cuda()
    y = torch.full((M, N), float("nan"), dtype=ydt, device=x.device)
    K.gemm_skinny(x, w, y, M, N, Kd, bias=bias, residual=res)
    ref = x.double() @ w.double().t() + bias.double()
    if res is not None:
        ref = ref + res.double()
    tol = 2e-6 if ydt == torch.float32 else (2e-3 if ydt == torch.float16 else 1.6e-2)
    assert torch.isfinite(y).all() and _rel(y, ref) < tol


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_row_split_uses_the_skinny_tail(cuda, dt, monkeypatch):
    """64 x 257 rows through K.gemm: 16384 rows on the tiled kernel + 64 on ffvc_gemm_skinny == the single launch up to fp32 summation order;
    launches the split does not cover (activation epilogue) are left alone."""
    g = torch.Generator().manual_seed(20)
    M, N, Kd = 16448, 1024, 1024
    x = torch.randn(M, Kd, generator=g).to(dt).cuda()
    w = (torch.randn(N, Kd, generator=g) * 0.05).to(dt).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    outs = {}
    for split in (False, True):
        monkeypatch.setattr(K, "_GEMM_ROWSPLIT", split)
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd, bias=bias, residual=res)
        h = torch.empty(M, N, dtype=dt, device=x.device)
        K.gemm(x, w, h, M, N, Kd, ldx=Kd, ldw=Kd, bias=bias, act=K.ACT_GELU)
        outs[split] = (y, h)
    ref = x.double() @ w.double().t() + bias.double() + res.double()
    assert _rel(outs[True][0], ref) < 2e-6 and _rel(outs[False][0], ref) < 2e-6
    assert torch.equal(outs[True][1], outs[False][1])


def test_fp8_updates_are_batched_and_lazy(cuda):
    """fp8_next_scale only marks a stream; ONE ffvc_fp8_update_many launch folds every marked stream's amax into its scale, and a
    stream nobody flushed is updated by its next producer.  Streams that saw no tensor keep their scale."""
    g = torch.Generator().manual_seed(13)
    a = torch.randn(64, 64, generator=g).half().cuda()
    sa, sb, sidle = K.Fp8Scale(K.E4M3, a.device), K.Fp8Scale(K.E5M2, a.device), K.Fp8Scale(K.E4M3, a.device)
    K.fp8_quant(a, sa), K.fp8_quant(a, sb), K.fp8_quant(a, sidle)
    K.fp8_next_scale(sidle)
    K.fp8_flush_updates()
    idle0 = sidle.state.clone()
    K.fp8_quant(a * 4, sa), K.fp8_quant(a * 8, sb)
    s_a0, s_b0 = sa.state[0].item(), sb.state[0].item()
    K.fp8_next_scale(sa), K.fp8_next_scale(sb)
    assert sa.state[0].item() == s_a0 and sa.pending and sb.pending          # nothing enqueued yet
    K.fp8_flush_updates()
    amax = a.float().abs().max().item()
    assert abs(sa.state[0].item() - 448.0 / (4 * amax * K.FP8_MARGIN)) < 1e-3 * sa.state[0].item()
    assert abs(sb.state[0].item() - 57344.0 / (8 * amax * K.FP8_MARGIN)) < 1e-3 * sb.state[0].item()
    assert sa.state[1].item() == 0 and not sa.pending and not sb.pending
    idle1 = sidle.state.clone()
    assert torch.equal(idle1[[0, 2, 3]], idle0[[0, 2, 3]])                   # no tensor since its last update: scale kept
    sw = K.Fp8Scale(K.E4M3, a.device)                                        # a frozen weight: the flush must never move its scale
    K.fp8_quant(a.float() * 0.05, sw, frozen=True)
    w0 = sw.state.clone()
    K.fp8_next_scale(sa)
    K.fp8_flush_updates()
    assert torch.equal(sw.state, w0) and w0[1].item() == 0
    K.fp8_quant(a * 2, sa)
    K.fp8_next_scale(sa)
    q = K.fp8_quant(a * 2, sa)                                               # no flush: the producer enqueues this stream's update itself
    assert not sa.pending and abs(sa.state[0].item() - 448.0 / (2 * amax * K.FP8_MARGIN)) < 1e-3 * sa.state[0].item()
    ref = (a.float() * 2 * sa.state[0]).clamp(-448, 448).to(torch.float8_e4m3fn)
    assert torch.equal(q.cpu(), ref.view(torch.uint8).cpu())


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("G,M,N,rows", [(4, 1024, 1024, 1024), (3, 1024, 2048, 576), (8, 256, 512, 2048)])
def test_grouped_weight_gradients_in_one_launch(cuda, dt, G, M, N, rows):
    """include/ffvc.h grp_*: the weight gradients of G layers — separate operand allocations, gradients at a constant stride in
    one bucket — in ONE launch of full-K 256x256 tiles, accumulated into what the bucket held (round 5; mlp_mixer_pytorch.py:16-23
    x depth).  Against fp64 math; accumulation order is the only difference."""
    pad = 192                                                  # the layers' gradients are not back to back in the bucket
    stride = M * N + pad
    bucket = _mk((G * stride,), torch.float32, cuda, 7, 0.1)
    before = bucket.clone()
    keep = []                                                  # interleaved decoys: the operands are NOT at a constant stride
    dys, xs = [], []
    for g in range(G):
        dys.append(_mk((rows, M), dt, cuda, 10 + g, 0.5))
        keep.append(torch.empty(1000 * (g + 1) + 8, dtype=dt, device=cuda))
        xs.append(_mk((rows, N), dt, cuda, 30 + g, 0.5))
    K.gemm_grouped_wgrad(dys, xs, bucket, stride, M, N, rows, M, N)
    torch.cuda.synchronize()
    for g in range(G):
        ref = before[g * stride:g * stride + M * N].view(M, N).double() + dys[g].double().T @ xs[g].double()
        got = bucket[g * stride:g * stride + M * N].view(M, N)
        assert _rel(got, ref) < 2e-5, (g, _rel(got, ref))
        assert torch.equal(bucket[g * stride + M * N:(g + 1) * stride], before[g * stride + M * N:(g + 1) * stride])   # pads untouched
    # shapes the 256x256 weight-gradient kernel cannot take fail loudly instead of silently falling back
    from feed_forward_vqgan_clip_amd import _lib
    with pytest.raises(_lib.FFVCError):
        K.gemm_grouped_wgrad([d[:, :200].contiguous() for d in dys[:2]], xs[:2], bucket, stride, 200, N, rows, 200, N)


# ----------------------------------------------------------------------------- round 6: the two-workgroups-per-CU 256x128 ring kernel
@pytest.fixture
def gemm3_forced():
    K.set_option("gemm3", 1)
    yield
    K.set_option("gemm3", -100)       # back to "unset": FFVC_GEMM3 (default: off)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,Kd", [(1024, 512, 256), (768, 384, 96), (300, 200, 64), (2048, 1024, 1056), (256, 128, 64)])
def test_gemm3_every_epilogue_kind_matches_fp64(cuda, gemm3_forced, dt, M, N, Kd):
    """csrc/gemm3.hip forced on (ffvc_set_option gemm3 = 1): plain 16-bit store, fp32 + fp32 residual (with and without bias),
    GELU / QuickGELU forward (with pre-activation and with act' storage), aux-multiply dgrad with bias-gradient column sums —
    interior and ragged tiles, odd and even numbers of 32-deep stages — against fp64 math on the same 16-bit operands."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, Kd, generator=g).to(dt).cuda()
    w = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    tol = LOTOL[dt]
    p = x.double() @ w.double().t()
    sc = p.abs().max().item()
    y = torch.empty(M, N, dtype=dt, device=cuda)
    K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd)
    assert (y.double() - p).abs().max().item() < tol * sc
    y32 = torch.empty(M, N, dtype=torch.float32, device=cuda)
    K.gemm(x, w, y32, M, N, Kd, ldx=Kd, ldw=Kd, residual=res)
    assert (y32.double() - (p + res.double())).abs().max().item() < 3e-5 * (sc + 4)
    K.gemm(x, w, y32, M, N, Kd, ldx=Kd, ldw=Kd, residual=res, bias=b)
    assert (y32.double() - (p + res.double() + b.double())).abs().max().item() < 3e-5 * (sc + 8)
    pb = p + b.double()
    for code, fn, dfn in ((K.ACT_GELU, lambda t: F.gelu(t), lambda t: 0.5 * (1 + torch.erf(t / 2 ** 0.5)) + t * torch.exp(-0.5 * t * t) / (2 * torch.pi) ** 0.5),
                          (K.ACT_QUICKGELU, lambda t: t * torch.sigmoid(1.702 * t),
                           lambda t: torch.sigmoid(1.702 * t) * (1 + 1.702 * t * (1 - torch.sigmoid(1.702 * t))))):
        h, aux = torch.empty(M, N, dtype=dt, device=cuda), torch.empty(M, N, dtype=dt, device=cuda)
        K.gemm(x, w, h, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=code)
        assert (h.double() - fn(pb)).abs().max().item() < tol * (sc + 4)
        K.gemm(x, w, h, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=code, aux=aux, ldaux=N, flags=K.F_WRITE_PREACT)
        assert (h.double() - fn(pb)).abs().max().item() < tol * (sc + 4) and (aux.double() - pb).abs().max().item() < tol * (sc + 4)
        K.gemm(x, w, h, M, N, Kd, ldx=Kd, ldw=Kd, bias=b, act=code, aux=aux, ldaux=N, flags=K.F_WRITE_PREACT | K.F_AUX_ACTGRAD)
        assert (h.double() - fn(pb)).abs().max().item() < tol * (sc + 4)
        assert (aux.double() - dfn(pb)).abs().max().item() < (2e-3 if dt == torch.float16 else 1.6e-2) * 1.2
        d = torch.empty(M, N, dtype=dt, device=cuda)
        cs = torch.zeros(N, dtype=torch.float32, device=cuda)
        want = p * aux.double()
        if K.colsum_fusable(dt, N, Kd, N, N):
            K.gemm(x, w, d, M, N, Kd, ldx=Kd, ldw=Kd, act=code, aux=aux, ldaux=N, flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD, colsum=cs)
            assert (cs.double() - want.sum(0)).abs().max().item() < 2e-3 * want.abs().sum(0).max().item() + 1e-3
        else:
            K.gemm(x, w, d, M, N, Kd, ldx=Kd, ldw=Kd, act=code, aux=aux, ldaux=N, flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD)
        assert (d.double() - want).abs().max().item() < tol * want.abs().max().item() + 1e-6
        K.gemm(x, w, d, M, N, Kd, ldx=Kd, ldw=Kd, act=code, aux=aux, ldaux=N, flags=K.F_MUL_ACT_GRAD)      # aux read as a pre-activation
        assert (d.double() - p * dfn(aux.double())).abs().max().item() < 3 * tol * (p * dfn(aux.double())).abs().max().item() + 1e-6


def test_gemm3_is_bit_identical_to_the_256x256_kernel_on_plain_launches(cuda):
    """Same 16-bit operands, same fp32 accumulation order along K (32-deep MFMA steps in sequence): the two kernels must agree
    exactly on a plain launch, whatever the tile."""
    g = torch.Generator().manual_seed(6)
    M, N, Kd = 4096, 1024, 512
    x = torch.randn(M, Kd, generator=g).half().cuda()
    w = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).half().cuda()
    y0, y1 = torch.empty(M, N, dtype=torch.float16, device=cuda), torch.empty(M, N, dtype=torch.float16, device=cuda)
    K.set_option("gemm3", -1)
    K.gemm(x, w, y0, M, N, Kd, ldx=Kd, ldw=Kd)
    K.set_option("gemm3", 1)
    K.gemm(x, w, y1, M, N, Kd, ldx=Kd, ldw=Kd)
    K.set_option("gemm3", -100)       # back to "unset": FFVC_GEMM3 (default: off)
    assert torch.equal(y0, y1)


# ----------------------------------------------------------------------------- round 6: GroupNorm-backward statistics inside the dgrad convolution
@pytest.fixture
def row_tile_forced():
    K.set_option("conv_row", 2)        # the row-tile kernels whatever the grid size (the heuristic wants >= 256 tiles: batch 64 in the step)
    K._GNB_OK.clear()
    yield
    K.set_option("conv_row", 1)
    K._GNB_OK.clear()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,C,Cd,swish", [(2, 128, 128, 128, True), (1, 256, 128, 128, True), (2, 64, 256, 256, True), (4, 64, 128, 256, False)])
def test_dgrad_convolution_accumulates_the_groupnorm_backward_statistics(cuda, row_tile_forced, dt, B, H, C, Cd, swish):
    """FFVC_F_GNB_SUMS (csrc/conv3.hip + gemm_epilogue_perm16): the dgrad convolution that produces dy = d(loss)/d(act(GN(x))) also
    accumulates sum(ds) and sum(ds * xhat) per (image, group), ds = round_16(dy) * act'(.) * gamma — the statistics pass of
    ffvc_groupnorm_bwd.  Checked against fp64 math on the stored dy, and the apply-only backward against the two-pass one.
    (B, H, C, Cd): images, side, channels of the GroupNorm node (= dgrad output channels), channels of the convolution's output."""
    g = torch.Generator().manual_seed(17)
    G = 32
    x = torch.randn(B, H, H, C, generator=g).to(dt).cuda()                       # the GroupNorm node's input
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).cuda(), (0.2 * torch.randn(C, generator=g)).cuda()
    gout = (torch.randn(B, H, H, Cd, generator=g) * 0.1).to(dt).cuda()           # gradient of the convolution's output
    wd = (torch.randn(C, 9 * Cd, generator=g) * (9 * Cd) ** -0.5).to(dt).cuda()  # the dgrad filter [Cin, kh, kw, Cout]
    y, mean, rstd = K.groupnorm_fwd(x, gamma, beta, G, 1e-6, swish)
    dy0 = torch.empty(B, H, H, C, dtype=dt, device=cuda)
    K.gemm(gout, wd, dy0, B * H * H, C, 9 * Cd, ldw=9 * Cd, x_mode=K.OP_CONV3X3, conv=(H, H, Cd))
    assert K.conv_gnb_ok(gout, wd, dy0, x, mean, rstd, gamma, beta, B, H, H, Cd, C)
    sums = torch.zeros(B, G, 2, dtype=torch.float64, device=cuda)
    dy1 = torch.empty_like(dy0)
    K.gemm(gout, wd, dy1, B * H * H, C, 9 * Cd, ldw=9 * Cd, x_mode=K.OP_CONV3X3, conv=(H, H, Cd),
           gnb=(x, mean, rstd, gamma, beta, sums, swish, H * H, C // G))
    assert torch.equal(dy0, dy1)                                                  # the stored gradient itself is untouched
    xd, dyd = x.double(), dy1.double()
    xh = (xd.view(B, H * H, G, C // G) - mean.double().view(B, 1, G, 1)) * rstd.double().view(B, 1, G, 1)
    yv = xh * gamma.double().view(1, 1, G, C // G) + beta.double().view(1, 1, G, C // G)
    ds = dyd.view(B, H * H, G, C // G)
    if swish:
        sg = torch.sigmoid(yv)
        ds = ds * (sg * (1 + yv * (1 - sg)))
    ds = ds * gamma.double().view(1, 1, G, C // G)
    want = torch.stack([ds.sum((1, 3)), (ds * xh).sum((1, 3))], dim=-1)
    scale = ds.abs().sum((1, 3)).max().item()
    assert (sums - want).abs().max().item() < 2e-4 * scale, ((sums - want).abs().max().item(), scale)
    dres = torch.randn(B, H, H, C, generator=g).to(dt).cuda()
    dx2 = K.groupnorm_bwd(dy1, x, gamma, beta, mean, rstd, dres=dres, G=G, swish=swish)
    dx1 = K.groupnorm_bwd(dy1, x, gamma, beta, mean, rstd, dres=dres, G=G, swish=swish, sums=sums)
    assert (dx1.float() - dx2.float()).abs().max().item() <= 2 * LOTOL[dt] * dx2.float().abs().max().item()
