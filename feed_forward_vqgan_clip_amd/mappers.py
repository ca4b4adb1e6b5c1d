"""Prompt -> VQGAN-latent mappers (the only trainable part of the step), MI355X-native.

Same constructor arguments, forward contract `(B, clip_dim+noise_dim) -> (B, C, S, S)` and
state_dict key layout as the reference classes, so released `.th`
checkpoints load unchanged; the arithmetic runs in ffvc HIP kernels (ops.py).  torch.nn
layers are used ONLY as parameter holders / default initialisers — their forward is never called.
"""
import os

import torch
from torch import nn

from . import kernels as K
from . import ops
from .arena import ParamArena
from .kernels import ACT_GELU


class _MapperBase(nn.Module):
    """Shared arena / shadow management."""

    cdt = torch.float16        # package-wide default (main.DEFAULT_COMPUTE_DTYPE)

    def prepare(self, cdt=None):
        """Move parameters into a flat arena and build the compute-dtype shadows. Idempotent per dtype."""
        if cdt is not None:
            self.cdt = cdt
        arena = getattr(self, "_ffvc_arena", None)
        if arena is not None and arena.cdt == self.cdt:
            return self
        self._ffvc_arena = None
        arena = ParamArena(self, self.cdt)
        self._build_packs(arena)
        arena.refresh()
        self.register_load_state_dict_post_hook(lambda m, _: m._ffvc_arena.refresh() if m._ffvc_arena else None)
        return self

    def _arena(self):
        if getattr(self, "_ffvc_arena", None) is None:
            self.prepare()
        return self._ffvc_arena

    def refresh_shadows(self):
        self._arena().refresh()

    def _build_packs(self, arena):
        raise NotImplementedError


class _PreNormResidual(nn.Module):
    """Parameter holder mirroring the reference's `mixer.{i}.{j}` sub-tree: `.norm` and `.fn`."""

    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = nn.LayerNorm(dim)


def _ff_holder(d_in, factor, conv):
    mk = (lambda a, b: nn.Conv1d(a, b, 1)) if conv else nn.Linear
    # indices 0 and 3 carry parameters; 1, 2, 4 are GELU / Dropout placeholders in the reference
    return nn.Sequential(mk(d_in, d_in * factor), nn.Identity(), nn.Identity(), mk(d_in * factor, d_in), nn.Identity())


class Mixer(_MapperBase):
    """MLP-Mixer mapper (reference: mlp_mixer_pytorch.py:70-91 `Mixer`, :25-38 `MLPMixer`)."""

    def __init__(self, input_dim, image_size, channels, patch_size, dim, depth, expansion_factor=4, dropout=0.0):
        super().__init__()
        assert (image_size % patch_size) == 0, "image must be divisible by patch size"
        if patch_size != 1:
            # the reference itself cannot run any other value: its forward ends in x.view(bs, S, S, C) on a [bs, (S/p)^2, C]
            # tensor (mlp_mixer_pytorch.py:88-89) and raises "shape ... is invalid for input of size ..." for p > 1
            # (tests/test_mappers_cpu.py pins that); main.py:479-488 only ever passes 1
            raise NotImplementedError("Mixer: patch_size must be 1 (the reference's own forward raises for any other value: "
                                      "mlp_mixer_pytorch.py:88-89 views [bs, (S/p)^2, C] as [bs, S, S, C])")
        self.dropout = float(dropout)       # nn.Dropout after GELU and after the second Linear of every FeedForward (:20-22)
        self.input_dim, self.channels, self.image_size, self.dim, self.depth = input_dim, channels, image_size, dim, depth
        P = image_size * image_size
        self.mixer = nn.Sequential(
            nn.Identity(),                                   # Rearrange placeholder (index 0)
            nn.Linear(channels, dim),
            *[nn.Sequential(_PreNormResidual(dim, _ff_holder(P, expansion_factor, True)),
                            _PreNormResidual(dim, _ff_holder(dim, expansion_factor, False))) for _ in range(depth)],
            nn.LayerNorm(dim),
        )
        self.proj = nn.Linear(input_dim, image_size * image_size * channels)
        self.final_proj = nn.Linear(dim, channels)

    def _build_packs(self, arena):
        mk = arena.make_weights
        self._w_proj = mk(self.proj.weight, self.proj.bias)
        self._w_embed = mk(self.mixer[1].weight, self.mixer[1].bias)
        self._w_final = mk(self.final_proj.weight, self.final_proj.bias)
        self._blocks = []
        for i in range(2, self.depth + 2):
            tok, ch = self.mixer[i][0], self.mixer[i][1]
            self._blocks.append((tok.norm, mk(tok.fn[0].weight, tok.fn[0].bias), mk(tok.fn[3].weight, tok.fn[3].bias),
                                 ch.norm, mk(ch.fn[0].weight, ch.fn[0].bias), mk(ch.fn[3].weight, ch.fn[3].bias)))
        # the channel MLP's two weight gradients of consecutive blocks go out as grouped launches (ops.WgradGroup).  The FIRST blocks
        # stay on per-layer launches: their gradients are the last ones backward produces, so under data parallelism every
        # millisecond they wait for a group to fill is exchange time nothing can hide (a group saves ~0.1 ms of kernel time)
        first = ops.wgrad_group_size() if self.depth > ops.wgrad_group_size() else 0
        ops.group_weights([b[4] for b in self._blocks[first:]])
        ops.group_weights([b[5] for b in self._blocks[first:]])

    def forward(self, x):
        self._arena()
        cdt, f32 = self.cdt, torch.float32
        B, S, C = x.shape[0], self.image_size, self.channels
        h = ops.linear(ops.cast(x.float(), cdt), self._w_proj)                  # mlp_mixer_pytorch.py:85
        h = ops.transpose_last2(h.view(B, C, S * S))                            # :86 + Rearrange (:31) -> (B, S*S, C)
        h = ops.linear(h, self._w_embed, out_dtype=f32)                         # :32  fp32 residual stream
        drop = self.dropout if self.training else 0.0                           # the reference never .eval()s the net in train
        for (n1, t1, t2, n2, c1, c2) in self._blocks:
            hn, hid = ops.layernorm_fork(h, n1.weight, n1.bias, cdt)
            h = ops.token_mlp(hn, t1, t2, residual=hid, out_dtype=f32, drop=drop)          # :34 (+ residual :14)
            hn, hid = ops.layernorm_fork(h, n2.weight, n2.bias, cdt)
            h = ops.mlp(hn, c1, c2, ACT_GELU, residual=hid, out_dtype=f32, drop=drop)      # :35
        fin = self.mixer[self.depth + 2]
        hn = ops.layernorm(h, fin.weight, fin.bias, cdt)                        # :37
        z = ops.linear(hn, self._w_final, out_dtype=f32)                        # :88
        return z.view(B, S, S, C).permute(0, 3, 1, 2)                           # :89-90 (non-contiguous, like the reference)


# ---------------------------------------------------------------------------
# VitGAN mappers (reference: vitgan.py:221-260 `Generator`, :262-305 `SimpleGenerator`)
# ---------------------------------------------------------------------------
_SLN_SHARE = os.environ.get("FFVC_SLN_SHARE", "0") != "0" and os.environ.get("FFVC_SLN_INPLACE", "0") != "0"      # A/B: one running gradient sum for the shared modulation input


class _SLN(nn.Module):
    """Holder for vitgan.py:8-21 SLN: `.ln` LayerNorm + scalar `.gamma`, `.beta` of shape (1,1,1)."""

    def __init__(self, dim):
        super().__init__()
        self.ln = nn.LayerNorm(dim)
        self.gamma = nn.Parameter(torch.randn(1, 1, 1))
        self.beta = nn.Parameter(torch.randn(1, 1, 1))


class _GAttention(nn.Module):
    def __init__(self, dim, num_heads, dim_head=None):
        super().__init__()
        self.num_heads = num_heads
        self.dim_head = int(dim / num_heads) if dim_head is None else dim_head       # vitgan.py:62
        self.weight_dim = self.num_heads * self.dim_head
        self.to_qkv = nn.Linear(dim, self.weight_dim * 3, bias=False)
        self.w_out = nn.Linear(self.weight_dim, dim, bias=True)


class _GMLP(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.linear1 = nn.Linear(dim, hidden)
        self.linear2 = nn.Linear(hidden, dim)


class _GBlock(nn.Module):
    def __init__(self, dim, num_heads, dim_head, mlp_ratio=4):
        super().__init__()
        self.attn = _GAttention(dim, num_heads, dim_head)
        self.norm1 = _SLN(dim)
        self.norm2 = _SLN(dim)
        self.mlp = _GMLP(dim, dim * mlp_ratio)


class _GEncoder(nn.Module):
    def __init__(self, dim, blocks, num_heads, dim_head):
        super().__init__()
        self.blocks = nn.Sequential(*[_GBlock(dim, num_heads, dim_head) for _ in range(blocks)])


class _VitGANBase(_MapperBase):
    def _build_vit_packs(self, arena):
        mk = arena.make_weights
        self._bp = []
        for blk in self.Transformer_Encoder.blocks:
            # 6 heads x (1024 // 6 = 170) channels = 1020: the 3060-wide qkv rows and the 1020-deep w_out reduction are padded to
            # multiples of 8 elements (zero weight rows / columns) so every GEMM of the block runs on the LDS-DMA kernels
            inner = blk.attn.num_heads * blk.attn.dim_head
            pq, po = (-3 * inner) % 8, (-inner) % 8
            self._bp.append((blk.norm1, mk(blk.attn.to_qkv.weight, None, pad_n=pq),
                             mk(blk.attn.w_out.weight, blk.attn.w_out.bias, pad_k=po),
                             blk.norm2, mk(blk.mlp.linear1.weight, blk.mlp.linear1.bias),
                             mk(blk.mlp.linear2.weight, blk.mlp.linear2.bias), blk.attn))
        self._w_mlp = mk(self.mlp.weight, self.mlp.bias)
        self._w_outp = mk(self.w_out[0].weight, self.w_out[0].bias)
        # 16 tokens x a per-GPU batch of 32 = 512 rows: a weight gradient is 64 tiles with an 8-step reduction, a launch that never
        # fills the chip.  The MLP's two gradients of 8 consecutive blocks go out as one grouped launch (ops.WgradGroup)
        g = int(os.environ.get("FFVC_VIT_WGRAD_GROUP", "0"))     # (0 = off, the default: see ops._SLN_INPLACE)
        ops.group_weights([b[4] for b in self._bp], g)
        ops.group_weights([b[5] for b in self._bp], g)

    def _encode(self, hl, x, share=None):
        """GTransformerEncoder (vitgan.py:120-164): hl, x fp32 (B, T, dim).  share: ops.SharedGrad collecting x's gradient."""
        cdt, f32 = self.cdt, torch.float32
        drop = self.dropout if self.training else 0.0      # after attention (:133) and inside MLP (:36-41)
        B, T, dim = x.shape
        align = 2 if cdt != torch.float32 else 1          # GEMM operands need 4-byte aligned head blocks
        for bi, (n1, Wqkv, Wout, n2, W1, W2, att) in enumerate(self._bp):
            H, dh = att.num_heads, att.dim_head
            dhp = (dh + align - 1) // align * align
            tiny = dh != 64 and K.attn_tiny_ok(T, dh)
            y, hid = ops.sln_fork(hl, x, n1.ln.weight, n1.ln.bias, n1.gamma, n1.beta, cdt, share, bi == 0)   # vitgan.py:132
            qkv = ops.linear(y, Wqkv)                                                            # (B,T,(d k h) + pad) :81
            if tiny:       # a handful of tokens: one launch per direction, read in the projection's own (d k h) order
                o = ops.attention_tiny(qkv, H, dh, float(dim) ** -0.5, "dkh", out_ld=Wout.K)     # :82-93, scale = dim^-0.5 :65
            else:
                if Wqkv.N != 3 * H * dh:
                    qkv = ops.copy2d(qkv, B * T, 3 * H * dh, Wqkv.N, 3 * H * dh)
                qkv = ops.transpose_pad(qkv.view(B * T, dh, 3 * H), dhp).view(B, T, 3 * H * dhp)  # -> (k h d) :82
                o = ops.attention(qkv, H, float(dim) ** -0.5)
                if dhp != dh or Wout.K != H * dh:                                                # per-head pad off, row pad on
                    o = ops.copy2d(o, B * T * H, dh, dhp, dh)
                    if Wout.K != H * dh:
                        o = ops.copy2d(o, B * T, H * dh, H * dh, Wout.K)
            hl = ops.linear(o.view(B, T, Wout.K), Wout, residual=hid, out_dtype=f32, drop=drop)  # w_out + hl :97,132
            y, hid = ops.sln_fork(hl, x, n2.ln.weight, n2.ln.bias, n2.gamma, n2.beta, cdt, share)
            hl = ops.mlp(y, W1, W2, ACT_GELU, residual=hid, out_dtype=f32, drop=drop)            # :133
        return hl


class Generator(_VitGANBase):
    def __init__(self, initialize_size=8, dim=384, blocks=6, num_heads=6, dim_head=None, dropout=0, out_channels=3,
                 input_dim=1024):
        super().__init__()
        self.dropout = float(dropout)
        self.initialize_size, self.dim, self.out_channels = initialize_size, dim, out_channels
        T = initialize_size * 8
        self.pos_emb1D = nn.Parameter(torch.randn(T, dim))
        self.mlp = nn.Linear(input_dim, T * dim)
        self.Transformer_Encoder = _GEncoder(dim, blocks, num_heads, dim_head)
        self.w_out = nn.Sequential(nn.Linear(dim, T * out_channels))
        self.sln_norm = _SLN(dim)

    def _build_packs(self, arena):
        self._build_vit_packs(arena)

    def forward(self, noise):
        self._arena()
        cdt, f32 = self.cdt, torch.float32
        T = self.initialize_size * 8
        B = noise.shape[0]
        x = ops.linear(ops.cast(noise.float(), cdt), self._w_mlp, out_dtype=f32).view(B, T, self.dim)   # vitgan.py:254
        share = ops.SharedGrad() if (self._bp and _SLN_SHARE) else None
        hl = self._encode(ops.broadcast_rows(self.pos_emb1D, B), x, share)                               # :255
        s = self.sln_norm
        y, _ = ops.sln_fork(hl, x, s.ln.weight, s.ln.bias, s.gamma, s.beta, cdt, share)                 # :256
        out = ops.linear(y, self._w_outp, out_dtype=f32)                                                # :257
        return out.view(B, self.out_channels, T, T)                                                     # raw view :258-259


class SimpleGenerator(_VitGANBase):
    def __init__(self, size=8, in_channels=256, dim=384, blocks=6, num_heads=6, dim_head=None, dropout=0,
                 out_channels=3, input_dim=1024):
        super().__init__()
        self.dropout = float(dropout)
        self.size, self.dim, self.out_channels = size, dim, out_channels
        N = size * size
        self.pos_emb1D = nn.Parameter(torch.randn(N, dim))
        self.mlp = nn.Linear(input_dim, N * dim)
        self.inp = nn.Linear(input_dim, N * dim)
        self.Transformer_Encoder = _GEncoder(dim, blocks, num_heads, dim_head)
        self.w_out = nn.Sequential(nn.Linear(dim, out_channels))
        self.sln_norm = _SLN(dim)

    def _build_packs(self, arena):
        self._build_vit_packs(arena)
        self._w_inp = arena.make_weights(self.inp.weight, self.inp.bias)

    def forward(self, noise):
        self._arena()
        cdt, f32 = self.cdt, torch.float32
        N, B = self.size * self.size, noise.shape[0]
        nz = ops.cast(noise.float(), cdt)
        inp = ops.linear(nz, self._w_inp, out_dtype=f32)                                                # vitgan.py:297
        x = ops.linear(nz, self._w_mlp, out_dtype=f32).view(B, N, self.dim)                             # :298
        inp_emb = ops.transpose_last2(inp.view(B, self.dim, N))                                         # :299
        share = ops.SharedGrad() if (self._bp and _SLN_SHARE) else None
        hl = self._encode(ops.broadcast_rows(self.pos_emb1D, B, inp_emb), x, share)                      # :300
        s = self.sln_norm
        y, _ = ops.sln_fork(hl, x, s.ln.weight, s.ln.bias, s.gamma, s.beta, cdt, share)
        out = ops.linear(y, self._w_outp, out_dtype=f32)
        return out.view(B, self.size, self.size, self.out_channels).permute(0, 3, 1, 2)                 # :303


# ---------------------------------------------------------------------------
# x-transformer mapper (reference: transformer.py:5-46 over x-transformers==0.19.1
# ContinuousTransformerWrapper(Decoder); PARITY UNPINNED upstream, see oracle/mappers.py)
# ---------------------------------------------------------------------------
class _XAttention(nn.Module):
    def __init__(self, dim, heads, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(dim, inner, bias=False)
        self.to_v = nn.Linear(dim, inner, bias=False)
        self.to_out = nn.Linear(inner, dim)


class _XFeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.Sequential(nn.Sequential(nn.Linear(dim, dim * mult), nn.Identity()), nn.Identity(),
                                 nn.Linear(dim * mult, dim))


class _XAttnLayers(nn.Module):
    def __init__(self, dim, depth, heads):
        super().__init__()
        layers = []
        for _ in range(depth):
            layers.append(nn.ModuleList([nn.LayerNorm(dim), _XAttention(dim, heads), nn.Identity()]))
            layers.append(nn.ModuleList([nn.LayerNorm(dim), _XFeedForward(dim), nn.Identity()]))
        self.layers = nn.ModuleList(layers)


class _XPosEmb(nn.Module):
    def __init__(self, dim, max_seq_len):
        super().__init__()
        self.emb = nn.Embedding(max_seq_len, dim)


class _XWrapper(nn.Module):
    def __init__(self, dim_in, dim_out, max_seq_len, dim, depth, heads):
        super().__init__()
        self.pos_emb = _XPosEmb(dim, max_seq_len)
        self.project_in = nn.Linear(dim_in, dim)
        self.attn_layers = _XAttnLayers(dim, depth, heads)
        self.norm = nn.LayerNorm(dim)
        self.project_out = nn.Linear(dim, dim_out)


class XTransformer(_MapperBase):
    def __init__(self, input_dim, image_size, channels, dim, depth, heads, initial_proj=True, add_input=True):
        super().__init__()
        self.input_dim, self.image_size, self.channels, self.dim, self.depth, self.heads = \
            input_dim, image_size, channels, dim, depth, heads
        self.add_input, self.initial_proj = add_input, initial_proj
        self.transformer = _XWrapper(dim if initial_proj else input_dim, channels,
                                     image_size * image_size + (0 if add_input else 1), dim, depth, heads)   # transformer.py:11-20
        if initial_proj:
            self.proj = nn.Linear(input_dim, image_size * image_size * dim)                                   # :22-23

    def _build_packs(self, arena):
        mk = arena.make_weights
        t = self.transformer
        self._w_proj = mk(self.proj.weight, self.proj.bias) if self.initial_proj else None
        self._w_in = mk(t.project_in.weight, t.project_in.bias)
        self._w_outp = mk(t.project_out.weight, t.project_out.bias)
        self._xl = []
        L = t.attn_layers.layers
        for j in range(self.depth):
            (n1, a, _), (n2, f, _) = L[2 * j], L[2 * j + 1]
            self._xl.append((n1, mk(a.to_q.weight, None), mk(a.to_k.weight, None), mk(a.to_v.weight, None),
                             mk(a.to_out.weight, a.to_out.bias), n2, mk(f.net[0][0].weight, f.net[0][0].bias),
                             mk(f.net[2].weight, f.net[2].bias)))

    def forward(self, x):
        self._arena()
        cdt, f32 = self.cdt, torch.float32
        B, S, dim = x.shape[0], self.image_size, self.dim
        n = S * S
        t = self.transformer
        xc = ops.cast(x.float(), cdt)
        if self.initial_proj:
            h = ops.linear(xc, self._w_proj).view(B, n, dim)                                # transformer.py:30-31
            h = ops.linear(h, self._w_in, out_dtype=f32)                                    # project_in
            L = n
        elif self.add_input:
            # every position carries the same input row (:34-36): project it once, broadcast over the n positions
            h = ops.linear(xc, self._w_in, out_dtype=f32).unsqueeze(1).expand(B, n, dim)
            L = n
        else:
            # token 0 = the input, n zero tokens behind it (:38-40); the zero rows go through project_in like any other
            # row (one use of the weight pack per step keeps the fused wgrad / all-reduce bookkeeping simple)
            xin = ops.copy2d(xc, B, xc.shape[1], xc.shape[1], (n + 1) * xc.shape[1]).view(B, n + 1, -1)   # [x | 0 ... 0] per sample
            h = ops.linear(xin, self._w_in, out_dtype=f32)
            L = n + 1
        if h.is_contiguous():                                                               # scaled abs. pos. emb.: own kernels
            h = ops.broadcast_rows(t.pos_emb.emb.weight[:L], B, h, dim ** -0.5)
        else:                                                                               # (the expanded add_input branch)
            h = (h + t.pos_emb.emb.weight[:L] * (dim ** -0.5)).contiguous()
        for (n1, Wq, Wk, Wv, Wo, n2, W1, W2) in self._xl:
            hn, hid = ops.layernorm_fork(h, n1.weight, n1.bias, cdt)                        # pre-norm
            o = ops.attention(ops.qkv3(hn, Wq, Wk, Wv), self.heads, 64 ** -0.5, True)       # causal, dim_head 64
            h = ops.linear(o, Wo, residual=hid, out_dtype=f32)
            hn, hid = ops.layernorm_fork(h, n2.weight, n2.bias, cdt)
            h = ops.mlp(hn, W1, W2, ACT_GELU, residual=hid, out_dtype=f32)
        hn = ops.layernorm(h, t.norm.weight, t.norm.bias, cdt)
        z = ops.linear(hn, self._w_outp, out_dtype=f32)
        if L != n:
            z = z[:, 1:].contiguous()                                                       # drop the input token (:42-43)
        return z.view(B, S, S, self.channels).permute(0, 3, 1, 2)                           # transformer.py:44-45


def build_other(config, input_dim, vq_image_size, vq_channels):
    """The non-Mixer branches of build_model (main.py:459-478,489-499)."""
    if config.model_type == "vitgan":
        return Generator(initialize_size=vq_image_size // 8, dropout=config.dropout, out_channels=vq_channels,
                         input_dim=input_dim, dim=config.dim, num_heads=config.get("num_heads", 6), blocks=config.depth)
    if config.model_type == "simple_vitgan":
        return SimpleGenerator(size=vq_image_size, dropout=config.dropout, out_channels=vq_channels, input_dim=input_dim,
                               dim=config.dim, num_heads=config.get("num_heads", 6), blocks=config.depth)
    if config.model_type == "xtransformer":
        return XTransformer(input_dim=input_dim, image_size=vq_image_size, channels=vq_channels, dim=config.dim,
                            depth=config.depth, heads=config.get("num_heads", 6),
                            initial_proj=config.get("initial_proj", True), add_input=config.get("add_input", False))
    raise ValueError("model_type should be 'vitgan' or  'mlp_mixer' or 'xtransformer'")
