"""CLIP ViT image tower + text tower on ffvc HIP kernels (frozen perceptor, dgrad only).

Stands in for the object `load_clip_model` returns (main.py:1308-1333): `.encode_text(tokens)`,
`.encode_image(x)`, `.logit_scale`.  Architecture and state_dict keys are those of
clip.model.CLIP, whose math is restated in the reference at cloob.py:170-255,412-553.

Precision plan: the image tower runs in the compute dtype (bf16 MFMA, fp32
accumulate, fp32 LayerNorm / softmax statistics, fp32 residual stream); the text tower (0.5 % of
the step's FLOPs, no backward) is evaluated at fp32 grade: on the exact fp32 MFMA in parity mode, as split-precision
f16 GEMMs (`_SplitLinear`) otherwise.
"""
import math
import os

import numpy as np
import torch

from . import kernels as K
from . import ops
from .kernels import ACT_GELU, ACT_QUICKGELU

VIT_B32 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
               context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)
VIT_B16 = dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
               context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12)
VIT_L14 = dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
               context_length=77, vocab_size=49408, transformer_width=768, transformer_heads=12, transformer_layers=12)


def random_state_dict(cfg=VIT_B32, seed=1234):
    """Random weights with CLIP's key layout and the reference-style init (cloob.py:481-508)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s, std=1.0: torch.randn(*s, generator=g) * std  # noqa: E731
    sd = {}
    W, L = cfg["vision_width"], cfg["vision_layers"]
    P, R = cfg["vision_patch_size"], cfg["image_resolution"]
    TW, TL = cfg["transformer_width"], cfg["transformer_layers"]

    def blocks(prefix, width, layers):
        proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
        attn_std, fc_std = width ** -0.5, (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}.resblocks.{i}"
            sd[p + ".attn.in_proj_weight"] = rn(3 * width, width, std=attn_std)
            sd[p + ".attn.in_proj_bias"] = torch.zeros(3 * width)
            sd[p + ".attn.out_proj.weight"] = rn(width, width, std=proj_std)
            sd[p + ".attn.out_proj.bias"] = torch.zeros(width)
            for n in ("ln_1", "ln_2"):
                sd[f"{p}.{n}.weight"] = torch.ones(width)
                sd[f"{p}.{n}.bias"] = torch.zeros(width)
            sd[p + ".mlp.c_fc.weight"] = rn(4 * width, width, std=fc_std)
            sd[p + ".mlp.c_fc.bias"] = torch.zeros(4 * width)
            sd[p + ".mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std)
            sd[p + ".mlp.c_proj.bias"] = torch.zeros(width)

    sd["visual.conv1.weight"] = rn(W, 3, P, P, std=1.0 / math.sqrt(3 * P * P))
    sd["visual.class_embedding"] = rn(W, std=W ** -0.5)
    sd["visual.positional_embedding"] = rn((R // P) ** 2 + 1, W, std=W ** -0.5)
    for n in ("ln_pre", "ln_post"):
        sd[f"visual.{n}.weight"] = torch.ones(W)
        sd[f"visual.{n}.bias"] = torch.zeros(W)
    blocks("visual.transformer", W, L)
    sd["visual.proj"] = rn(W, cfg["embed_dim"], std=W ** -0.5)
    sd["token_embedding.weight"] = rn(cfg["vocab_size"], TW, std=0.02)
    sd["positional_embedding"] = rn(cfg["context_length"], TW, std=0.01)
    blocks("transformer", TW, TL)
    sd["ln_final.weight"] = torch.ones(TW)
    sd["ln_final.bias"] = torch.zeros(TW)
    sd["text_projection"] = rn(TW, cfg["embed_dim"], std=TW ** -0.5)
    sd["logit_scale"] = torch.tensor(np.log(1 / 0.07), dtype=torch.float32)
    return sd


def _f(t):
    return t.detach().float().cuda().contiguous()


class _Block:
    def __init__(self, sd, p, cdt, need_dgrad, act=ACT_QUICKGELU, fp8=False):
        self.act = act
        fz = lambda w, b: ops.Weights.frozen(sd[w], sd[b], cdt, need_dgrad, fp8=fp8)  # noqa: E731
        self.ln1 = (_f(sd[p + ".ln_1.weight"]), _f(sd[p + ".ln_1.bias"]))
        self.ln2 = (_f(sd[p + ".ln_2.weight"]), _f(sd[p + ".ln_2.bias"]))
        self.in_proj = fz(p + ".attn.in_proj_weight", p + ".attn.in_proj_bias")
        self.out_proj = fz(p + ".attn.out_proj.weight", p + ".attn.out_proj.bias")
        self.c_fc = fz(p + ".mlp.c_fc.weight", p + ".mlp.c_fc.bias")
        self.c_proj = fz(p + ".mlp.c_proj.weight", p + ".mlp.c_proj.bias")

    def __call__(self, x, heads, cdt, causal):
        """cloob.py:202-205: x + attn(ln_1(x)); x + mlp(ln_2(x)). x: fp32 residual stream (N, L, D)."""
        f32 = torch.float32
        dh = x.shape[-1] // heads
        xn, xid = ops.layernorm_fork(x, *self.ln1, cdt, f8_for=self.in_proj)       # fp8 tower: LN writes the e4m3 operand of its linear
        o = ops.attention(ops.linear(xn, self.in_proj), heads, dh ** -0.5, causal, f8_for=self.out_proj)
        x = ops.linear(o, self.out_proj, residual=xid, out_dtype=f32)
        xn, xid = ops.layernorm_fork(x, *self.ln2, cdt, f8_for=self.c_fc)
        return ops.mlp(xn, self.c_fc, self.c_proj, self.act, residual=xid, out_dtype=f32)


class _SplitLinear:
    """Frozen fp32 Linear evaluated on the 16-bit matrix pipes at fp32 grade (ffvc_split3): the weight is stored once as
    f16 [N, 3K] = [hi | hi | lo], the activation is split per call into [hi | lo | hi]; one f16 GEMM of depth 3K with fp32
    accumulation then sums x_hi w_hi + x_lo w_hi + x_hi w_lo — every product but lo x lo (~2^-22 relative).  Bias, activation
    and the fp32 residual ride in the GEMM epilogue.  ~5x the speed of the exact fp32 MFMA (1/16 of the 16-bit rate)."""

    def __init__(self, weight, bias):
        w = weight.detach().reshape(weight.shape[0], -1).float().cuda().contiguous()
        self.N, self.K = w.shape
        # range handling: f16 segments overflow above 65504 and lose the lo term below 2^-14.  The weight gets a
        # per-tensor power-of-two scale that puts its largest entry at ~2^10 (exact: only exponents move) and leaves through the
        # GEMM's alpha; activations are checked once per checkpoint by CLIP._probe_text_tower (non-finite -> exact fp32 MFMA)
        amax = float(w.abs().max())
        self.scale = 2.0 ** (10 - int(np.ceil(np.log2(amax)))) if amax > 0 and np.isfinite(amax) else 1.0
        self.w3 = K.split3(w * self.scale, torch.float16, weight_order=True)
        self.bias = None if bias is None else bias.detach().float().cuda().contiguous()

    def __call__(self, x, act=K.ACT_NONE, residual=None):
        rows = x.numel() // self.K
        x3 = K.split3(x.reshape(rows, self.K))
        y = torch.empty(*x.shape[:-1], self.N, dtype=torch.float32, device=x.device)
        K.gemm(x3, self.w3, y, rows, self.N, 3 * self.K, ldx=3 * self.K, ldw=3 * self.K, bias=self.bias, residual=residual,
               act=act, alpha=1.0 / self.scale)
        return y


class _TextBlock:
    """ResidualAttentionBlock of the (frozen, forward-only) text tower, cloob.py:202-205, with the four Linear layers as
    split-precision GEMMs; LayerNorm, the causal attention (77 tokens) and the residual stream stay fp32."""

    def __init__(self, sd, p, act):
        self.act = act
        self.ln1 = (_f(sd[p + ".ln_1.weight"]), _f(sd[p + ".ln_1.bias"]))
        self.ln2 = (_f(sd[p + ".ln_2.weight"]), _f(sd[p + ".ln_2.bias"]))
        self.in_proj = _SplitLinear(sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"])
        self.out_proj = _SplitLinear(sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"])
        self.c_fc = _SplitLinear(sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
        self.c_proj = _SplitLinear(sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])

    def __call__(self, x, heads):
        f32 = torch.float32
        dh = x.shape[-1] // heads
        o = ops.attention(self.in_proj(ops.layernorm(x, *self.ln1, f32)), heads, dh ** -0.5, True)
        x = self.out_proj(o, residual=x)
        h = self.c_fc(ops.layernorm(x, *self.ln2, f32), act=self.act)
        return self.c_proj(h, residual=x)


class _TakeToken(torch.autograd.Function):
    """x[:, idx, :] of a contiguous fp32 (N, L, D) tensor (class token, cloob.py:251)."""

    @staticmethod
    def forward(ctx, x, idx):
        N, L, D = x.shape
        ctx.dims = (N, L, D, idx)
        out = torch.empty(N, D, dtype=torch.float32, device=x.device)
        K.copy_rows(x[:, idx], L * D, out, D, N, D)
        return out

    @staticmethod
    def backward(ctx, g):
        N, L, D, idx = ctx.dims
        dx = torch.zeros(N, L, D, dtype=torch.float32, device=g.device)
        K.copy_rows(g.contiguous(), D, dx[:, idx], L * D, N, D)
        return dx, None


class CLIP:
    def __init__(self, state_dict, cdt=torch.float16, vision_heads=None, text_heads=None, quick_gelu=True, fp8=False,
                 text_exact=None):
        """quick_gelu: OpenAI checkpoints and open_clip's `-quickgelu` architectures use x*sigmoid(1.702x) in the MLPs
        (cloob.py:179-181); the other open_clip architectures (ViT-B-32 / ViT-L-14 on LAION-2B, main.py:1323-1329) use
        the exact erf GELU.
        fp8: the four Linear layers of every image-tower block (in_proj, out_proj, c_fc, c_proj; forward and dgrad) run on
        the fp8 MFMA path (BASELINE.json configs[4]): e4m3 weights / activations, e5m2 gradients, per-tensor delayed
        scaling, fp32 accumulation; LayerNorm, softmax, residual stream, patch embedding and projection keep `cdt` / fp32.
        text_exact: True = the text tower's Linear layers on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the 16-bit
        rate; what round 2 shipped); False = split-precision f16 GEMMs (`_SplitLinear`, fp32-grade: features agree with the exact
        path to ~1e-6 relative, 7.4 -> ~2 ms per step at cfg2).  Default: exact in fp32 parity mode, split otherwise
        (FFVC_TEXT_EXACT=1 forces the exact path)."""
        if not torch.cuda.is_available():
            raise RuntimeError("CLIP needs a HIP device; there is no CPU fallback")
        sd, self.cdt = state_dict, cdt
        self.fp8 = bool(fp8)
        act = ACT_QUICKGELU if quick_gelu else ACT_GELU
        w = sd["visual.conv1.weight"]
        self.width, self.patch = w.shape[0], w.shape[-1]
        self.vision_heads = vision_heads or self.width // 64                       # cloob.py:446
        self.conv1 = ops.Weights.frozen(w.reshape(self.width, -1), None, cdt)      # K index = c*P*P + ky*P + kx
        pos = _f(sd["visual.positional_embedding"])
        self.vpos = pos
        self.cls_pos0 = (_f(sd["visual.class_embedding"]) + pos[0]).contiguous()
        self.grid = int(round(math.sqrt(pos.shape[0] - 1)))
        self.image_resolution = self.grid * self.patch
        self.ln_pre = (_f(sd["visual.ln_pre.weight"]), _f(sd["visual.ln_pre.bias"]))
        self.ln_post = (_f(sd["visual.ln_post.weight"]), _f(sd["visual.ln_post.bias"]))
        n = 0
        self.vblocks = []
        while f"visual.transformer.resblocks.{n}.ln_1.weight" in sd:
            self.vblocks.append(_Block(sd, f"visual.transformer.resblocks.{n}", cdt, True, act, fp8=fp8))
            n += 1
        self.vproj = ops.Weights.frozen(sd["visual.proj"].t().contiguous(), None, cdt)
        self.embed_dim = sd["visual.proj"].shape[1]
        # text tower: exact fp32
        f32 = torch.float32
        self.tok_emb = _f(sd["token_embedding.weight"])
        self.tpos = _f(sd["positional_embedding"])
        self.context_length = self.tpos.shape[0]
        self.twidth = self.tpos.shape[1]
        self.text_heads = text_heads or self.twidth // 64
        if text_exact is None:
            text_exact = cdt == f32 or os.environ.get("FFVC_TEXT_EXACT", "0") == "1"
        self.text_exact = bool(text_exact) or (self.twidth % 8 != 0)
        self.tblocks = []
        n = 0
        while f"transformer.resblocks.{n}.ln_1.weight" in sd:
            p = f"transformer.resblocks.{n}"
            self.tblocks.append(_Block(sd, p, f32, False, act) if self.text_exact else _TextBlock(sd, p, act))
            n += 1
        self.ln_final = (_f(sd["ln_final.weight"]), _f(sd["ln_final.bias"]))
        self.tproj = ops.Weights.frozen(sd["text_projection"].t().contiguous(), None, f32, False)
        self.logit_scale = _f(sd["logit_scale"]) if "logit_scale" in sd else torch.tensor(np.log(1 / 0.07)).cuda()
        if not self.text_exact:
            self._probe_text_tower(sd, act)

    def _probe_text_tower(self, sd, act):
        """One forward of the split-precision text tower on a probe batch (every position filled, SOT .. EOT), once per checkpoint:
        if an activation leaves f16's range somewhere (a checkpoint with large-magnitude channels), the features come out
        non-finite -> rebuild the tower on the exact fp32 MFMA instead of returning NaN features later."""
        V = self.tok_emb.shape[0]
        L = self.context_length
        g = torch.Generator().manual_seed(0)
        tok = torch.randint(1, max(2, V - 2), (4, L), generator=g)
        tok[:, -1] = V - 1
        feats = self.encode_text(tok)
        if not bool(torch.isfinite(feats).all()):
            self.text_exact = True
            f32 = torch.float32
            self.tblocks = [_Block(sd, f"transformer.resblocks.{n}", f32, False, act) for n in range(len(self.tblocks))]

    # -- image tower ------------------------------------------------------------
    def encode_patches(self, patches):
        """patches: (N, grid^2, 3*P*P) compute dtype, already mean/std normalised -> (N, embed_dim) fp32."""
        f32 = torch.float32
        x = ops.patch_embed(patches, self.conv1, self.cls_pos0, self.vpos)          # cloob.py:237-244
        x = ops.layernorm(x, *self.ln_pre, f32)                                     # :245
        for blk in self.vblocks:
            x = blk(x, self.vision_heads, self.cdt, False)                          # :247-249
        cls = _TakeToken.apply(x, 0)
        cn = ops.layernorm(cls, *self.ln_post, self.cdt)                            # :251
        return ops.linear(cn, self.vproj, out_dtype=f32)                            # :253-254

    def patchify(self, image):
        """(N,3,R,R) -> (N, grid^2, 3*P*P) in the compute dtype (API-compat path; the train step fuses this
        into the cutout kernel)."""
        N, P, g = image.shape[0], self.patch, self.grid
        p = image.reshape(N, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(N, g * g, 3 * P * P)
        return ops.cast(p.float(), self.cdt)

    def encode_image(self, image):
        return self.encode_patches(self.patchify(image))

    # -- text tower ---------------------------------------------------------------
    @torch.no_grad()
    def encode_text(self, text):
        """text: int64 (B, L) -> (B, embed_dim) fp32 (cloob.py:525-538)."""
        if text.dtype != torch.long:
            raise TypeError("encode_text expects int64 token ids (main.py:733 dispatches on torch.long)")
        f32 = torch.float32
        text = text.cuda().contiguous()
        B, L = text.shape
        x = K.gather_rows(self.tok_emb, text, f32, pos=self.tpos, period=L)         # :526-528
        for blk in self.tblocks:                                                    # causal mask :510-516
            x = blk(x, self.text_heads, f32, True) if self.text_exact else blk(x, self.text_heads)
        xn = ops.layernorm(x, *self.ln_final, f32)                                  # :532
        eot = K.eot_gather(xn, text)                                                # :536
        return ops.linear(eot, self.tproj, out_dtype=f32)

    def float(self):
        return self

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, _flag=False):
        return self
