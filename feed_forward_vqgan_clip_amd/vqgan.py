"""VQGAN f16 decoder + codebook on ffvc HIP kernels (frozen, dgrad only).

Replaces what the reference obtains from taming-transformers (`load_vqgan_model`, main.py:84-103)
and calls in `synth` (main.py:140-143): `model.quantize.embedding.weight`, `model.decode(z_q)`.
Architecture/key names follow taming's `VQModel`/`Decoder`; a real
`vqgan_imagenet_f16_16384.ckpt` state_dict can be passed to `VQGAN(state_dict)`.

Layout: activations are NHWC in the compute dtype; every 3x3 conv is an implicit GEMM with bias /
residual / nearest-2x upsample fused; GroupNorm+swish is one stats + one apply kernel; the
single-head spatial attention is batched GEMMs + a row softmax.
"""
import math
import os
import types

import torch
from torch import nn

from . import kernels as K
from . import ops

F16_16384 = dict(ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=256,
                 z_channels=256, out_ch=3, embed_dim=256, n_embed=16384)


def random_state_dict(cfg=F16_16384, seed=1234, codebook_std=1.0):
    """Random-init weights with taming's key layout (synthetic benchmark weights):
    convs default kaiming-uniform, GroupNorm affine (1, 0), codebook N(0, 1) instead of taming's degenerate
    U(-1/n, 1/n)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cin, cout, k):
        bound = 1.0 / math.sqrt(cin * k * k)
        sd[name + ".weight"] = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def norm(name, c):
        sd[name + ".weight"] = torch.ones(c)
        sd[name + ".bias"] = torch.zeros(c)

    def res(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".nin_shortcut", cin, cout, 1)

    def attn(name, c):
        norm(name + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{name}.{n}", c, c, 1)

    ch, mult, nrb = cfg["ch"], cfg["ch_mult"], cfg["num_res_blocks"]
    sd["quantize.embedding.weight"] = torch.randn(cfg["n_embed"], cfg["embed_dim"], generator=g) * codebook_std
    conv("post_quant_conv", cfg["embed_dim"], cfg["z_channels"], 1)
    block_in = ch * mult[-1]
    conv("decoder.conv_in", cfg["z_channels"], block_in, 3)
    res("decoder.mid.block_1", block_in, block_in)
    attn("decoder.mid.attn_1", block_in)
    res("decoder.mid.block_2", block_in, block_in)
    curr = cfg["resolution"] // 2 ** (len(mult) - 1)
    for lvl in reversed(range(len(mult))):
        block_out = ch * mult[lvl]
        for i in range(nrb + 1):
            res(f"decoder.up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
            if curr in cfg["attn_resolutions"]:
                attn(f"decoder.up.{lvl}.attn.{i}", block_in)
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", block_in, block_in, 3)
            curr *= 2
    norm("decoder.norm_out", block_in)
    conv("decoder.conv_out", block_in, cfg["out_ch"], 3)
    return sd


_DEC_STREAMS = int(os.environ.get("FFVC_DEC_STREAMS", "1"))
_DEC_SIZES = os.environ.get("FFVC_DEC_SIZES", "")     # e.g. "24,40": unequal parts (the parts then drift out of phase)


class _Res:
    def __init__(self, sd, p, cdt, fp8=False):
        f = lambda k: sd[p + k].detach().float().cuda().contiguous()  # noqa: E731
        self.n1 = (f(".norm1.weight"), f(".norm1.bias"))
        self.n2 = (f(".norm2.weight"), f(".norm2.bias"))
        self.conv1 = ops.ConvWeights(sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], cdt, fp8)
        self.conv2 = ops.ConvWeights(sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], cdt, fp8)
        self.nin = None
        if (p + ".nin_shortcut.weight") in sd:
            self.nin = ops.Weights.frozen(sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"], cdt)

    def __call__(self, x):
        # fp8 decoder: each norm writes the e4m3 operand of the convolution behind it, and norm2's backward the e5m2 operand of conv1's dgrad
        xn, xid = ops.groupnorm_fork(x, *self.n1, True, f8_for=self.conv1)
        h = ops.conv3x3(xn, self.conv1, gn=True)                   # -> norm2: moments from the conv epilogue
        hn, _ = ops.groupnorm_fork(h, *self.n2, True, f8_for=self.conv2, grad_sole=True)
        sc = xid if self.nin is None else ops.linear(xid, self.nin)
        return ops.conv3x3(hn, self.conv2, residual=sc, gn=True)   # block output -> the next block's norm


class _Attn:
    def __init__(self, sd, p, cdt):
        self.norm = (sd[p + ".norm.weight"].detach().float().cuda().contiguous(),
                     sd[p + ".norm.bias"].detach().float().cuda().contiguous())
        w = torch.cat([sd[f"{p}.{n}.weight"].reshape(sd[f"{p}.{n}.weight"].shape[0], -1) for n in ("q", "k", "v")], 0)
        b = torch.cat([sd[f"{p}.{n}.bias"] for n in ("q", "k", "v")], 0)
        self.qkv = ops.Weights.frozen(w, b, cdt)           # fused q|k|v 1x1 convs: one GEMM, one consumer of GN(x)
        self.proj = ops.Weights.frozen(sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"], cdt)
        self.C = w.shape[1]

    def __call__(self, x):
        B, H, W, C = x.shape
        hn, xid = ops.groupnorm_fork(x, *self.norm, False)
        qkv = ops.linear(hn.view(B, H * W, C), self.qkv)
        o = ops.attention(qkv, 1, float(C) ** -0.5)
        out = ops.linear(o, self.proj, residual=xid.view(B, H * W, C), gn_hw=H * W)
        return ops.carry_gn(out, out.view(B, H, W, C))


class VQGAN:
    """Frozen VQGAN: `.quantize.embedding.weight` (fp32 codebook) and `.decode` like the object `synth` receives
    (main.py:140-143), plus the NHWC fast path used by the fused train step."""

    def __init__(self, state_dict, cfg=F16_16384, cdt=torch.float16, fp8=False):
        """fp8: the 3x3 convolutions whose geometry the fp8 row kernel covers (128-multiple channel counts, 64 / 128 / 256 k wide
        images, a full chip of tiles) run forward (e4m3 x e4m3) and dgrad (e5m2 x e4m3) on the fp8 MFMA path with per-tensor delayed
        scaling; everything else (GroupNorm, attention, 1x1 convs, the small levels, conv_out) stays in `cdt`."""
        if not torch.cuda.is_available():
            raise RuntimeError("VQGAN needs a HIP device; there is no CPU fallback")
        sd, self.cfg, self.cdt = state_dict, cfg, cdt
        self.fp8 = bool(fp8) and cdt != torch.float32
        cb = sd["quantize.embedding.weight"].detach().float().cuda().contiguous()
        self.quantize = types.SimpleNamespace(embedding=types.SimpleNamespace(weight=cb))
        self.codebook = cb
        self.cnorm = K.rownorm_sq(cb)
        # 16-bit modes: the distance GEMM runs split-precision on the f16 pipes (ops._VQFn); the parity mode keeps exact fp32 MFMA.
        # |c| must fit f16's range for the hi part (a real f16-16384 codebook is O(1))
        self.cb3 = K.split3(cb, torch.float16, weight_order=True) if (cdt != torch.float32 and cb.shape[1] % 8 == 0 and
                                                                     float(cb.abs().max()) < 6e4) else None
        self.post_quant = ops.Weights.frozen(sd["post_quant_conv.weight"], sd["post_quant_conv.bias"], cdt)
        d = "decoder"
        f8 = self.fp8
        self.conv_in = ops.ConvWeights(sd[d + ".conv_in.weight"], sd[d + ".conv_in.bias"], cdt, f8)
        self.mid = [_Res(sd, d + ".mid.block_1", cdt, f8), _Attn(sd, d + ".mid.attn_1", cdt), _Res(sd, d + ".mid.block_2", cdt, f8)]
        self.levels = []
        mult, nrb = cfg["ch_mult"], cfg["num_res_blocks"]
        curr = cfg["resolution"] // 2 ** (len(mult) - 1)
        for lvl in reversed(range(len(mult))):
            stages = []
            for i in range(nrb + 1):
                stages.append(_Res(sd, f"{d}.up.{lvl}.block.{i}", cdt, f8))
                if curr in cfg["attn_resolutions"]:          # decided from the CONFIG resolution (App. A.1)
                    stages.append(_Attn(sd, f"{d}.up.{lvl}.attn.{i}", cdt))
            up = None
            if lvl != 0:
                up = ops.ConvWeights(sd[f"{d}.up.{lvl}.upsample.conv.weight"], sd[f"{d}.up.{lvl}.upsample.conv.bias"], cdt, f8)
                curr *= 2
            self.levels.append((stages, up))
        self.norm_out = (sd[d + ".norm_out.weight"].detach().float().cuda().contiguous(),
                         sd[d + ".norm_out.bias"].detach().float().cuda().contiguous())
        self.conv_out = ops.ConvWeights(sd[d + ".conv_out.weight"], sd[d + ".conv_out.bias"], cdt)
        self.z_min, self.z_max = float(cb.min()), float(cb.max())      # main.py:645-646,763 use the scalar min/max
        self._dec_streams = []

    # -- NHWC fast path -------------------------------------------------------
    def decode_nhwc(self, z_q):
        """z_q: (B, S, S, C) compute dtype -> (B, 16S, 16S, 3) fp32 in [-1, 1]-ish (VQModel.decode).
        FFVC_DEC_STREAMS=n (experiment, default 1): the batch in n parts, each on its own HIP stream (forward AND backward: autograd
        replays a node on its forward stream), so that one part's HBM-bound GroupNorm passes can share the chip with another part's
        MFMA-bound convolutions."""
        n = _DEC_STREAMS
        if n > 1 and z_q.is_cuda and z_q.shape[0] >= 2 * n and z_q.shape[0] % n == 0:
            main = torch.cuda.current_stream()
            while len(self._dec_streams) < n - 1:
                self._dec_streams.append(torch.cuda.Stream())
            outs = []
            sizes = [int(v) for v in _DEC_SIZES.split(",")] if _DEC_SIZES else None
            parts = z_q.split(sizes, 0) if (sizes and sum(sizes) == z_q.shape[0] and len(sizes) == n) else z_q.chunk(n, 0)
            for i, part in enumerate(parts):
                if i == 0:
                    outs.append(self._decode_one(part))
                    continue
                st = self._dec_streams[i - 1]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs.append(self._decode_one(part))
            for st in self._dec_streams[:n - 1]:
                main.wait_stream(st)
            for o in outs[1:]:
                o.record_stream(main)
            return torch.cat(outs, 0)
        return self._decode_one(z_q)

    def _decode_one(self, z_q):
        h = ops.linear(z_q, self.post_quant)
        h = ops.conv3x3(h, self.conv_in, gn=True)
        for m in self.mid:
            h = m(h)
        for stages, up in self.levels:
            for s in stages:
                h = s(h)
            if up is not None:
                h = ops.conv3x3(h, up, upsample=True, gn=True)
        hn, _ = ops.groupnorm_fork(h, *self.norm_out, True)
        return ops.conv3x3(hn, self.conv_out, out_dtype=torch.float32)

    def quantize_nhwc(self, z_nhwc, force_idx=None):
        """(B,S,S,C) fp32 -> (z_q compute dtype with straight-through grad, indices)."""
        return ops.vector_quantize(z_nhwc, self.codebook, self.cnorm, self.cdt, force_idx, self.cb3)

    # -- reference-shaped API ---------------------------------------------------
    def decode(self, z_q):
        """(B, C, S, S) -> (B, 3, 16S, 16S), like taming's VQModel.decode."""
        x = ops.cast(z_q.permute(0, 2, 3, 1), self.cdt)
        return self.decode_nhwc(x).permute(0, 3, 1, 2)


def vector_quantize(x, codebook, vq=None):
    """main.py:134-138 signature. x: (..., C) fp32. `vq` supplies the cached ||c||^2 when given."""
    cb = codebook.detach().float().contiguous()
    cn = vq.cnorm if vq is not None else K.rownorm_sq(cb)
    return ops.vector_quantize(x.float(), cb, cn, torch.float32)[0]


def synth_nhwc(model, z_nhwc, force_idx=None):
    """NHWC core of synth(): (B,S,S,C) fp32 -> (xr NHWC fp32 in [0,1], indices)."""
    z_q, idx = model.quantize_nhwc(z_nhwc, force_idx)
    dec = model.decode_nhwc(z_q)
    return ops.clamp_with_grad(dec, 0.0, 1.0, mul=0.5, add=0.5), idx


def synth(model, z):
    """main.py:140-143: z (B,C,S,S) -> RGB (B,3,H,W) in [0,1] (a permuted view of the NHWC result)."""
    xr, _ = synth_nhwc(model, z.permute(0, 2, 3, 1))
    return xr.permute(0, 3, 1, 2)


def load_vqgan_model(config_path, checkpoint_path, cdt=torch.float16, fp8=False):
    """main.py:84-103: yaml read with PyYAML; the three targets the reference accepts —
      taming.models.vqgan.VQModel                        decoder / post_quant_conv / quantize.embedding
      taming.models.vqgan.GumbelVQ                       same decoder; the codebook is `quantize.embed` (main.py:95 aliases it)
      taming.models.cond_transformer.Net2NetTransformer  its `first_stage_model` (a VQModel) is the model (main.py:96-100)
    The checkpoint is a pytorch-lightning file: read through `checkpoint_io.tolerant_load` (lightning / omegaconf / taming are
    not installed), `state_dict` entry, strict=False semantics: only decoder / post_quant / codebook keys are used."""
    import yaml

    from . import checkpoint_io

    with open(config_path) as f:
        conf = yaml.safe_load(f)
    target = conf["model"]["target"]
    params = conf["model"]["params"]
    prefix = ""
    if target == "taming.models.cond_transformer.Net2NetTransformer":
        params = params["first_stage_config"]["params"]
        prefix = "first_stage_model."
    elif target not in ("taming.models.vqgan.VQModel", "taming.models.vqgan.GumbelVQ"):
        raise ValueError(f"unknown model type: {target}")
    dd = params["ddconfig"]
    cfg = dict(ch=dd["ch"], ch_mult=tuple(dd["ch_mult"]), num_res_blocks=dd["num_res_blocks"],
               attn_resolutions=tuple(dd["attn_resolutions"]), resolution=dd["resolution"], z_channels=dd["z_channels"],
               out_ch=dd["out_ch"], embed_dim=params["embed_dim"], n_embed=params["n_embed"])
    if str(checkpoint_path).startswith("random:"):
        sd = random_state_dict(cfg, seed=int(str(checkpoint_path).split(":", 1)[1]))
    else:
        ckpt = checkpoint_io.tolerant_load(checkpoint_path)
        sd = ckpt["state_dict"] if isinstance(ckpt, dict) and "state_dict" in ckpt else ckpt
        if prefix:
            sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        if "quantize.embedding.weight" not in sd and "quantize.embed.weight" in sd:      # GumbelVQ
            sd = dict(sd)
            sd["quantize.embedding.weight"] = sd["quantize.embed.weight"]
    return VQGAN(sd, cfg, cdt, fp8=fp8)
