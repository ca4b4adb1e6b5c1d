"""Host surface of the training hot path — same names / config keys / call order as the
reference's main.py (`train`, `build_model`, `load_model`, `load_dataset`, `load_clip_model`,
`load_vqgan_model`, `MakeCutouts`, `synth`, `vector_quantize`, `clamp_with_grad`,
`replace_grad`, `tv_loss`), with every tensor op of the step running in ffvc HIP kernels.

    python -m feed_forward_vqgan_clip_amd.main train configs/example.yaml

Built: `train`, `test` / `load_model` / the PIL image grids (inference and checkpoint compatibility, incl. legacy
pickled-module `model.th` files and pytorch-lightning VQGAN checkpoints through checkpoint_io), `tokenize`, `encode_text`,
`encode_text_and_images` (feature cache) and the Net2Net prior's `load_prior_model` / sampling.  Not built (SURVEY.md §2 "—"):
`evaluate`, `encode_text_and_images_webdataset`, `train_prior`.
No weights / BPE vocabulary ship with this repo: `vqgan_checkpoint: "random:<seed>"`,
`clip_model_path: "random:<seed>"` and `path: "synthetic:<n>"` select seeded synthetic
weights / token batches.
"""
import json
import math
import os
import sys
import time

import torch
from torch import nn

from . import augment as _augment
from . import clip as _clip
from . import distributed as hvd
from . import kernels as K
from . import ops
from . import vqgan as _vqgan
from .mappers import Mixer
from .optim import CosineAnnealingLR, FusedAdam
from .vqgan import load_vqgan_model, synth, synth_nhwc, vector_quantize  # noqa: F401

CLIP_SIZE = {"RN50": 224, "RN101": 224, "RN50x4": 288, "RN50x16": 384, "ViT-B/32": 224, "ViT-B/16": 224,
             "ViT-L/14": 224, "openclip/ViT-B-32-quickgelu/laion400m_e32": 224, "openclip/ViT-B-32/laion2b_e16": 224}
CLIP_DIM = {"RN50": 1024, "RN101": 512, "RN50x4": 640, "RN50x16": 768, "ViT-B/32": 512, "ViT-B/16": 512,
            "ViT-L/14": 768, "openclip/ViT-B-32-quickgelu/laion400m_e32": 512, "openclip/ViT-B-32/laion2b_e16": 512}
CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
CLIP_STD = [0.26862954, 0.26130258, 0.27577711]
SOT, EOT = 49406, 49407


class Config(dict):
    """OmegaConf-like: attribute access for required keys (KeyError -> AttributeError), `.get` for optional."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"missing config key '{k}'")

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def load(path):
        import yaml

        with open(path) as f:
            return Config(yaml.safe_load(f))


def clamp_with_grad(x, lo, hi):
    """ClampWithGrad.apply (main.py:118-132)."""
    return ops.clamp_with_grad(x, float(lo), float(hi))


class _ReplaceGrad(torch.autograd.Function):
    """ReplaceGrad (main.py:105-116): value of x_forward, gradient to x_backward (pure autograd routing)."""

    @staticmethod
    def forward(ctx, x_forward, x_backward):
        ctx.shape = x_backward.shape
        return x_forward.view_as(x_forward)

    @staticmethod
    def backward(ctx, g):
        return None, g.sum_to_size(ctx.shape)


replace_grad = _ReplaceGrad.apply


class MakeCutouts(nn.Module):
    """main.py:154-229.  Source image = (AdaptiveAvg+AdaptiveMax)/2 pooling to pool_size (pool=True, :213-215) or the
    decoded image itself (pool=False, :216-217), repeated cutn times; then the augmentation list driven by per-cutout parameters
    (augment.py): every name of main.py:166-198 — 'Af','Pe','Ji','Er' (the default set), 'Ro','Re','Re2','Cr','Cc','Ji2','Er2','Gn',
    'R' as fused resampling launches (ffvc_augment_fwd/bwd: one launch for the default set), 'Sh','Et','Ts' (sharpness / elastic /
    thin-plate spline) as their own image -> image kernels in between (augment.plan()); the additive noise
    `U(0,noise_fac)*N(0,1)` (:222-225); with interpolate=True adaptive average pooling to interp_size (:226-228).
    `sequential=True` (THE DEFAULT since round 5; config `augment_sequential`): every resampling operator gets its own pass, as
    kornia's nn.Sequential resamples (main.py:199: two bilinear interpolations for Af -> Pe) — what the reference computes.
    `sequential=False` composes consecutive warps into one interpolation (one launch for the default set): cheaper by one
    224x224 pass each way and measurably NOT what kornia computes (image 1.3e-2 rel-rms, loss up to 6.4e-5,
    profiles/r04_augment_deviation.txt), so it is opt-in."""

    def __init__(self, cut_size, cutn, cut_pow=1.0, pool_size=None, interp_size=None, augs=None, pool=True,
                 interpolate=False, sequential=True):
        super().__init__()
        augs = tuple(augs) if augs else ("Af", "Pe", "Ji", "Er")      # main.py:164-165 (empty list -> defaults)
        for a in augs:
            if a not in _augment.SUPPORTED:
                raise NotImplementedError(f"augs={list(augs)}: '{a}' is not one of main.py:166-198's names "
                                          f"({list(_augment.SUPPORTED)})")
        self.all_augs = augs
        self.augs = tuple(a for a in augs if a != "R")                # what is left when 'R' is the identity
        self.pool, self.interpolate = bool(pool), bool(interpolate)
        self.pool_size = pool_size or cut_size                         # main.py:200-201
        self.interp_size = interp_size or self.pool_size              # main.py:202-203
        self.cut_size, self.cutn, self.noise_fac = cut_size, cutn, 0.1
        self.sequential = bool(sequential)
        self.generator = None                                         # torch.Generator for reproducible parameter draws

    def _plain(self, H):
        """Only the pooling branch + noise: the single-kernel cutouts path."""
        return self.pool and self.pool_size == self.cut_size and not self.augs and \
            not (self.interpolate and self.interp_size != self.cut_size)

    def source_size(self, H):
        return self.pool_size if self.pool else H

    def batch_size_px(self, H):
        """Side of the augmented batch before the optional interpolate step (what the noise tensor must have)."""
        return _augment.out_size(self.cut_size, self.all_augs, self.source_size(H))

    def out_size_px(self, H):
        return self.interp_size if self.interpolate else self.batch_size_px(H)

    def draw_aug_params(self, n, device, H=None):
        """Per-cutout augmentation parameters (None when the configuration needs no resampling): the parameter dict of ONE fused
        launch, or the list of segments augment.plan() returns when the chain needs several."""
        src = self.source_size(H if H is not None else self.cut_size)
        if not self.augs and src == self.cut_size:
            return None
        chain = _augment.draw_chain(n, self.cut_size, self.all_augs if src != self.cut_size else self.augs,
                                    generator=self.generator, src_size=src, device=device)
        segs = _augment.plan(chain, n, self.cut_size, src, sequential=self.sequential)
        if torch.device(device).type == "cuda":
            # pinned staging (caching host allocator) keeps the upload asynchronous: a pageable copy would stall the host
            # until everything already queued on the stream (mapper + decoder) has run
            up = lambda v: v if (not torch.is_tensor(v) or v.is_cuda) else v.pin_memory().to(device, non_blocking=True)  # noqa: E731
            segs = [(kind, {k: up(v) for k, v in prm.items()}) for kind, prm in segs]
        return segs[0][1] if len(segs) == 1 else segs

    def patches(self, xr_nhwc, patch, mean, std, out_dtype, facs=None, noise=None, aug_params=None):
        """NHWC fp32 image batch -> normalised ViT patch rows with the configured augmentations."""
        B, H = xr_nhwc.shape[0], xr_nhwc.shape[1]
        n = self.cutn * B
        size = self.batch_size_px(H)
        if facs is None and self.noise_fac:
            facs, noise = self.draw_noise(n, xr_nhwc.device, size)
        if self._plain(H):
            return ops.cutouts(xr_nhwc, self.cut_size, self.cutn, patch, mean, std, out_dtype, noise=noise, facs=facs)
        src = self.source_size(H)
        if aug_params is None:
            aug_params = self.draw_aug_params(n, xr_nhwc.device, H)
        if aug_params is None:                                        # identity chain: still goes through the resampler
            aug_params = _augment.draw_params(n, self.cut_size, (), src_size=src)
            aug_params = {k: v.to(xr_nhwc.device) for k, v in aug_params.items()}
        segs = [("fused", aug_params)] if isinstance(aug_params, dict) else list(aug_params)
        gn = None
        for kind, prm in segs:
            if kind == "fused" and prm.get("gn") is not None and "Gn" in self.augs:
                gn = prm["gn"] if gn is None else torch.maximum(gn, prm["gn"])
        if gn is not None:                              # 'Gn' N(0,1) noise + the U(0,noise_fac)*N(0,1) term = one Gaussian
            if noise is None:
                noise = torch.randn(n, 3, size, size, device=xr_nhwc.device)
            facs = gn.to(xr_nhwc.device) if facs is None else torch.sqrt(facs * facs + gn.to(facs.device) ** 2)
        # pool=False: the same kernel at cut == H is the identity pooling, i.e. a NHWC -> NCHW copy of the decoded image
        x = ops.cutouts(xr_nhwc, src, 1, src, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), torch.float32).view(B, 3, src, src)
        zero3, one3 = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
        cutn = self.cutn                                  # the first fused launch repeats the pooled image cutn times
        for i, (kind, prm) in enumerate(segs):
            last = i == len(segs) - 1
            if kind == "Sh":
                x = ops.sharpness(x, prm["factor"], prm["on"])
            elif kind == "Et":
                x = ops.warp_grid(x, K.elastic_grid(prm["noise"]), prm["on"])
            elif kind == "Ts":
                x = ops.warp_grid(x, K.tps_grid(prm["tps"], x.shape[-1]), prm["on"])
            elif not last:                                # intermediate fused launch: a plain fp32 image batch, no noise
                so = prm.get("out", size)
                x = ops.augment(x, prm, cutn, so, zero3, one3, torch.float32, out_size=so).view(n, 3, so, so)
                cutn = 1
            elif not self.interpolate or self.interp_size == size:
                return ops.augment(x, prm, cutn, patch, mean, std, out_dtype, noise=noise, facs=facs, out_size=size)
            else:
                batch = ops.augment(x, prm, cutn, size, zero3, one3, torch.float32, noise=noise, facs=facs,
                                    out_size=size).view(n, 3, size, size)
                return ops.avgpool_patches(batch, self.interp_size, patch, mean, std, out_dtype)          # main.py:226-228
        raise RuntimeError("MakeCutouts: the augmentation plan did not end in a fused launch")

    def draw_noise(self, n, device, size=None):
        if not self.noise_fac:
            return None, None
        size = size or self.cut_size
        facs = torch.empty(n, device=device).uniform_(0, self.noise_fac)             # main.py:224
        noise = torch.randn(n, 3, size, size, device=device)                         # main.py:225
        return facs, noise

    def forward(self, input, facs=None, noise=None, aug_params=None):
        """(B,3,H,W) in [0,1] -> (cutn*B, 3, S, S) fp32, cut-major like `repeat(cutn,1,1,1)` (main.py:218); S = cut_size
        for every configuration whose chain resizes / crops (or interp_size with interpolate=True)."""
        xr = input.permute(0, 2, 3, 1)
        S = self.out_size_px(input.shape[2])
        out = self.patches(xr.float(), S, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), torch.float32, facs, noise, aug_params)
        return out.view(self.cutn * input.shape[0], 3, S, S)


def tv_loss(Y_hat):
    """main.py:423-428 on an NCHW batch, computed by ffvc_tv_loss_fwd/bwd on the NHWC view."""
    return ops.tv_loss_nhwc(Y_hat.permute(0, 2, 3, 1).float())


DEFAULT_COMPUTE_DTYPE = "f16"     # the ONE default of train / test / bench.py / INTEGRATION.md (see _cdt)


def _cdt(config):
    """Compute dtype of the step (config key `compute_dtype`: f16 | bf16 | fp32).  Default f16: IEEE half storage / MFMA inputs
    with fp32 accumulation, fp32 residual streams, statistics, VQ distances and loss, and a loss-scaled backward guarded against
    overflow (optim.FusedAdam.check_overflow).  BASELINE.json names bf16 for cfg2; f16 has the same width and MFMA rate and an
    8x finer mantissa, which is what keeps the loss inside north_star's 1e-4 of the CPU reference (bf16: 6e-5..2e-3, measured);
    `compute_dtype: bf16` selects BASELINE's own format, bench.py times both."""
    name = str(config.get("compute_dtype", DEFAULT_COMPUTE_DTYPE))
    return {"bf16": torch.bfloat16, "f16": torch.float16, "fp16": torch.float16, "fp32": torch.float32, "f32": torch.float32}[name]


def clip_arch(model_type):
    """(architecture dict, quick_gelu) of a perceptor name: OpenAI names (main.py:1330-1332) or
    `openclip/<arch>/<pretrained>` (main.py:1323-1329; open_clip architectures use the erf GELU unless the name says
    `-quickgelu`)."""
    table = {"ViT-B/32": _clip.VIT_B32, "ViT-B/16": _clip.VIT_B16, "ViT-L/14": _clip.VIT_L14,
             "ViT-B-32": _clip.VIT_B32, "ViT-B-16": _clip.VIT_B16, "ViT-L-14": _clip.VIT_L14}
    if model_type.startswith("openclip/"):
        parts = model_type.split("/")
        if len(parts) != 3:
            raise ValueError(f"clip_model '{model_type}': expected openclip/<arch>/<pretrained>")
        name = parts[1]
        quick = name.endswith("-quickgelu")
        arch = table.get(name[:-len("-quickgelu")] if quick else name)
    else:
        arch, quick = table.get(model_type), True
    if arch is None:
        raise ValueError(f"clip_model '{model_type}' is not built on the HIP path (ViT-B/32, ViT-B/16, ViT-L/14 and their "
                         f"open_clip spellings; the ResNet / CLOOB perceptors are out of scope)")
    return arch, quick


def load_clip_model(model_type, path=None, cdt=torch.float16, fp8=False):
    """main.py:1308-1333 for the OpenAI / OpenCLIP ViT families (state_dict in clip.model.CLIP layout).
    fp8: image-tower linears on the fp8 MFMA path (config key `clip_fp8`, BASELINE.json configs[4])."""
    arch, quick = clip_arch(model_type)
    if path is None or str(path).startswith("random:"):
        seed = int(str(path).split(":", 1)[1]) if path else 1234
        sd = _clip.random_state_dict(arch, seed)
    else:
        obj = torch.load(path, map_location="cpu", weights_only=False)
        sd = obj.state_dict() if hasattr(obj, "state_dict") else obj.get("state_dict", obj)
    return _clip.CLIP(sd, cdt, quick_gelu=quick, fp8=fp8)


def synthetic_tokens(n, seed=0, context_length=77):
    """Seeded int64 token rows: SOT, L~U{4..30} ids in [1, 49405], EOT (= max id so argmax finds it), zeros."""
    g = torch.Generator().manual_seed(seed)
    toks = torch.zeros(n, context_length, dtype=torch.long)
    L = torch.randint(4, 31, (n,), generator=g)
    for i in range(n):
        toks[i, 0] = SOT
        toks[i, 1:1 + L[i]] = torch.randint(1, SOT, (int(L[i]),), generator=g)
        toks[i, 1 + L[i]] = EOT
    return toks


def load_dataset(path, bpe_path=None):
    """main.py:1293-1306.  `.pkl` files (token tensor or (inp, out) feature tuple) load as in the reference; a text file
    (one prompt per line) or a glob of text files is tokenised with the CLIP BPE tokenizer (tokenizer.py; the vocabulary
    file is not shipped: FFVC_BPE_VOCAB / bpe_path); `synthetic:<n>[:seed]` yields seeded token rows."""
    if path.startswith("synthetic:"):
        parts = path.split(":")
        return synthetic_tokens(int(parts[1]), int(parts[2]) if len(parts) > 2 else 0)
    if path.endswith("pkl"):
        return torch.load(path, weights_only=False)
    from . import tokenizer
    if "*" in path:
        from glob import glob
        texts = [open(f).read().strip() for f in sorted(glob(path))]
    else:
        texts = [t.strip() for t in open(path).readlines()]
    return tokenizer.tokenize(texts, truncate=True, bpe_path=bpe_path)


def tokenize(paths, out="tokenized.pkl", max_length=None, batch_size=None, bpe_path=None):
    """main.py:395-421: tokenise a prompt file / glob once and save the int64 rows to a `.pkl` dataset."""
    from . import tokenizer
    if "*" in paths:
        from glob import glob
        texts = [open(f).read().strip() for f in sorted(glob(paths))]
    else:
        texts = [ln.strip() for ln in open(paths).readlines()]
        if max_length:
            texts = [t for t in texts if len(t) <= max_length]
    toks = tokenizer.tokenize(texts, truncate=True, bpe_path=bpe_path)
    torch.save(toks, out)
    return toks


def clip_dim_size(config):
    """(clip_dim, clip_size) of a config: explicit keys, then the reference's tables (main.py:53-80), then the architecture
    of an `openclip/<arch>/<pretrained>` name the tables do not list."""
    dim = config.get("clip_dim", CLIP_DIM.get(config.clip_model))
    size = config.get("clip_size", CLIP_SIZE.get(config.clip_model))
    if dim is None or size is None:
        arch, _ = clip_arch(config.clip_model)
        dim = dim if dim is not None else arch["embed_dim"]
        size = size if size is not None else arch["image_resolution"]
    return dim, size


def encode_text(data_path, out="text_features.pkl", clip_model="ViT-B/32", clip_path=None, batch_size=256, bpe_path=None):
    """Text-feature cache: run the frozen text tower ONCE over a prompt dataset and save the fp32
    features; `train` on the resulting `.pkl` takes the pre-computed-feature branch of main.py:733,737 (anything that is
    not torch.long is used as features), so the text tower leaves the training step altogether."""
    toks = load_dataset(data_path, bpe_path=bpe_path)
    toks = toks[0] if isinstance(toks, tuple) else toks
    if toks.dtype != torch.long:
        raise TypeError("encode_text: the dataset already holds features")
    perceptor = load_clip_model(clip_model, path=clip_path)
    feats = torch.cat([perceptor.encode_text(toks[i:i + batch_size].cuda()).float().cpu()
                       for i in range(0, len(toks), batch_size)])
    torch.save(feats, out)
    return feats


def clip_preprocess(img, size):
    """clip.load's eval transform [upstream clip-anytorch]: bicubic resize of the short side to `size`, centre crop,
    RGB, [0,1], CLIP mean/std.  img: PIL image -> (3, size, size) fp32."""
    import numpy as np
    from PIL import Image
    w, h = img.size
    s = size / min(w, h)
    img = img.convert("RGB").resize((max(size, round(w * s)), max(size, round(h * s))), Image.BICUBIC)
    w, h = img.size
    l, t_ = (w - size) // 2, (h - size) // 2
    a = torch.from_numpy(np.asarray(img.crop((l, t_, l + size, t_ + size)), dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (a - torch.tensor(CLIP_MEAN).view(3, 1, 1)) / torch.tensor(CLIP_STD).view(3, 1, 1)


def encode_text_and_images(folder, *, img_ext="jpg", text_ext="txt", out="features.pkl", clip_model="ViT-B/32",
                           clip_path=None, bpe_path=None, batch_size=64):
    """main.py:231-279: (caption, image) file pairs of a folder -> (text_features, image_features) `.pkl`, the pair
    dataset `train` consumes with `input_loss` (main.py:812-824): inputs = text features, targets = image features."""
    from glob import glob

    from PIL import Image

    from . import tokenizer
    text_paths = sorted(glob(os.path.join(folder, "*." + text_ext)))
    if not text_paths:
        raise FileNotFoundError(f"encode_text_and_images: no *.{text_ext} files in {folder}")
    img_paths = [t[:-len(text_ext)] + img_ext for t in text_paths]
    perceptor = load_clip_model(clip_model, path=clip_path)
    size = clip_dim_size(Config(clip_model=clip_model))[1]
    tf, imf = [], []
    for i in range(0, len(text_paths), batch_size):
        toks = tokenizer.tokenize([open(p).read() for p in text_paths[i:i + batch_size]], truncate=True, bpe_path=bpe_path)
        tf.append(perceptor.encode_text(toks.cuda()).float().cpu())
        imgs = torch.stack([clip_preprocess(Image.open(p), size) for p in img_paths[i:i + batch_size]])
        with torch.no_grad():
            imf.append(perceptor.encode_image(imgs.cuda()).float().cpu())
    feats = (torch.cat(tf), torch.cat(imf))
    torch.save(feats, out)
    return feats


def build_model(config, vq_channels=None):
    """main.py:448-502 (the VQGAN is NOT re-loaded here just to read z_channels; pass it in)."""
    clip_dim = clip_dim_size(config)[0]
    if vq_channels is None:
        vq_channels = config.get("vq_channels", 256)
    vq_image_size = config.get("vq_image_size", 16)
    noise_dim = config.noise_dim
    if config.model_type == "mlp_mixer":
        net = Mixer(input_dim=clip_dim + noise_dim, image_size=vq_image_size, channels=vq_channels, patch_size=1,
                    dim=config.dim, depth=config.depth, dropout=config.dropout)
    elif config.model_type in ("vitgan", "simple_vitgan", "xtransformer"):
        from . import mappers
        net = mappers.build_other(config, clip_dim + noise_dim, vq_image_size, vq_channels)
    else:
        raise ValueError("model_type should be 'vitgan' or  'mlp_mixer' or 'xtransformer'")
    return net


def load_model(path, cdt=torch.float16, vq_channels=256):
    """main.py:1273-1290: dict checkpoints {"state_dict","config","step","epoch"} and the legacy form, a pickled module
    instance (`model.th`).  The legacy object graph is read with stand-in classes (checkpoint_io), its parameters are
    flattened into a state_dict and loaded into a freshly built mapper, so neither the reference's classes nor its
    `_fix_*_gelu_issue` patches are needed."""
    from . import checkpoint_io

    ckpt = checkpoint_io.tolerant_load(path)
    if isinstance(ckpt, dict):
        config, sd = Config(checkpoint_io.plain_config(ckpt["config"])), ckpt["state_dict"]
    elif checkpoint_io.is_module_like(ckpt):
        config = Config(checkpoint_io.plain_config(getattr(ckpt, "config", None) or {}))
        if "model_type" not in config:
            raise ValueError(f"{path}: pickled module without a usable `config` attribute")
        sd = checkpoint_io.module_state_dict(ckpt)
    else:
        raise ValueError(f"{path}: neither a checkpoint dict nor a pickled module")
    net = build_model(config, vq_channels)
    net.load_state_dict(sd)
    net.config = config
    return net.cuda().prepare(cdt)


def reserve_device_memory(gib=None, device=None):
    """Hand torch's caching allocator ONE large free block before the first step (sized for 288 GB of HBM3E: 45 % of the device by
    default, `FFVC_RESERVE_GIB` / config `reserve_gib` override, 0 = off).  The step's working set (~45 GB of saved activations at cfg2,
    ~20 GB of it crossing to the weight-gradient stream every step) otherwise makes the allocator return to hipMalloc for hundreds of MB at a
    time — 14 GB of new segments over 20 steps after the warm-up, measured — and a hipMalloc is an occasional 30-60 ms stall in the middle
    of a step.  A block the allocator already owns is split and re-joined without the driver."""
    if not torch.cuda.is_available():
        return 0.0
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    if gib is None:
        env = os.environ.get("FFVC_RESERVE_GIB")
        gib = float(env) if env is not None else 0.45 * torch.cuda.get_device_properties(device).total_memory / 2 ** 30
    free, _ = torch.cuda.mem_get_info(device)
    gib = min(float(gib), 0.8 * free / 2 ** 30)
    if gib < 1.0:
        return 0.0
    blk = torch.empty(int(gib * 2 ** 30), dtype=torch.uint8, device=device)
    del blk                      # back to the allocator's free list (NOT to the driver): one segment of `gib` GiB
    # Requests of up to 1 MiB live in a pool of their own, grown 2 MiB (one hipMalloc) at a time.  The 512-row mappers (VitGAN /
    # x-transformer: cfg3 / cfg4) allocate thousands of such tensors per step, some of them released across streams, so that pool kept
    # growing long after the warm-up: 2-83 new segments inside ten timed steps, and a step anywhere between 59 and 118 ms
    # (profiles/r06_cfg3_launch_diet.txt).  Pre-grow it on the stream of the step and on the weight-gradient stream.
    small = os.environ.get("FFVC_RESERVE_SMALL_MIB")
    small = 1024 if small is None else int(small)
    if small > 0:
        streams = [(torch.cuda.current_stream(device), small)]
        if ops._SIDE["enabled"]:
            with torch.cuda.device(device):
                streams.append((ops._side_stream(), small // 4))
        for st, mib in streams:
            with torch.cuda.stream(st):
                hold = [torch.empty(1 << 20, dtype=torch.uint8, device=device) for _ in range(mib)]
            del hold
    return gib


class TrainStep:
    """One rank's state for the loop body of `train` (main.py:715-837): mapper, frozen VQGAN + CLIP, cutout
    parameters, optimizer.  `__call__(inp, out)` performs forward, loss, backward and the optimizer step."""

    def __init__(self, config, net, vq, perceptor, opt, scheduler=None):
        self.config, self.net, self.vq, self.perceptor, self.opt, self.scheduler = config, net, vq, perceptor, opt, scheduler
        clip_model = config.clip_model
        self.clip_dim, self.clip_size = clip_dim_size(config)
        self.cutn, self.repeat = config.cutn, config.repeat
        self.make_cutouts = MakeCutouts(cut_size=config.get("cut_size", self.clip_size), cutn=self.cutn,
                                        augs=config.get("augs"), pool=config.get("pool", True),
                                        pool_size=config.get("pool_size", self.clip_size),
                                        interpolate=config.get("interpolate", False),
                                        interp_size=config.get("interp_size", self.clip_size),
                                        sequential=config.get("augment_sequential", True))
        if config.get("noise_fac") is not None:
            self.make_cutouts.noise_fac = config.get("noise_fac")
        if config.diversity_coef:
            raise NotImplementedError("diversity_coef > 0 needs the LPIPS VGG16 features (out of scope, SURVEY.md §2)")
        self.noise_dim, self.nb_noise = config.noise_dim, config.nb_noise
        self.NOISE = None
        if self.noise_dim and self.nb_noise:                                     # main.py:680-687
            self.NOISE = getattr(net, "NOISE", None)
            if self.NOISE is None:
                self.NOISE = torch.randn(self.nb_noise, self.noise_dim)
            self.NOISE = hvd.broadcast(self.NOISE.cuda(), root_rank=0)
            net.NOISE = self.NOISE
        self.input_loss = config.get("input_loss", False)
        self.input_loss_coef = config.get("input_loss_coef", 1)
        self.target_loss_coef = config.get("target_loss_coef", 1)
        self.clip_grad_norm = config.get("clip_grad_norm")
        self.normalize_input = config.get("normalize_input", False)
        self.l2_coef, self.tv_coef = config.get("l2_coef", 0.0), config.get("tv_coef", 0.0)
        self._prefetched, self._text_stream = None, None
        self._graph = None                                                      # captured step (enable_graph)
        # Host run-ahead bound (round 6): the host enqueues a step in ~25 ms, the GPU runs it in ~128, so without a synchronisation point
        # (bench.py's timed loop has none) the host is a dozen steps ahead within seconds.  At most `steps_in_flight` steps are queued;
        # the GPU still never idles.  Measured neutral on the step time and on the allocator (the growth of torch's caching allocator
        # inside the timed region — 14 GB of new segments per 20 steps — is fragmentation, fixed by reserve_device_memory above).
        self.steps_in_flight = int(config.get("steps_in_flight", os.environ.get("FFVC_STEPS_IN_FLIGHT", "2")))
        self._inflight = []

    def features(self, t):
        if t.dtype != torch.long:
            return t.float().cuda()
        pf = self._prefetched
        if pf is not None and pf[0] == (t.data_ptr(), tuple(t.shape)):
            self._prefetched = None
            torch.cuda.current_stream().wait_event(pf[2])
            pf[1].record_stream(torch.cuda.current_stream())
            return pf[1]
        return self.perceptor.encode_text(t).float()

    def prefetch(self, tokens):
        """Encode the NEXT batch's prompts (main.py:733, frozen text tower, no dependence on the weights being
        trained) on its own HIP stream, so the small fp32 GEMMs fill idle CUs under the current step's backward pass
        instead of sitting on the critical path.  The following `__call__` / `forward_loss` with the same token tensor
        picks the result up."""
        if tokens is None or tokens.dtype != torch.long or not tokens.is_cuda:
            return
        if self._text_stream is None:
            self._text_stream = torch.cuda.Stream()
        side = self._text_stream
        side.wait_stream(torch.cuda.current_stream())
        tokens.record_stream(side)
        with torch.cuda.stream(side), torch.no_grad():
            feats = self.perceptor.encode_text(tokens).float()
        ev = torch.cuda.Event()
        ev.record(side)
        self._prefetched = ((tokens.data_ptr(), tuple(tokens.shape)), feats, ev, tokens)

    def forward_loss(self, inp, out=None, facs=None, noise=None, aug_params=None, noise_vec_in=None, force_idx=None):
        """main.py:729-811 -> (loss, intermediates).  facs / noise / aug_params / noise_vec_in pin the step's random
        draws (cutout noise, augmentation parameters, the mapper's conditioning noise) for parity tests."""
        inp_feats = self.features(inp)                                          # :733
        return self._loss_from_feats(inp_feats, None if (out is None or out is inp) else self.features(out), facs, noise, aug_params,
                                     noise_vec_in, force_idx)

    def _loss_from_feats(self, inp_feats, out_feats_in=None, facs=None, noise=None, aug_params=None, noise_vec_in=None, force_idx=None):
        """forward_loss behind the text tower: everything from the prompt features on (the part a captured step replays)."""
        K.fp8_flush_updates()           # fp8 tower / decoder: last step's amax -> this step's scales, one launch for all tensor streams
        raw = inp_feats
        if self.normalize_input:
            inp_feats = torch.nn.functional.normalize(inp_feats, dim=1)         # :734-735
        out_feats = raw if out_feats_in is None else out_feats_in               # :737 (out is inp: the same encoding, not repeated)
        bs = len(inp_feats)
        if self.repeat != 1:
            inp_feats = inp_feats.repeat(self.repeat, 1)                        # :739-740
            out_feats = out_feats.repeat(self.repeat, 1)
        inp_feats_net, noise_vec = inp_feats, None
        if self.noise_dim and noise_vec_in is not None:
            noise_vec = noise_vec_in
            inp_feats_net = torch.cat((inp_feats, noise_vec), dim=1)
        elif self.noise_dim:                                                    # :741-751
            if self.nb_noise:
                inds = torch.randperm(len(self.NOISE))[:self.repeat]
                noise_vec = self.NOISE[inds.to(self.NOISE.device)].repeat(bs, 1).view(bs, self.repeat, -1) \
                    .permute(1, 0, 2).contiguous().view(bs * self.repeat, -1)
            else:
                noise_vec = torch.randn(len(inp_feats), self.noise_dim, device=inp_feats.device)
            inp_feats_net = torch.cat((inp_feats, noise_vec), dim=1)
        z = self.net(inp_feats_net)                                             # :754
        z_nhwc = z.permute(0, 2, 3, 1)                                          # contiguous for NHWC-native mappers
        l2 = ops.mean_sq(z_nhwc) if self.l2_coef > 0 else None                  # :758-762 (mean is layout-independent)
        z_nhwc = ops.clamp_with_grad(z_nhwc, self.vq.z_min, self.vq.z_max)      # :763
        xr, idx = synth_nhwc(self.vq, z_nhwc, force_idx)                        # :767 (force_idx: parity instrumentation)
        patches = self.make_cutouts.patches(xr, self.perceptor.patch, tuple(CLIP_MEAN), tuple(CLIP_STD),
                                            self.perceptor.cdt, facs, noise, aug_params)   # :796-797 fused
        embed = self.perceptor.encode_patches(patches)                          # :799
        dists = ops.spherical_loss(embed, out_feats, self.target_loss_coef)     # :801-811
        if self.input_loss:
            dists = dists + ops.spherical_loss(embed, inp_feats, self.input_loss_coef)   # :812-824
        loss = dists
        tv = None
        if l2 is not None:
            loss = loss + self.l2_coef * l2                                     # :831
        if self.tv_coef > 0:
            tv = ops.tv_loss_nhwc(xr)                                           # :769-773
            loss = loss + self.tv_coef * tv                                     # :831
        return loss, {"z": z, "xr": xr, "embed": embed, "indices": idx, "text_feats": inp_feats, "dists": dists,
                      "l2": l2, "tv": tv, "noise_vec": noise_vec}

    # ---- captured step (hipGraph) --------------------------------------------------------------------------------------------
    # The mappers with thousands of small launches per step (VitGAN, x-transformer: ~2400) are bound by the host's enqueue rate
    # (35 ms of Python / ctypes per step against ~45 ms of kernels).  enable_graph() records everything behind the text tower —
    # mapper, VQ, decoder, cutouts, image tower, loss, backward, weight-gradient side stream, Adam — ONCE into a hipGraph
    # (torch.cuda.CUDAGraph: capture on a fresh stream, allocations from the graph's private pool, no synchronisation or
    # hipMalloc inside: the library's own scratch buffers exist after the eager warm-up steps) and replays it per step.  What
    # changes from step to step enters through device memory: the prompt features (static buffer, filled by the eager text
    # tower / prefetch), the augmentation parameters (drawn on the host as before, copied into static device tensors), the
    # cutout noise (torch's graph-safe Philox state), the optimizer's scalars (FusedAdam.graph_pre_step).
    def enable_graph(self, batch_size, warmup_tokens=None):
        """Capture the step for `batch_size` prompts per call.  Needs dropout 0, no noise bank draw on the host (nb_noise), a
        single process (the gradient exchange stays eager) and at least one eager step before (scratch allocations)."""
        if hvd.is_distributed():
            raise RuntimeError("enable_graph: the captured step is single-GPU (the RCCL exchange is not captured)")
        if float(self.config.get("dropout", 0) or 0) > 0:
            raise RuntimeError("enable_graph: dropout seeds are host-side kernel arguments; dropout must be 0")
        if self.noise_dim and self.nb_noise:
            raise RuntimeError("enable_graph: the noise-bank draw (torch.randperm on the host) is not capturable")
        dev = next(self.net.parameters()).device
        B = int(batch_size)
        n = self.cutn * B * self.repeat
        self._g_B = B
        self._g_feats = torch.zeros(B, self.clip_dim, dtype=torch.float32, device=dev)
        H = 16 * int(self.config.vq_image_size)
        prm = self.make_cutouts.draw_aug_params(n, dev, H)
        self._g_aug = prm                                            # static device tensors (dict or list of segments)
        self.opt.enable_graph_hyper()
        # One EAGER pass on the stream the capture will use: the library keeps per-stream scratch (in-kernel split-K partial tiles)
        # that is hipMalloc'ed on a stream's first use — inside a capture that would invalidate it — and torch's allocator
        # warms the pool.  It is a genuine training step on `warmup_tokens`.
        if warmup_tokens is None:
            raise RuntimeError("enable_graph: pass one token batch for the eager warm-up step on the capture stream")
        self._g_stream = torch.cuda.Stream()
        self._g_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._g_stream):
            self._g_feats.copy_(self.features(warmup_tokens))
            self._g_warm_out = self._graph_body()
        torch.cuda.current_stream().wait_stream(self._g_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self.opt._capturing = True
        try:
            with torch.cuda.graph(g, stream=self._g_stream):
                self._g_out = self._graph_body()
        finally:
            self.opt._capturing = False
        self._graph = g

    def _graph_body(self):
        loss, mid = self._loss_from_feats(self._g_feats, aug_params=self._g_aug)
        self.opt.zero_grad()
        ls = getattr(self.opt, "loss_scale", 1.0)
        self._g_ls = ls
        (loss if ls == 1.0 else loss * ls).backward()
        if self.clip_grad_norm:
            self.opt.clip_grad_norm_(self.clip_grad_norm)
        self.opt.step()
        return loss.detach(), mid

    def _refresh_static_aug(self, n, H):
        """Fresh augmentation draws into the static device tensors the captured kernels read."""
        new = self.make_cutouts.draw_aug_params(n, "cpu", H)
        if new is None:                                              # configurations without a resampling chain
            return
        pairs = [(self._g_aug, new)] if isinstance(new, dict) else [(a[1], b[1]) for a, b in zip(self._g_aug, new)]
        for dst, src in pairs:
            for k, v in src.items():
                if torch.is_tensor(v):
                    dst[k].copy_(v.pin_memory(), non_blocking=True)

    def _graph_step(self, inp, next_inp):
        feats = self.features(inp)
        self._g_feats.copy_(feats, non_blocking=True)
        self.prefetch(next_inp)
        self._refresh_static_aug(self.cutn * self._g_B * self.repeat, 16 * int(self.config.vq_image_size))
        if getattr(self.opt, "loss_scale", 1.0) != self._g_ls:       # the loss scale is baked into the recorded backward
            # re-capture.  enable_graph's eager warm-up IS a training step on `inp` (optimizer step counted once): it stands in
            # for this call's replay, so the batch is applied once and the scheduler ticks once
            self._graph = None
            self.enable_graph(self._g_B, inp)
            out = self._g_warm_out
        else:
            self.opt.graph_pre_step()
            self._graph.replay()
            out = self._g_out
        if self.scheduler is not None:
            self.scheduler.step()
        return out[0].clone(), out[1]          # the captured loss tensor is overwritten by the next replay: hand out a copy

    def __call__(self, inp, out=None, facs=None, noise=None, aug_params=None, next_inp=None, noise_vec_in=None):
        if (self._graph is not None and out is None and facs is None and noise is None and aug_params is None and
                noise_vec_in is None and len(inp) == self._g_B):
            return self._graph_step(inp, next_inp)
        loss, mid = self.forward_loss(inp, out, facs, noise, aug_params, noise_vec_in)
        self.prefetch(next_inp)                                                 # next batch's text tower under this backward
        self.opt.zero_grad()                                                    # :825
        ls = getattr(self.opt, "loss_scale", 1.0)
        (loss if ls == 1.0 else loss * ls).backward()                           # :832 (f16 mode: loss-scaled gradients)
        if self.clip_grad_norm:
            self.opt.clip_grad_norm_(self.clip_grad_norm)                       # :833-834 (after the exchange in DP)
        self.opt.step()                                                         # :835
        if self.scheduler is not None:
            self.scheduler.step()                                               # :836-837
        self._throttle()
        return loss.detach(), mid

    def _throttle(self):
        """Wait (host side only) until at most `steps_in_flight` steps are queued on the GPU; see __init__."""
        if self.steps_in_flight <= 0:
            return
        ev = torch.cuda.Event()
        ev.record()
        self._inflight.append(ev)
        while len(self._inflight) > self.steps_in_flight:
            self._inflight.pop(0).synchronize()


def make_grid(images, nrow=8, padding=2):
    """torchvision.utils.make_grid for a (N,3,H,W) float batch in [0,1] (main.py:899-901) -> uint8 HxWx3 numpy array."""
    import numpy as np
    x = images.detach().float().cpu().clamp(0, 1)
    n, c, h, w = x.shape
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(n / xmaps))
    H, W = h + padding, w + padding
    grid = torch.zeros(c, H * ymaps + padding, W * xmaps + padding)
    for k in range(n):
        yy, xx = divmod(k, xmaps)
        grid[:, yy * H + padding:yy * H + padding + h, xx * W + padding:xx * W + padding + w] = x[k]
    return (grid.permute(1, 2, 0) * 255).add_(0.5).clamp_(0, 255).to(torch.uint8).numpy() if c == 3 else \
        np.repeat((grid[0] * 255).add_(0.5).clamp_(0, 255).to(torch.uint8).numpy()[:, :, None], 3, axis=2)


def save_grid(images, path, nrow=8):
    from PIL import Image
    Image.fromarray(make_grid(images, nrow=nrow)).save(path)


@torch.no_grad()
def generate(net, vq, feats):
    """Forward-only mapper -> clamp -> VQ -> decoder (main.py:934-942,1057-1059): (n, clip_dim[+noise]) -> (n,3,H,W)."""
    z = net(feats)
    z_nhwc = ops.clamp_with_grad(z.permute(0, 2, 3, 1), vq.z_min, vq.z_max)
    xr, _ = synth_nhwc(vq, z_nhwc)
    return xr.permute(0, 3, 1, 2)


def train(config_file):
    """main.py:504-974: the train loop with the reference's artefacts — `checkpoint.th` / `checkpoint_ema.th` / `opt.th`
    (same dict layouts), `progress*.png` / `fixed_batch_progress*.png` grids, fast CLIP-score evaluation (`eval_path`),
    scalars `loss, dists, diversity, l2, tv` (JSONL instead of tensorboard, which is not installed offline)."""
    config = Config.load(config_file)
    if "folder" not in config:
        config.folder = os.path.dirname(config_file)
    os.makedirs(config.folder, exist_ok=True)
    use_ema = config.get("use_ema", False)
    hvd.init()
    if torch.cuda.is_available():
        torch.cuda.set_device(hvd.local_rank())
    cdt = _cdt(config)
    if config.get("batch_invariant", False):
        # the library reads the switch at its first GEMM launch: with it, a prompt's latent and codes do not depend on the batch around
        # it (tests/test_fullsize_gpu.py::test_batch_rows_bit_identical_without_inkernel_splitk); costs the small-batch ViT / VitGAN shapes
        os.environ["FFVC_SK_FIXUP"] = "0"
    toks = load_dataset(config.path)
    vq = load_vqgan_model(config.vqgan_config, config.vqgan_checkpoint, cdt, fp8=bool(config.get("decoder_fp8", False)))
    perceptor = load_clip_model(config.clip_model, path=config.get("clip_model_path"), cdt=cdt, fp8=bool(config.get("clip_fp8", False)))
    vq_channels = vq.codebook.shape[1]
    checkpoint_path = os.path.join(config.folder, "checkpoint.th")
    checkpoint_ema_path = os.path.join(config.folder, "checkpoint_ema.th")
    net = build_model(config, vq_channels)
    net.step, net.epoch = 0, 0
    if os.path.exists(checkpoint_path):
        print(f"Resuming model from checkpoint {checkpoint_path}...")
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        net.load_state_dict(ckpt["state_dict"])
        net.epoch, net.step = ckpt["epoch"], ckpt["step"]
    net = net.cuda().prepare(cdt)
    net.config = config
    base_lr = config.lr
    opt = FusedAdam(net.parameters(), lr=base_lr)
    opt.loss_scale = float(config.get("loss_scale", 4096.0 if cdt == torch.float16 else 1.0))
    opt.skip_step_on_overflow = bool(config.get("skip_step_on_overflow", True))   # an overflowed f16 backward skips the WHOLE update
    opt_path = os.path.join(config.folder, "opt.th")
    if os.path.exists(opt_path):
        print(f"Resuming optimizer state from {opt_path}")
        opt.load_state_dict(torch.load(opt_path, map_location="cpu", weights_only=False))
    if use_ema:                                                                   # main.py:598-616
        ema_state = None
        if os.path.exists(checkpoint_ema_path):
            ema_state = torch.load(checkpoint_ema_path, map_location="cpu", weights_only=False)["state_dict"]
        opt.enable_ema(config.get("ema_decay", 0.995), ema_state)
    log_interval = config.get("log_interval", 100)
    rank_zero = hvd.rank() == 0
    if hvd.size() > 1:
        # fp32 on the wire like hvd.DistributedOptimizer(opt) (main.py:627) unless the configuration asks: `grad_wire: bf16` for
        # every slice, `grad_wire_tail: bf16` for the slices no backward work is left to hide
        opt = hvd.DistributedOptimizer(opt, wire_dtype=torch.bfloat16 if config.get("grad_wire") == "bf16" else None,
                                       tail_wire_dtype=torch.bfloat16 if config.get("grad_wire_tail") == "bf16" else None)
        hvd.broadcast_parameters(net, root_rank=0)
        hvd.broadcast_optimizer_state(opt, root_rank=0)
    scheduler = None
    if config.get("scheduler") is not None:
        if config.scheduler == "cosine":                                          # main.py:702-709; resumes mid-schedule
            scheduler = CosineAnnealingLR(opt, T_max=config.max_steps, eta_min=0, base_lrs=[base_lr], last_epoch=net.step)
        else:
            raise ValueError(config.scheduler)
    data = tuple(toks) if isinstance(toks, tuple) else (toks, toks)
    same = data[0] is data[1]
    print(f"Number of examples:{len(data[0])}")
    eval_data = load_dataset(config.eval_path) if config.get("eval_path") else None      # main.py:661-666
    eval_perceptor = perceptor
    if eval_data is not None and config.get("eval_clip_model"):
        eval_perceptor = load_clip_model(config.eval_clip_model, path=config.get("eval_clip_model_path"), cdt=cdt)
    bs = config.batch_size
    sampler = hvd.DistributedSampler(len(data[0]), shuffle=True)
    stepper = TrainStep(config, net, vq, perceptor, opt, scheduler)
    if config.get("reserve_gib") is not None:                                  # opt-in for training runs (bench.py reserves by default)
        reserve_device_memory(float(config.get("reserve_gib")))
    first_sel = torch.tensor(list(iter(sampler))[:bs])
    first_batch = (data[0][first_sel], data[1][first_sel])                       # main.py:679
    log_f = open(os.path.join(config.folder, "scalars.jsonl"), "a") if rank_zero else None
    avg_dev = torch.ones((), dtype=torch.float32, device="cuda")                  # avg_loss = 1. (main.py:694)
    step = net.step
    t_last = time.time()
    zero = torch.zeros((), dtype=torch.float32, device="cuda")
    try:
        for epoch in range(net.epoch, config.epochs):
            sampler.set_epoch(epoch)
            order = list(iter(sampler))
            nxt = None
            for i in range(0, len(order), bs):
                if nxt is None:
                    sel = torch.tensor(order[i:i + bs])
                    nxt = (data[0][sel].cuda(), data[1][sel].cuda())
                inp, out = nxt
                nxt = None
                if i + bs < len(order):                  # one batch of look-ahead for the text-tower prefetch
                    sel = torch.tensor(order[i + bs:i + 2 * bs])
                    nxt = (data[0][sel].cuda(), data[1][sel].cuda())
                loss, mid = stepper(inp, None if same else out, next_inp=nxt[0] if (nxt is not None and same) else None)
                avg_dev.mul_(0.99).add_(loss, alpha=0.01)        # main.py:861, every step, no host sync
                if step % log_interval == 0:                     # a collective: EVERY rank takes it at the same steps
                    if opt.loss_scale != 1.0:                    # f16: overflow events since the last log -> back the scale off
                        n_bad = opt.check_overflow()
                        if n_bad and rank_zero:
                            print(f"step:{step:05d} non-finite gradients in {n_bad} wavefront(s) since the last log: those "
                                  f"elements were skipped, loss_scale -> {opt.loss_scale:g}")
                    sc = hvd.allreduce_scalars(loss, mid["dists"].detach(), zero if mid["l2"] is None else mid["l2"].detach(),
                                               zero if mid["tv"] is None else mid["tv"].detach(), avg_dev)
                    if rank_zero:
                        lv, dv, l2v, tvv, avg = [float(t.item()) for t in sc]
                        dt, t_last = time.time() - t_last, time.time()
                        print(f"epoch:{epoch:03d}, step:{step:05d}, avg_loss:{avg:.3f}, loss:{lv:.3f}, dists:{dv:.3f}, "
                              f"div:0.000, l2:{l2v:.3f} tv:{tvv} sec/interval:{dt:.2f}")
                        rec = {"step": step, "loss": lv, "dists": dv, "diversity": 0.0, "l2": l2v, "tv": tvv, "avg_loss": avg,
                               "lr": opt.param_groups[0]["lr"]}
                        if eval_data is not None:
                            rec.update(_fast_eval(net, vq, eval_perceptor, eval_data, bs, stepper.clip_size, stepper.noise_dim))
                            print(f"Eval dists: {rec['eval_dists']:.3f}\nEval clip score: {rec['eval_clip_score']:.3f}")
                        log_f.write(json.dumps(rec) + "\n")
                        log_f.flush()
                        xr = mid["xr"].detach().permute(0, 3, 1, 2)
                        save_grid(xr, os.path.join(config.folder, "progress.png"), nrow=bs)           # main.py:899-901
                        save_grid(xr, os.path.join(config.folder, f"progress_{step:010d}.png"), nrow=bs)
                        net.step = step
                        torch.save({"state_dict": net.state_dict(), "config": dict(config), "step": step, "epoch": epoch},
                                   checkpoint_path)
                        if use_ema:
                            torch.save({"state_dict": opt.ema_state_dict(), "config": dict(config), "step": step,
                                        "epoch": epoch}, checkpoint_ema_path)
                        torch.save(opt.state_dict(), opt_path)
                        _save_fixed_batch(stepper, first_batch, use_ema, os.path.join(config.folder, "fixed_batch_progress"),
                                          step, bs)
                step += 1
                if config.get("max_steps") is not None and step >= config.max_steps:
                    return
    finally:
        if log_f is not None:
            log_f.close()


@torch.no_grad()
def _fast_eval(net, vq, perceptor, eval_data, bs, clip_size, noise_dim=0):
    """main.py:868-897: CLIP distance / score of images generated for held-out prompts (bilinear resize to clip_size).
    (With noise_dim > 0 the reference feeds the bare text features to the mapper and fails on the input width; here a
    fresh noise vector is appended, as `test` does, main.py:1041-1053.)"""
    ds, cs = [], []
    scale = perceptor.logit_scale.exp().float().cpu()
    mean = torch.tensor(CLIP_MEAN, device="cuda").view(1, -1, 1, 1)
    std = torch.tensor(CLIP_STD, device="cuda").view(1, -1, 1, 1)
    for i in range(0, len(eval_data), bs):
        chunk = eval_data[i:i + bs].cuda()
        emb = perceptor.encode_text(chunk).float() if chunk.dtype == torch.long else chunk.float()
        xr = generate(net, vq, emb if not noise_dim else torch.cat((emb, torch.randn(len(emb), noise_dim, device=emb.device)), 1))
        xr = torch.nn.functional.interpolate(xr, size=(clip_size, clip_size), mode="bilinear")
        embed = torch.nn.functional.normalize(perceptor.encode_image((xr - mean) / std).float(), dim=1)
        H = torch.nn.functional.normalize(emb, dim=-1)
        ds.append(H.sub(embed).norm(dim=-1).div(2).arcsin().pow(2).mul(2).cpu())
        cs.append((scale * (H * embed).sum(dim=1).cpu()))
    return {"eval_dists": float(torch.cat(ds).mean()), "eval_clip_score": float(torch.cat(cs).mean())}


@torch.no_grad()
def _save_fixed_batch(stepper, first_batch, use_ema, prefix, step, bs):
    """main.py:920-948: images of a fixed batch, with the EMA weights when enabled."""
    inp = first_batch[0].cuda()
    feats = stepper.features(inp)
    if stepper.normalize_input:
        feats = torch.nn.functional.normalize(feats, dim=1)
    if stepper.noise_dim:
        nz = stepper.NOISE[:len(feats)] if stepper.NOISE is not None and len(stepper.NOISE) >= len(feats) else \
            torch.randn(len(feats), stepper.noise_dim, device=feats.device)
        feats = torch.cat((feats, nz.to(feats.device)), dim=1)
    net, opt = stepper.net, getattr(stepper.opt, "opt", stepper.opt)
    if use_ema:
        saved = {k: v.detach().clone() for k, v in net.state_dict().items()}
        net.load_state_dict(opt.ema_state_dict())
    xr = generate(net, stepper.vq, feats)
    if use_ema:
        net.load_state_dict(saved)
    save_grid(xr, prefix + ".png", nrow=bs)
    save_grid(xr, f"{prefix}_{step:010d}.png", nrow=bs)


def test(model_path, text_or_path, *, nb_repeats=1, out_path="gen.png", images_per_row=None, seed=None,
         cdt=torch.float16, bpe_path=None, prior_path=None):
    """main.py:977-1061: prompts -> PNG grid; prior_path: a `train_prior` checkpoint whose flow maps the text embedding
    to an image-embedding sample before the mapper (main.py:1022-1023,1037-1040; prior.py).  `text_or_path`: "a|b|c", a `.txt` file with
    one prompt per line, a `.pkl` of token rows / features, or `synthetic:<n>[:seed]` token rows."""
    if seed is not None:
        torch.manual_seed(seed)
    net = load_model(model_path, cdt)
    config = net.config
    perceptor = load_clip_model(config.clip_model, path=config.get("clip_model_path"), cdt=cdt, fp8=bool(config.get("clip_fp8", False)))
    vq = load_vqgan_model(config.vqgan_config, config.vqgan_checkpoint, cdt)
    if text_or_path.startswith("synthetic:") or text_or_path.endswith(".pkl"):
        toks = load_dataset(text_or_path)
        toks = toks[0] if isinstance(toks, tuple) else toks
    else:
        from . import tokenizer
        texts = [t.strip() for t in open(text_or_path).readlines()] if text_or_path.endswith(".txt") else \
            text_or_path.split("|")
        toks = tokenizer.tokenize(texts, truncate=True, bpe_path=bpe_path)
    H = perceptor.encode_text(toks.cuda()).float() if toks.dtype == torch.long else toks.float().cuda()
    if config.get("normalize_input", False):
        H = torch.nn.functional.normalize(H, dim=1)
    H = H.repeat(nb_repeats, 1)
    if prior_path:                                                                # main.py:1037-1040
        from . import prior as _prior
        H = _prior.load_prior_model(prior_path).sample(H.view(len(H), -1, 1, 1)).view(len(H), -1)
    if config.noise_dim:                                                          # main.py:1041-1053
        bank = getattr(net, "NOISE", None)
        if bank is not None:
            bank = bank[:len(H)] if len(bank) > len(H) else bank[torch.randint(0, len(bank), (len(H),))]
            H = torch.cat((H, bank.to(H.device)), dim=1)
        else:
            H = torch.cat((H, torch.randn(len(H), config.noise_dim, device=H.device)), dim=1)
    xr = generate(net, vq, H)
    save_grid(xr, out_path, nrow=images_per_row if images_per_row else nb_repeats)
    return xr


def _cli(argv):
    if len(argv) >= 2 and argv[0] == "train":
        return train(argv[1])
    if len(argv) >= 3 and argv[0] == "test":
        import argparse
        ap = argparse.ArgumentParser(prog="main.py test")
        ap.add_argument("model_path")
        ap.add_argument("text_or_path")
        ap.add_argument("--nb-repeats", type=int, default=1)
        ap.add_argument("--out-path", default="gen.png")
        ap.add_argument("--images-per-row", type=int, default=None)
        ap.add_argument("--seed", type=int, default=None)
        ap.add_argument("--bpe-path", default=None)
        ap.add_argument("--prior-path", default=None)
        a = ap.parse_args(argv[1:])
        test(a.model_path, a.text_or_path, nb_repeats=a.nb_repeats, out_path=a.out_path, images_per_row=a.images_per_row,
             seed=a.seed, bpe_path=a.bpe_path, prior_path=a.prior_path)
        return 0
    if len(argv) >= 2 and argv[0] == "tokenize":
        tokenize(argv[1], *(argv[2:3] or ["tokenized.pkl"]))
        return 0
    if len(argv) >= 2 and argv[0] == "encode_text":
        encode_text(argv[1], *(argv[2:3] or ["text_features.pkl"]))
        return 0
    if len(argv) >= 2 and argv[0] == "encode_text_and_images":
        encode_text_and_images(argv[1], out=(argv[2:3] or ["features.pkl"])[0])
        return 0
    print("usage: python -m feed_forward_vqgan_clip_amd.main train <config.yaml> | test <model.th> <prompts> [...] | "
          "tokenize <prompts> [out.pkl] | encode_text <dataset> [out.pkl] | encode_text_and_images <folder> [out.pkl]",
          file=sys.stderr)
    return 2


if __name__ == "__main__":
    sys.exit(_cli(sys.argv[1:]) or 0)
