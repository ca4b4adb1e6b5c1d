"""Oracle (CPU fp32) restatement of the VQGAN f16 decoder.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in taming-transformers-rom1504==0.0.6
(requirements.txt:2; call sites main.py:29-31,84-103,141-142), which is neither in
/root/reference nor installed here.  This restates its published algorithm
(taming/modules/diffusionmodules/model.py: Decoder, ResnetBlock, AttnBlock, Upsample,
Normalize, nonlinearity; taming/models/vqgan.py: VQModel.decode, post_quant_conv) as
summarised in SURVEY.md App. A.1, with the upstream state_dict key names so that a real
`vqgan_imagenet_f16_16384.ckpt` state_dict can be fed in unchanged.
"""
import torch
import torch.nn.functional as F

# ddconfig of vqgan_imagenet_f16_16384.yaml
F16_16384 = dict(ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,),
                 resolution=256, z_channels=256, out_ch=3, embed_dim=256, n_embed=16384)


def _gn(x, sd, p):
    # Normalize(): GroupNorm(32, c, eps=1e-6, affine=True)
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(x, sd, p, padding):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=padding)


def resnet_block(x, sd, p):
    h = _conv(_swish(_gn(x, sd, p + ".norm1")), sd, p + ".conv1", 1)
    h = _conv(_swish(_gn(h, sd, p + ".norm2")), sd, p + ".conv2", 1)   # dropout p=0
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv(x, sd, p + ".nin_shortcut", 0)
    return x + h


def attn_block(x, sd, p):
    h = _gn(x, sd, p + ".norm")
    q, k, v = (_conv(h, sd, p + "." + n, 0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)       # b, hw, c
    k = k.reshape(b, c, hh * ww)                         # b, c, hw
    w_ = torch.bmm(q, k) * (int(c) ** -0.5)              # b, hw_q, hw_k
    w_ = F.softmax(w_, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(h, sd, p + ".proj_out", 0)


def decoder_forward(sd, z, cfg=F16_16384, prefix="decoder"):
    """taming Decoder.forward (temb=None). z: (B, z_channels, S, S) -> (B, out_ch, 16S, 16S)."""
    num_levels = len(cfg["ch_mult"])
    nrb = cfg["num_res_blocks"]
    p = prefix
    h = _conv(z, sd, p + ".conv_in", 1)
    h = resnet_block(h, sd, p + ".mid.block_1")
    h = attn_block(h, sd, p + ".mid.attn_1")
    h = resnet_block(h, sd, p + ".mid.block_2")
    curr_res = cfg["resolution"] // 2 ** (num_levels - 1)
    for i_level in reversed(range(num_levels)):
        has_attn = curr_res in cfg["attn_resolutions"]      # decided from CONFIG resolution, not runtime size
        for i_block in range(nrb + 1):
            h = resnet_block(h, sd, f"{p}.up.{i_level}.block.{i_block}")
            if has_attn:
                h = attn_block(h, sd, f"{p}.up.{i_level}.attn.{i_block}")
        if i_level != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(h, sd, f"{p}.up.{i_level}.upsample.conv", 1)
            curr_res *= 2
    h = _swish(_gn(h, sd, p + ".norm_out"))
    return _conv(h, sd, p + ".conv_out", 1)


def decode(sd, z_q, cfg=F16_16384):
    """VQModel.decode(quant) = decoder(post_quant_conv(quant))."""
    return decoder_forward(sd, _conv(z_q, sd, "post_quant_conv", 0), cfg)
