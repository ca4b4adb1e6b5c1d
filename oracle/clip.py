"""Oracle (CPU fp32) restatement of the CLIP ViT image tower + text tower.  TEST INFRASTRUCTURE ONLY.

The reference loads the perceptor from clip-anytorch==2.2.0 (requirements.txt:3,
main.py:33-34,1332), absent here; the SAME architecture is stated in-repo at
cloob.py:170-255 (LayerNorm/QuickGELU/ResidualAttentionBlock/Transformer/VisualTransformer)
and cloob.py:412-553 (CLIP.encode_image / encode_text), which is what this file follows and
what tools/gen_golden.py pins it against (cloob.CLIP, random weights).
State_dict keys: SURVEY.md App. C (identical to upstream clip.model.CLIP).
"""
import torch
import torch.nn.functional as F


def _ln(x, sd, p):
    # cloob.py:170-176: LayerNorm computed in fp32, eps 1e-5
    return F.layer_norm(x.float(), (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5).to(x.dtype)


def _quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)                       # cloob.py:179-181


def _mha(x, sd, p, heads, mask):
    """nn.MultiheadAttention(d, heads)(x, x, x, attn_mask=mask) for x of shape (N, L, D) (cloob.py:187,199-200)."""
    N, L, D = x.shape
    hd = D // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(N, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.view(N, L, heads, hd).transpose(1, 2)
    v = v.view(N, L, heads, hd).transpose(1, 2)
    att = q @ k.transpose(-1, -2)
    if mask is not None:
        att = att + mask
    o = att.softmax(dim=-1) @ v
    o = o.transpose(1, 2).reshape(N, L, D)
    return F.linear(o, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def _resblock(x, sd, p, heads, mask, quick_gelu=True):
    x = x + _mha(_ln(x, sd, p + ".ln_1"), sd, p + ".attn", heads, mask)          # cloob.py:203
    h = F.linear(_ln(x, sd, p + ".ln_2"), sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
    # open_clip architectures without the `-quickgelu` suffix (main.py:1323-1329) use nn.GELU() here [upstream]
    h = F.linear(_quick_gelu(h) if quick_gelu else F.gelu(h), sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"])
    return x + h                                                                   # cloob.py:204


def _n_layers(sd, prefix):
    n = 0
    while f"{prefix}.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


def encode_image(sd, image, heads=None, quick_gelu=True):
    """cloob.py:236-255 (VisualTransformer.forward). image: (N,3,R,R), already mean/std normalised."""
    w = sd["visual.conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    heads = heads or width // 64                                                   # cloob.py:446
    x = F.conv2d(image, w, None, stride=patch)                                     # :237
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)                          # :238-239
    cls = sd["visual.class_embedding"].to(x.dtype).expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1)                                                 # :240-243
    x = x + sd["visual.positional_embedding"]                                      # :244
    x = _ln(x, sd, "visual.ln_pre")                                                # :245
    for i in range(_n_layers(sd, "visual.transformer")):
        x = _resblock(x, sd, f"visual.transformer.resblocks.{i}", heads, None, quick_gelu)     # :247-249
    x = _ln(x[:, 0, :], sd, "visual.ln_post")                                      # :251
    return x @ sd["visual.proj"]                                                   # :253-254


def encode_text(sd, text, heads=None, quick_gelu=True):
    """cloob.py:525-538 (CLIP.encode_text). text: int64 (B, 77)."""
    x = sd["token_embedding.weight"][text]                                         # :526
    L = x.shape[1]
    x = x + sd["positional_embedding"][:L]                                         # :528
    width = x.shape[-1]
    heads = heads or width // 64                                                   # transformer_heads = width/64 upstream (8 for ViT-B/32)
    mask = torch.full((L, L), float("-inf"), device=x.device).triu_(1)             # cloob.py:510-516
    for i in range(_n_layers(sd, "transformer")):
        x = _resblock(x, sd, f"transformer.resblocks.{i}", heads, mask, quick_gelu)
    x = _ln(x, sd, "ln_final")                                                     # :532
    eot = text.argmax(dim=-1)                                                      # :536 (EOT = highest id)
    return x[torch.arange(x.shape[0]), eot] @ sd["text_projection"]
