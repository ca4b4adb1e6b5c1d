"""Oracle (CPU fp32) restatement of the prompt->latent mappers.  TEST INFRASTRUCTURE ONLY.

Functions take the reference's state_dict (SURVEY.md App. C key names) and return the same
tensors the reference modules return.  dropout must be 0 (stochastic otherwise).
"""
import torch
import torch.nn.functional as F


def _ln(x, sd, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _lin(x, sd, prefix, bias=True):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


# ---------------------------------------------------------------------------
# MLP-Mixer mapper — mlp_mixer_pytorch.py:70-91 (Mixer), :25-38 (MLPMixer),
# :7-14 (PreNormResidual), :16-23 (FeedForward).  GELU is the exact erf form
# (main.py:431-438 forces approximate="none").
# ---------------------------------------------------------------------------
def mixer_forward(sd, x, *, image_size, channels, depth):
    B, S, C = x.shape[0], image_size, channels
    h = _lin(x, sd, "proj")                                   # mlp_mixer_pytorch.py:85
    h = h.view(B, C, S, S)                                    # :86
    h = h.permute(0, 2, 3, 1).reshape(B, S * S, C)            # Rearrange 'b c h w -> b (h w) c', patch 1 (:31)
    h = _lin(h, sd, "mixer.1")                                # :32
    for i in range(2, depth + 2):
        p = f"mixer.{i}"
        # token mixing: Conv1d(k=1) treats tokens as channels (:28,34)
        n = _ln(h, sd, p + ".0.norm")
        t = F.conv1d(n, sd[p + ".0.fn.0.weight"], sd[p + ".0.fn.0.bias"])
        t = F.gelu(t)
        t = F.conv1d(t, sd[p + ".0.fn.3.weight"], sd[p + ".0.fn.3.bias"])
        h = h + t                                             # :14
        # channel mixing (:35)
        n = _ln(h, sd, p + ".1.norm")
        t = F.gelu(_lin(n, sd, p + ".1.fn.0"))
        h = h + _lin(t, sd, p + ".1.fn.3")
    h = _ln(h, sd, f"mixer.{depth + 2}")                      # :37
    h = _lin(h, sd, "final_proj")                             # :88
    return h.view(B, S, S, C).permute(0, 3, 1, 2)             # :89-90


# ---------------------------------------------------------------------------
# VitGAN mapper — vitgan.py:221-260 (Generator), :262-305 (SimpleGenerator),
# :120-135 (GEncoderBlock), :44-97 (Attention, generator branch), :24-41 (MLP),
# :8-21 (SLN).
# ---------------------------------------------------------------------------
def _sln(hl, w, sd, prefix):
    # vitgan.py:20-21: gamma * w * LN(hl) + beta * w, gamma/beta are (1,1,1) scalars
    return sd[prefix + ".gamma"] * w * _ln(hl, sd, prefix + ".ln") + sd[prefix + ".beta"] * w


def _vitgan_attention(x, sd, prefix, heads, dim):
    B, T, _ = x.shape
    qkv = F.linear(x, sd[prefix + ".to_qkv.weight"])          # vitgan.py:81 (no bias)
    dh = qkv.shape[-1] // (3 * heads)
    qkv = qkv.view(B, T, dh, 3, heads)                        # 'b t (d k h) -> k b h t d' (:82)
    q, k, v = (qkv[:, :, :, j, :].permute(0, 3, 1, 2) for j in range(3))
    att = torch.einsum("bhid,bhjd->bhij", q, k) * (dim ** -0.5)   # :90-91 (scale uses FULL dim, :65)
    att = att.softmax(dim=-1)
    o = torch.einsum("bhij,bhjd->bhid", att, v)
    o = o.permute(0, 2, 1, 3).reshape(B, T, heads * dh)       # 'b h t d -> b t (h d)' (:96)
    return _lin(o, sd, prefix + ".w_out")


def _vitgan_blocks(hl, x, sd, blocks, heads, dim):
    for i in range(blocks):
        p = f"Transformer_Encoder.blocks.{i}"
        hl_t = _vitgan_attention(_sln(hl, x, sd, p + ".norm1"), sd, p + ".attn", heads, dim) + hl  # :132
        m = F.gelu(_lin(_sln(hl_t, x, sd, p + ".norm2"), sd, p + ".mlp.linear1"))
        hl = _lin(m, sd, p + ".mlp.linear2") + hl_t           # :133
    return hl


def vitgan_forward(sd, noise, *, initialize_size, dim, blocks, num_heads, out_channels):
    T = initialize_size * 8
    x = _lin(noise, sd, "mlp").view(-1, T, dim)               # vitgan.py:254
    hl = _vitgan_blocks(sd["pos_emb1D"], x, sd, blocks, num_heads, dim)   # :255
    x = _sln(hl, x, sd, "sln_norm")                           # :256
    x = _lin(x, sd, "w_out.0")                                # :257
    return x.reshape(x.shape[0], out_channels, T, T)          # raw view (:258-259)


def simple_vitgan_forward(sd, noise, *, size, dim, blocks, num_heads, out_channels):
    N = size * size
    inp = _lin(noise, sd, "inp")                              # vitgan.py:297
    x = _lin(noise, sd, "mlp").view(-1, N, dim)               # :298
    inp_emb = inp.view(inp.shape[0], dim, N).permute(0, 2, 1) # :299
    hl = _vitgan_blocks(inp_emb + sd["pos_emb1D"], x, sd, blocks, num_heads, dim)   # :300
    x = _sln(hl, x, sd, "sln_norm")
    x = _lin(x, sd, "w_out.0")
    return x.view(x.shape[0], size, size, out_channels).permute(0, 3, 1, 2)       # :303


# ---------------------------------------------------------------------------
# x-transformer mapper — transformer.py:5-46 over x-transformers==0.19.1
# ContinuousTransformerWrapper(Decoder).  PARITY UNPINNED: x_transformers is not in
# /root/reference nor installed; restated from the published 0.19.1 source
# (SURVEY.md App. A.3).  All three input modes of transformer.py:29-40 (initial_proj / add_input).
# ---------------------------------------------------------------------------
def xtransformer_forward(sd, x, *, image_size, channels, dim, depth, heads, dim_head=64,
                         pos_scale=True, initial_proj=True, add_input=True):
    B, S = x.shape[0], image_size
    n = S * S
    nout = n
    t = "transformer"
    if initial_proj:
        h = _lin(x, sd, "proj").view(B, n, dim)               # transformer.py:30-31
    elif add_input:
        h = x.view(B, 1, -1).repeat(1, n, 1)                  # :34-36
    else:
        h = torch.cat((x.view(B, 1, -1), torch.zeros(B, n, x.shape[1], dtype=x.dtype)), dim=1)   # :38-40
        n = n + 1
    h = _lin(h, sd, t + ".project_in")                        # wrapper: Linear(dim_in, dim) (dim_in given)
    pos = sd[t + ".pos_emb.emb.weight"][:n]
    h = h + (pos * (dim ** -0.5) if pos_scale else pos)       # AbsolutePositionalEmbedding (scaled)
    causal = torch.ones(n, n, dtype=torch.bool, device=h.device).triu_(1)
    for j in range(depth):
        a = f"{t}.attn_layers.layers.{2 * j}"
        r = h
        y = _ln(h, sd, a + ".0")                              # pre-norm
        q = F.linear(y, sd[a + ".1.to_q.weight"]).view(B, n, heads, dim_head).transpose(1, 2)
        k = F.linear(y, sd[a + ".1.to_k.weight"]).view(B, n, heads, dim_head).transpose(1, 2)
        v = F.linear(y, sd[a + ".1.to_v.weight"]).view(B, n, heads, dim_head).transpose(1, 2)
        dots = torch.einsum("bhid,bhjd->bhij", q, k) * (dim_head ** -0.5)
        dots = dots.masked_fill(causal, -torch.finfo(dots.dtype).max)
        o = torch.einsum("bhij,bhjd->bhid", dots.softmax(dim=-1), v)
        o = o.transpose(1, 2).reshape(B, n, heads * dim_head)
        h = _lin(o, sd, a + ".1.to_out") + r
        f = f"{t}.attn_layers.layers.{2 * j + 1}"
        r = h
        y = _ln(h, sd, f + ".0")
        y = F.gelu(_lin(y, sd, f + ".1.net.0.0"))
        h = _lin(y, sd, f + ".1.net.2") + r
    h = _ln(h, sd, t + ".norm")
    h = _lin(h, sd, t + ".project_out")
    if n != nout:
        h = h[:, 1:]                                          # :42-43
    return h.reshape(B, S, S, channels).permute(0, 3, 1, 2)   # transformer.py:44-45
