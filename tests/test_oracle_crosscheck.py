"""Independent witnesses for oracle restatements whose upstream source is NOT in /root/reference (SURVEY.md §7 step 1,
VERDICT r2 #6).  CPU only.

* oracle/clip.py (follows cloob.py:170-255, 412-553; pinned to cloob.CLIP by the golden fixtures) against HuggingFace
  `transformers.CLIPModel` — a third implementation of the same published architecture — at the FULL ViT-B/32 dimensions with
  `hidden_act="quick_gelu"`, random weights, through a key mapping between the two state_dict layouts.
* oracle/vqgan.py `resnet_block` / `attn_block` (taming-transformers 0.0.6, absent) against the taming-derived VQ-VAE blocks
  that ship inside `transformers` (Chameleon's VQ-VAE encoder blocks are a port of taming's ResnetBlock / AttnBlock).
"""
import pytest
import torch

from oracle import clip as oclip

transformers = pytest.importorskip("transformers")


def _hf_clip(layers):
    from transformers import CLIPConfig, CLIPModel
    cfg = CLIPConfig(
        text_config=dict(vocab_size=49408, hidden_size=512, intermediate_size=2048, num_hidden_layers=layers, num_attention_heads=8,
                         max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=49407, bos_token_id=49406, pad_token_id=0),
        vision_config=dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=layers, num_attention_heads=12, image_size=224,
                           patch_size=32, hidden_act="quick_gelu"),
        projection_dim=512)
    torch.manual_seed(0)
    m = CLIPModel(cfg).eval()
    with torch.no_grad():                       # HF initialises biases / LayerNorm to constants: randomise everything
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
        for n, p in m.named_parameters():
            if "layer_norm" in n or "layrnorm" in n or "layernorm" in n:
                if n.endswith("weight"):
                    p.add_(1.0)
    return m


def _to_openai_keys(hf):
    """HF CLIPModel state_dict -> the OpenAI / cloob key layout the oracle (and the product) read."""
    h = hf.state_dict()
    sd = {}

    def tower(src, dst, n):
        for i in range(n):
            s, d = f"{src}.encoder.layers.{i}", f"{dst}.resblocks.{i}"
            sd[d + ".ln_1.weight"], sd[d + ".ln_1.bias"] = h[s + ".layer_norm1.weight"], h[s + ".layer_norm1.bias"]
            sd[d + ".ln_2.weight"], sd[d + ".ln_2.bias"] = h[s + ".layer_norm2.weight"], h[s + ".layer_norm2.bias"]
            sd[d + ".attn.in_proj_weight"] = torch.cat([h[s + f".self_attn.{x}_proj.weight"] for x in "qkv"])
            sd[d + ".attn.in_proj_bias"] = torch.cat([h[s + f".self_attn.{x}_proj.bias"] for x in "qkv"])
            sd[d + ".attn.out_proj.weight"], sd[d + ".attn.out_proj.bias"] = h[s + ".self_attn.out_proj.weight"], h[s + ".self_attn.out_proj.bias"]
            sd[d + ".mlp.c_fc.weight"], sd[d + ".mlp.c_fc.bias"] = h[s + ".mlp.fc1.weight"], h[s + ".mlp.fc1.bias"]
            sd[d + ".mlp.c_proj.weight"], sd[d + ".mlp.c_proj.bias"] = h[s + ".mlp.fc2.weight"], h[s + ".mlp.fc2.bias"]

    n = hf.config.vision_config.num_hidden_layers
    tower("vision_model", "visual.transformer", n)
    tower("text_model", "transformer", n)
    sd["visual.conv1.weight"] = h["vision_model.embeddings.patch_embedding.weight"]
    sd["visual.class_embedding"] = h["vision_model.embeddings.class_embedding"]
    sd["visual.positional_embedding"] = h["vision_model.embeddings.position_embedding.weight"]
    sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"] = h["vision_model.pre_layrnorm.weight"], h["vision_model.pre_layrnorm.bias"]
    sd["visual.ln_post.weight"], sd["visual.ln_post.bias"] = h["vision_model.post_layernorm.weight"], h["vision_model.post_layernorm.bias"]
    sd["visual.proj"] = h["visual_projection.weight"].t().contiguous()
    sd["token_embedding.weight"] = h["text_model.embeddings.token_embedding.weight"]
    sd["positional_embedding"] = h["text_model.embeddings.position_embedding.weight"]
    sd["ln_final.weight"], sd["ln_final.bias"] = h["text_model.final_layer_norm.weight"], h["text_model.final_layer_norm.bias"]
    sd["text_projection"] = h["text_projection.weight"].t().contiguous()
    return {k: v.detach().clone() for k, v in sd.items()}


def _feat(x):
    return x if isinstance(x, torch.Tensor) else getattr(x, "pooler_output", x[0])


def test_oracle_clip_matches_hf_transformers_at_vit_b32_dims():
    hf = _hf_clip(layers=3)                    # full widths / heads / patch / context, 3 of the 12 identical blocks (CPU time)
    sd = _to_openai_keys(hf)
    g = torch.Generator().manual_seed(1)
    img = torch.randn(2, 3, 224, 224, generator=g)
    tok = torch.zeros(3, 77, dtype=torch.long)
    for i, L in enumerate((5, 20, 76)):
        tok[i, 0] = 49406
        tok[i, 1:L] = torch.randint(1, 49000, (L - 1,), generator=g)
        tok[i, L] = 49407                      # EOT = highest id (cloob.py:536 argmax pooling)
    with torch.no_grad():
        want_i = _feat(hf.get_image_features(pixel_values=img))
        want_t = _feat(hf.get_text_features(input_ids=tok))
        got_i = oclip.encode_image(sd, img, heads=12)
        got_t = oclip.encode_text(sd, tok, heads=8)
    for got, want in ((got_i, want_i), (got_t, want_t)):
        assert got.shape == want.shape == (got.shape[0], 512)
        rel = ((got - want).norm() / want.norm()).item()
        assert rel < 2e-5, rel


def test_oracle_vqgan_decoder_matches_the_taming_derived_decoder_in_transformers():
    """`transformers.models.janus.JanusVQVAEDecoder` is a port of the taming Decoder (GroupNorm(32, eps 1e-6) + swish ResnetBlocks
    with nin_shortcut, single-head AttnBlock scaled by c^-0.5 with the softmax over keys, nearest-2x + conv Upsample, num_res_blocks + 1
    blocks per level, attention on the 16x16 level).  With the f16 layout (ch_mult (1,1,2,2,4), attn_resolutions (16,), resolution 256)
    the two architectures coincide, so the whole decoder restatement gets a second, independently written witness."""
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    from transformers.models.janus.modeling_janus import JanusVQVAEDecoder
    from oracle import vqgan as ovq

    cfg = dict(ch=32, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, attn_resolutions=(16,), resolution=256, z_channels=64, out_ch=3)
    jc = JanusVQVAEConfig(base_channels=32, channel_multiplier=[1, 1, 2, 2, 4], num_res_blocks=2, latent_channels=64, out_channels=3,
                          dropout=0.0, in_channels=3, double_latent=False, embed_dim=64, num_embeddings=128)
    torch.manual_seed(0)
    dec = JanusVQVAEDecoder(jc).eval()
    with torch.no_grad():
        for n, p in dec.named_parameters():
            p.copy_(torch.randn_like(p) * 0.05)
            if "norm" in n and n.endswith("weight"):
                p.add_(1.0)
    nlev = 5
    sd = {}
    for k, v in dec.state_dict().items():
        parts = k.split(".")
        if parts[0] == "up":                                   # janus builds the levels in forward order, taming indexes them by level
            parts[1] = str(nlev - 1 - int(parts[1]))
        sd["decoder." + ".".join(parts)] = v.detach().clone()
    assert "decoder.up.4.attn.2.proj_out.weight" in sd and "decoder.up.0.upsample.conv.weight" not in sd
    z = torch.randn(2, 64, 2, 2, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = dec(z.clone())
        got = ovq.decoder_forward(sd, z, cfg)
    assert got.shape == want.shape == (2, 3, 32, 32)
    assert ((got - want).norm() / want.norm()).item() < 1e-5


# ----------------------------------------------------------------------------- x-transformer mapper (transformer.py:5-46)
def _xt_state_dict(input_dim, image_size, channels, dim, depth, heads, initial_proj, extra_pos=0, seed=0):
    """Random weights in the key layout of ContinuousTransformerWrapper(Decoder) 0.19.1 as oracle/mappers.py reads it."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * 0.08     # noqa: E731
    n, inner, t = image_size * image_size + extra_pos, heads * 64, "transformer"
    sd = {}
    if initial_proj:
        sd["proj.weight"], sd["proj.bias"] = r(image_size * image_size * dim, input_dim), r(image_size * image_size * dim)
    din = dim if initial_proj else input_dim
    sd[t + ".project_in.weight"], sd[t + ".project_in.bias"] = r(dim, din), r(dim)
    sd[t + ".pos_emb.emb.weight"] = r(n, dim)
    for j in range(depth):
        a, f = f"{t}.attn_layers.layers.{2 * j}", f"{t}.attn_layers.layers.{2 * j + 1}"
        for p in (a, f):
            sd[p + ".0.weight"], sd[p + ".0.bias"] = 1.0 + r(dim), r(dim)
        for q in ("to_q", "to_k", "to_v"):
            sd[f"{a}.1.{q}.weight"] = r(inner, dim)
        sd[a + ".1.to_out.weight"], sd[a + ".1.to_out.bias"] = r(dim, inner), r(dim)
        sd[f + ".1.net.0.0.weight"], sd[f + ".1.net.0.0.bias"] = r(4 * dim, dim), r(4 * dim)
        sd[f + ".1.net.2.weight"], sd[f + ".1.net.2.bias"] = r(dim, 4 * dim), r(dim)
    sd[t + ".norm.weight"], sd[t + ".norm.bias"] = 1.0 + r(dim), r(dim)
    sd[t + ".project_out.weight"], sd[t + ".project_out.bias"] = r(channels, dim), r(channels)
    return sd


def test_oracle_xtransformer_matches_hf_gpt2_blocks():
    """Witness for oracle/mappers.py::xtransformer_forward (x-transformers 0.19.1 is absent): the Decoder of that release is a stack of
    pre-norm causal self-attention + GELU feed-forward blocks with a final LayerNorm — the GPT-2 block.  HuggingFace's GPT2Model
    (an independently written implementation: fused c_attn Conv1D, its own causal masking / softmax path, erf-GELU via
    `activation_function="gelu"`) is loaded with the same weights (q/k/v bias-free -> zero c_attn bias; learned positions scaled by
    dim^-0.5 as AbsolutePositionalEmbedding does) at dim = heads * 64, where the two architectures coincide, and fed the oracle's
    own `project_in` output through `inputs_embeds`.  Everything between project_in and project_out is then GPT-2's code."""
    from transformers import GPT2Config, GPT2Model
    from oracle import mappers as omap
    S, C, dim, depth, heads, idim = 4, 16, 128, 3, 2, 40
    n = S * S
    sd = _xt_state_dict(idim, S, C, dim, depth, heads, initial_proj=True, seed=3)
    cfg = GPT2Config(vocab_size=8, n_positions=n, n_embd=dim, n_layer=depth, n_head=heads, n_inner=4 * dim, activation_function="gelu",
                     resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, layer_norm_epsilon=1e-5, scale_attn_weights=True,
                     scale_attn_by_inverse_layer_idx=False, reorder_and_upcast_attn=False)
    gpt = GPT2Model(cfg).eval()
    t = "transformer"
    with torch.no_grad():
        gpt.wpe.weight.copy_(sd[t + ".pos_emb.emb.weight"] * dim ** -0.5)
        for j, blk in enumerate(gpt.h):
            a, f = f"{t}.attn_layers.layers.{2 * j}", f"{t}.attn_layers.layers.{2 * j + 1}"
            blk.ln_1.weight.copy_(sd[a + ".0.weight"]); blk.ln_1.bias.copy_(sd[a + ".0.bias"])
            blk.ln_2.weight.copy_(sd[f + ".0.weight"]); blk.ln_2.bias.copy_(sd[f + ".0.bias"])
            blk.attn.c_attn.weight.copy_(torch.cat([sd[f"{a}.1.to_{x}.weight"].t() for x in "qkv"], dim=1))   # Conv1D: [in, out]
            blk.attn.c_attn.bias.zero_()
            blk.attn.c_proj.weight.copy_(sd[a + ".1.to_out.weight"].t()); blk.attn.c_proj.bias.copy_(sd[a + ".1.to_out.bias"])
            blk.mlp.c_fc.weight.copy_(sd[f + ".1.net.0.0.weight"].t()); blk.mlp.c_fc.bias.copy_(sd[f + ".1.net.0.0.bias"])
            blk.mlp.c_proj.weight.copy_(sd[f + ".1.net.2.weight"].t()); blk.mlp.c_proj.bias.copy_(sd[f + ".1.net.2.bias"])
        gpt.ln_f.weight.copy_(sd[t + ".norm.weight"]); gpt.ln_f.bias.copy_(sd[t + ".norm.bias"])
    x = torch.randn(3, idim, generator=torch.Generator().manual_seed(4))
    F = torch.nn.functional
    with torch.no_grad():
        got = omap.xtransformer_forward(sd, x, image_size=S, channels=C, dim=dim, depth=depth, heads=heads)
        h = F.linear(x, sd["proj.weight"], sd["proj.bias"]).view(3, n, dim)
        h = F.linear(h, sd[t + ".project_in.weight"], sd[t + ".project_in.bias"])
        hid = gpt(inputs_embeds=h).last_hidden_state
        want = F.linear(hid, sd[t + ".project_out.weight"], sd[t + ".project_out.bias"]).view(3, S, S, C).permute(0, 3, 1, 2)
    assert got.shape == want.shape == (3, C, S, S)
    assert ((got - want).norm() / want.norm()).item() < 2e-5


@pytest.mark.parametrize("mode", ["initial_proj", "add_input", "prefix_token"])
def test_oracle_xtransformer_matches_a_module_built_restatement(mode):
    """Second, differently built restatement for the case GPT-2 cannot express (inner width heads*64 != dim: cfg4 runs dim 256 with
    6 heads) and for the three input modes of transformer.py:29-43: torch.nn modules (nn.LayerNorm / nn.Linear / nn.GELU) and
    torch's fused `scaled_dot_product_attention(is_causal=True)` instead of the oracle's explicit einsum / masked_fill / softmax."""
    from torch import nn
    from oracle import mappers as omap
    S, C, dim, depth, heads, idim = 3, 8, 48, 2, 3, 20
    initial_proj, add_input = mode == "initial_proj", mode != "prefix_token"
    n = S * S
    L = n + (0 if (initial_proj or add_input) else 1)
    sd = _xt_state_dict(idim, S, C, dim, depth, heads, initial_proj, extra_pos=L - n, seed=5)
    t = "transformer"

    def lin(prefix, bias=True):
        w = sd[prefix + ".weight"]
        m = nn.Linear(w.shape[1], w.shape[0], bias=bias)
        m.weight.data.copy_(w)
        if bias:
            m.bias.data.copy_(sd[prefix + ".bias"])
        return m

    def ln(prefix):
        m = nn.LayerNorm(dim)
        m.weight.data.copy_(sd[prefix + ".weight"]); m.bias.data.copy_(sd[prefix + ".bias"])
        return m

    x = torch.randn(2, idim, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        got = omap.xtransformer_forward(sd, x, image_size=S, channels=C, dim=dim, depth=depth, heads=heads, initial_proj=initial_proj,
                                        add_input=add_input)
        if initial_proj:
            h = lin("proj")(x).view(2, n, dim)
        elif add_input:
            h = x[:, None, :].expand(2, n, idim)
        else:
            h = torch.cat([x[:, None, :], x.new_zeros(2, n, idim)], dim=1)
        h = lin(t + ".project_in")(h) + sd[t + ".pos_emb.emb.weight"][:L] / dim ** 0.5
        for j in range(depth):
            a, f = f"{t}.attn_layers.layers.{2 * j}", f"{t}.attn_layers.layers.{2 * j + 1}"
            y = ln(a + ".0")(h)
            q, k, v = (lin(f"{a}.1.to_{c}", bias=False)(y).view(2, L, heads, 64).permute(0, 2, 1, 3) for c in "qkv")
            o = torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)       # default scale = 64^-0.5
            h = h + lin(a + ".1.to_out")(o.permute(0, 2, 1, 3).reshape(2, L, heads * 64))
            h = h + nn.Sequential(ln(f + ".0"), lin(f + ".1.net.0.0"), nn.GELU(), lin(f + ".1.net.2"))(h)
        out = lin(t + ".project_out")(ln(t + ".norm")(h))
        want = out[:, L - n:].reshape(2, S, S, C).permute(0, 3, 1, 2)
    assert got.shape == want.shape
    assert ((got - want).norm() / want.norm()).item() < 2e-5


# ----------------------------------------------------------------------------- kornia 0.5.10 restatement: PIL / colorsys witnesses
#  oracle/kornia_aug.py is built on torch's grid_sample.  PIL's Image.transform (its own C resampler, its own
# pixel-centre convention) evaluated with the coefficient map DERIVED BY HAND from kornia's two conventions must sample the same
# values wherever no padding rule is involved; colorsys is the textbook HSV transform.
def _pil_warp(img, method, data):
    """img (3,H,W) float tensor -> PIL bilinear transform of every channel ('F' mode: float32 pixels)."""
    from PIL import Image
    H, W = img.shape[-2:]
    out = [torch.from_numpy(__import__("numpy").array(
        Image.fromarray(c.numpy().astype("float32"), mode="F").transform((W, H), method, data, resample=Image.BILINEAR))) for c in img]
    return torch.stack(out)


def _interior(src_x, src_y, S, margin=1.5):
    return (src_x > margin) & (src_x < S - 1 - margin) & (src_y > margin) & (src_y < S - 1 - margin)


def test_kornia_warp_affine_convention_matches_pil():
    """warp_affine(M, align_corners=False) as kornia 0.5.10 runs it: the pixel matrix is normalised with the align_corners=TRUE
    map (normalize_homography) and sampled with align_corners=FALSE.  By hand: output index i reads source index
        T2(M^-1(T1(i))),  T1(i) = (i + .5)(W-1)/W,  T2(a) = a W/(W-1) - .5
    PIL's AFFINE reads source index A (i + .5) + c - .5  ->  A = R, c = t W/(W-1) for M^-1 = [R | t]."""
    from PIL import Image
    from oracle import kornia_aug as ka
    S = 48
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float64), torch.arange(S, dtype=torch.float64), indexing="ij")
    img = torch.stack([torch.sin(xx / 5) * torch.cos(yy / 7), (xx * yy) / S ** 2, torch.rand(S, S, generator=g, dtype=torch.float64)]).float()
    for angle, ty in ((11.0, 3.2), (-14.0, -4.1), (0.0, 2.5)):
        M = ka.get_affine_matrix2d(torch.tensor([angle]), torch.tensor([[0.0, ty]]), torch.tensor([[(S - 1) / 2.0, (S - 1) / 2.0]]))
        want = ka.warp_affine(img[None], M, padding_mode="border")[0]
        Mi = torch.linalg.inv(M[0])
        s = S / (S - 1.0)
        data = (Mi[0, 0].item(), Mi[0, 1].item(), Mi[0, 2].item() * s, Mi[1, 0].item(), Mi[1, 1].item(), Mi[1, 2].item() * s)
        got = _pil_warp(img, Image.AFFINE, data)
        sx = data[0] * (xx + .5) + data[1] * (yy + .5) + data[2] - .5
        sy = data[3] * (xx + .5) + data[4] * (yy + .5) + data[5] - .5
        m = _interior(sx, sy, S)
        assert m.float().mean() > 0.4
        assert (got - want)[:, m].abs().max().item() < 2e-5, (angle, ty)
        # and the map is NOT the naive "M^-1 in pixel coordinates" one: that version misses by a visible margin
        naive = _pil_warp(img, Image.AFFINE, (Mi[0, 0].item(), Mi[0, 1].item(), Mi[0, 2].item() + .5 - .5 * (Mi[0, 0] + Mi[0, 1]).item(),
                                              Mi[1, 0].item(), Mi[1, 1].item(), Mi[1, 2].item() + .5 - .5 * (Mi[1, 0] + Mi[1, 1]).item()))
        assert (naive - want)[:2, m].abs().max().item() > 1e-3


def test_kornia_warp_perspective_convention_matches_pil():
    """warp_perspective(M, align_corners=False): the destination grid is linspace(-1, 1, W) (align_corners=TRUE style, pixel i at
    2 i/(W-1) - 1), the sampling align_corners=FALSE: output index i reads source index  T2(M^-1(i)).
    PIL's PERSPECTIVE reads  P (i + .5) - .5  ->  P = diag(s, s, 1) . M^-1 . translate(-.5),  s = W/(W-1)."""
    from PIL import Image
    from oracle import kornia_aug as ka
    S = 48
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.arange(S, dtype=torch.float64), torch.arange(S, dtype=torch.float64), indexing="ij")
    img = torch.stack([torch.sin(xx / 4) + torch.cos(yy / 6), (xx - yy) / S, torch.rand(S, S, generator=g, dtype=torch.float64)]).float()
    start = torch.tensor([[[0.0, 0.0], [S - 1.0, 0.0], [S - 1.0, S - 1.0], [0.0, S - 1.0]]], dtype=torch.float64)
    sign = torch.tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]], dtype=torch.float64)
    for seed in (2, 3, 4):
        rv = torch.rand(1, 4, 2, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        end = start + 0.7 * S / 2 * rv * sign[None]
        M = ka.get_perspective_transform(start, end)
        want = ka.warp_perspective(img[None], M)[0]
        s = S / (S - 1.0)
        P = torch.diag(torch.tensor([s, s, 1.0], dtype=torch.float64)) @ torch.linalg.inv(M[0]) @ \
            torch.tensor([[1.0, 0, -.5], [0, 1.0, -.5], [0, 0, 1.0]], dtype=torch.float64)
        P = P / P[2, 2]
        data = tuple(P.reshape(-1)[:8].tolist())
        got = _pil_warp(img, Image.PERSPECTIVE, data)
        den = data[6] * (xx + .5) + data[7] * (yy + .5) + 1
        sx = (data[0] * (xx + .5) + data[1] * (yy + .5) + data[2]) / den - .5
        sy = (data[3] * (xx + .5) + data[4] * (yy + .5) + data[5]) / den - .5
        m = _interior(sx, sy, S)
        assert m.float().mean() > 0.2
        assert (got - want)[:, m].abs().max().item() < 5e-5, seed
        # a destination pixel whose source lies well outside the image is zero in both (kornia: zeros padding)
        far = (sx < -1.5) | (sx > S + .5) | (sy < -1.5) | (sy > S + .5)
        if far.any():
            assert want[:, far].abs().max().item() == 0.0 and got[:, far].abs().max().item() == 0.0


def test_kornia_hsv_round_trip_matches_colorsys():
    """kornia/color/hsv.py as restated (hue in radians, s = delta / (v + 1e-6)) against the standard library's colorsys on random and
    on degenerate pixels (greys, primaries, ties between channels)."""
    import colorsys
    import math
    from oracle import kornia_aug as ka
    g = torch.Generator().manual_seed(0)
    px = torch.rand(200, 3, generator=g, dtype=torch.float64)
    special = torch.tensor([[0, 0, 0], [1, 1, 1], [.5, .5, .5], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1], [1, 0, 1],
                            [.3, .3, .9], [.9, .3, .3], [.2, .8, .8]], dtype=torch.float64)
    px = torch.cat([px, special])
    img = px.t().reshape(1, 3, -1, 1)
    hsv = ka.rgb_to_hsv(img)
    for i, (r, gg, b) in enumerate(px.tolist()):
        h, s, v = colorsys.rgb_to_hsv(r, gg, b)
        assert abs(hsv[0, 2, i, 0].item() - v) < 1e-12
        assert abs(hsv[0, 1, i, 0].item() - s) < 2e-6 * max(1.0, s / max(v, 1e-3)) + (1e-4 if v < 1e-2 else 0)      # the eps in the quotient
        if s > 1e-9:
            dh = abs(hsv[0, 0, i, 0].item() / (2 * math.pi) - h)
            assert min(dh, 1 - dh) < 1e-9, (i, r, gg, b)
    # hsv -> rgb: the piecewise form equals colorsys for hues / saturations / values over the whole cube
    hs = torch.rand(300, 3, generator=g, dtype=torch.float64)
    back = ka.hsv_to_rgb(torch.stack([hs[:, 0] * 2 * math.pi, hs[:, 1], hs[:, 2]]).reshape(1, 3, -1, 1))
    for i, (h, s, v) in enumerate(hs.tolist()):
        want = colorsys.hsv_to_rgb(h, s, v)
        assert max(abs(back[0, c, i, 0].item() - want[c]) for c in range(3)) < 1e-12
    # adjust_hue / adjust_saturation: the composition the ColorJitter applies, against colorsys per pixel
    shift, fac = torch.tensor([0.07]), torch.tensor([1.08])
    out_h = ka.adjust_hue(img, shift * 2 * math.pi)
    out_s = ka.adjust_saturation(img, fac)
    for i, (r, gg, b) in enumerate(px.tolist()):
        h, s, v = colorsys.rgb_to_hsv(r, gg, b)
        if s > 1e-6 and v > 1e-2:
            want = colorsys.hsv_to_rgb((h + 0.07) % 1.0, s, v)
            assert max(abs(out_h[0, c, i, 0].item() - want[c]) for c in range(3)) < 5e-6
            want = colorsys.hsv_to_rgb(h, min(1.0, s * 1.08), v)
            assert max(abs(out_s[0, c, i, 0].item() - want[c]) for c in range(3)) < 5e-6


# ----------------------------------------------------------------------------- net2net prior: a module-built FORWARD flow
def test_prior_reverse_inverts_a_module_built_forward_flow():
    """ oracle/prior.py::reverse restates the SAMPLING direction of net2net's ConditionalFlatCouplingFlow
    (main.py:1447-1462; net2net absent).  Its witness: the published FORWARD direction, built from torch.nn modules the way the
    package builds it — ActNorm (h = scale * (x + loc)), InvLeakyRelu(0.9) (h = x * [1 | alpha]), the conditional double coupling
    (x1' = x1 * exp(s(x0, c)) + t(x0, c), halves swapped before the second pair), Shuffle (x[:, forward_idx]) — loaded from the same
    state_dict.  reverse(forward(x)) must be x, for every flow depth, and forward must actually move x."""
    from torch import nn
    from feed_forward_vqgan_clip_amd import prior as fprior
    from oracle import prior as oprior

    class FC(nn.Module):                                       # BasicFullyConnectedNet
        def __init__(self, dim, depth, hidden, out, tanh):
            super().__init__()
            layers = [nn.Linear(dim, hidden), nn.LeakyReLU()]
            for _ in range(depth):
                layers += [nn.Linear(hidden, hidden), nn.LeakyReLU()]
            layers.append(nn.Linear(hidden, out))
            if tanh:
                layers.append(nn.Tanh())
            self.main = nn.Sequential(*layers)

        def forward(self, x):
            return self.main(x)

    class Coupling(nn.Module):                                 # ConditionalDoubleVectorCouplingBlock.forward
        def __init__(self, C, E, hidden, depth):
            super().__init__()
            self.s = nn.ModuleList([FC(C // 2 + E, depth, hidden, C // 2, True) for _ in range(2)])
            self.t = nn.ModuleList([FC(C // 2 + E, depth, hidden, C // 2, False) for _ in range(2)])

        def forward(self, x, xc):
            for i in range(2):
                if i % 2 != 0:
                    x = torch.cat(torch.chunk(x, 2, dim=1)[::-1], dim=1)
                x0, x1 = torch.chunk(x, 2, dim=1)
                ci = torch.cat((x0, xc), dim=1)
                x = torch.cat((x0, x1 * self.s[i](ci).exp() + self.t[i](ci)), dim=1)
            return x

    class ActNorm(nn.Module):
        def __init__(self, C):
            super().__init__()
            self.loc, self.scale = nn.Parameter(torch.zeros(1, C, 1, 1)), nn.Parameter(torch.ones(1, C, 1, 1))
            self.register_buffer("initialized", torch.tensor(1, dtype=torch.uint8))

        def forward(self, x):
            return self.scale.reshape(1, -1) * (x + self.loc.reshape(1, -1))

    class Shuffle(nn.Module):
        def __init__(self, C):
            super().__init__()
            self.register_buffer("forward_shuffle_idx", torch.arange(C))
            self.register_buffer("backward_shuffle_idx", torch.arange(C))

        def forward(self, x):
            return x[:, self.forward_shuffle_idx]

    class Block(nn.Module):                                    # ConditionalFlatDoubleCouplingFlowBlock.forward
        def __init__(self, C, E, hidden, depth):
            super().__init__()
            self.norm_layer, self.coupling, self.shuffle = ActNorm(C), Coupling(C, E, hidden, depth), Shuffle(C)

        def forward(self, x, xc):
            h = self.norm_layer(x)
            h = h * ((h >= 0).to(h) + (h < 0).to(h) * 0.9)    # InvLeakyRelu(alpha=0.9).forward
            return self.shuffle(self.coupling(h, xc))

    class Flow(nn.Module):                                     # ConditionalFlatCouplingFlow.forward
        def __init__(self, C, D, E, hidden, depth, n_flows):
            super().__init__()
            self.embedder = FC(D, 2, 256, E, False)
            self.sub_layers = nn.ModuleList([Block(C, E, hidden, depth) for _ in range(n_flows)])

        def forward(self, x, cond):
            emb = self.embedder(cond)
            for blk in self.sub_layers:
                x = blk(x, emb)
            return x

    C, D, E, hidden, depth = 32, 24, 16, 48, 2
    for n_flows in (1, 3):
        sd = fprior.random_state_dict(C, D, E, hidden, depth, n_flows, seed=5 + n_flows)
        flow = Flow(C, D, E, hidden, depth, n_flows).double()
        flow.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, strict=True)
        g = torch.Generator().manual_seed(n_flows)
        x = torch.randn(7, C, generator=g, dtype=torch.float64)
        cond = torch.randn(7, D, generator=g, dtype=torch.float64)
        with torch.no_grad():
            z = flow(x, cond)
            back = oprior.reverse(sd, z.float(), cond.float(), n_flows)        # (the oracle computes in fp32)
        assert (z - x).abs().max().item() > 0.1                 # the flow does something
        assert (back.double() - x).abs().max().item() < 5e-5, n_flows
