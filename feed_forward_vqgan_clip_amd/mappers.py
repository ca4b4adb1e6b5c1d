"""Prompt -> VQGAN-latent mappers (the only trainable part of the step), MI355X-native.

Same constructor arguments, forward contract `(B, clip_dim+noise_dim) -> (B, C, S, S)` and
state_dict key layout as the reference classes, so released `.th`
checkpoints load unchanged; the arithmetic runs in ffvc HIP kernels (ops.py).  torch.nn
layers are used ONLY as parameter holders / default initialisers — their forward is never called.
"""
import torch
from torch import nn

from . import ops
from .arena import ParamArena
from .kernels import ACT_GELU


class _MapperBase(nn.Module):
    """Shared arena / shadow management."""

    cdt = torch.bfloat16

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
            raise NotImplementedError("ffvc Mixer supports patch_size=1 (the only value main.py:479-488 passes)")
        if dropout:
            raise NotImplementedError("dropout > 0 is not implemented in the HIP path (configs use dropout: 0)")
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

    def forward(self, x):
        self._arena()
        cdt, f32 = self.cdt, torch.float32
        B, S, C = x.shape[0], self.image_size, self.channels
        h = ops.linear(ops.cast(x.float(), cdt), self._w_proj)                  # mlp_mixer_pytorch.py:85
        h = ops.transpose_last2(h.view(B, C, S * S))                            # :86 + Rearrange (:31) -> (B, S*S, C)
        h = ops.linear(h, self._w_embed, out_dtype=f32)                         # :32  fp32 residual stream
        for (n1, t1, t2, n2, c1, c2) in self._blocks:
            hn, hid = ops.layernorm_fork(h, n1.weight, n1.bias, cdt)
            h = ops.token_mlp(hn, t1, t2, residual=hid, out_dtype=f32)          # :34 (+ residual :14)
            hn, hid = ops.layernorm_fork(h, n2.weight, n2.bias, cdt)
            h = ops.mlp(hn, c1, c2, ACT_GELU, residual=hid, out_dtype=f32)      # :35
        fin = self.mixer[self.depth + 2]
        hn = ops.layernorm(h, fin.weight, fin.bias, cdt)                        # :37
        z = ops.linear(hn, self._w_final, out_dtype=f32)                        # :88
        return z.view(B, S, S, C).permute(0, 3, 1, 2)                           # :89-90 (non-contiguous, like the reference)
