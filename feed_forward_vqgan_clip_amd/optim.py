"""Fused Adam over the mapper's flat parameter arena (reference: `optim.Adam(net.parameters(), lr)`
main.py:591, `clip_grad_norm_` :833-834, `CosineAnnealingLR` :702-709,836-837).

One ffvc_adam launch updates params, exp_avg, exp_avg_sq and rewrites the compute-dtype shadow;
the transposed shadows are refreshed right after.  state_dict()/load_state_dict() use
torch.optim.Adam's layout so `opt.th` files (main.py:911,592-596) are interchangeable.
"""
import math

import torch

from . import kernels as K
from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = list(params)
        arenas = {id(getattr(p, "_ffvc_arena", None)): getattr(p, "_ffvc_arena", None) for p in params}
        if len(arenas) != 1 or None in arenas.values():
            raise ValueError("FusedAdam: parameters must come from ONE prepared mapper (call net.prepare() first)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.arena = next(iter(arenas.values()))
        a = self.arena
        if len(params) != len(a.plist):
            raise ValueError("FusedAdam needs all parameters of the mapper")
        self._m = torch.zeros_like(a.params)
        self._v = torch.zeros_like(a.params)
        self._step = 0
        self.grad_scale = 1.0           # folded into the Adam kernel (1/world_size for summed all-reduce)
        for p in a.plist:
            o, n = a.param_range(p)
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self._m[o:o + n].view(p.shape),
                             "exp_avg_sq": self._v[o:o + n].view(p.shape)}

    def zero_grad(self, set_to_none=False):   # grads are views of one bucket: a single memset (main.py:825)
        self.arena.zero_grad()

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping folded into the update scale (clip_grad_norm_, main.py:833-834).
        Runs on the gradients as they are at call time (i.e. after the all-reduce in DP runs)."""
        ops.join_side_stream()
        ss = torch.zeros(1, dtype=torch.float32, device=self.arena.grads.device)
        K.sumsq(self.arena.grads, ss)
        total = math.sqrt(float(ss.item())) * abs(self.grad_scale)
        coef = min(1.0, max_norm / (total + 1e-6))
        self._clip_coef = coef
        return total

    @torch.no_grad()
    def step(self, closure=None):
        a = self.arena
        g = self.param_groups[0]
        ops.join_side_stream()          # weight gradients are produced on the side stream
        self._step += 1
        scale = self.grad_scale * getattr(self, "_clip_coef", 1.0)
        self._clip_coef = 1.0
        K.adam(a.params, a.grads, self._m, self._v, None if a.cdt == torch.float32 else a.shadow, g["lr"],
               g["betas"][0], g["betas"][1], g["eps"], self._step, scale)
        a.refresh(cast=False)
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))

    def load_state_dict(self, state_dict):
        sd = state_dict["state"]
        plist = self.arena.plist
        for i, p in enumerate(plist):
            if i in sd:
                st = sd[i]
                self.state[p]["exp_avg"].copy_(st["exp_avg"])
                self.state[p]["exp_avg_sq"].copy_(st["exp_avg_sq"])
                self._step = int(float(st["step"]))
        for k, v in state_dict["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max, eta_min=0) closed form (main.py:705-706)."""

    def __init__(self, optimizer, T_max, eta_min=0.0):
        self.opt, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = self.eta_min + (b - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
