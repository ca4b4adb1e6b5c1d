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
        self.loss_scale = 1.0           # gradients in the bucket are loss_scale x the true ones (f16 backward)
        self._ema, self._ema_decay, self._ema_updates = None, 0.0, 0
        self._clip = None               # device [coef, total_norm] of the pending clip_grad_norm_
        # overflow guard of the loss-scaled f16 backward: the Adam kernel skips (and counts) non-finite gradient elements, the
        # host reads the counter at its logging synchronisations (check_overflow) and backs the scale off / regrows it
        self._bad = torch.zeros(1, dtype=torch.int32, device=a.params.device)
        # captured-step mode (main.TrainStep.enable_graph): the per-step scalars of the update live in device memory — a pinned
        # host mirror is refreshed and copied over before every replay (graph_pre_step) — and the host bookkeeping of step()
        # is done there instead of inside the (recorded, not executed) capture call
        self._hyper, self._capturing = None, False
        self._good_steps = 0
        self.scale_growth_interval, self.scale_max, self.scale_min = 2000, 65536.0, 1.0
        # skip_step_on_overflow: with a loss scale != 1 and no clip_grad_norm pending, take one sum-of-squares pass over the gradient
        # bucket (1.3 GB, ~0.3 ms at cfg2) and feed the Adam kernel a NaN coefficient when it is not finite, so that an overflowed
        # backward skips the WHOLE update (parameters, moments and the EMA copy keep their state in every element: the kernel gates
        # the EMA blend on the coefficient too) instead of only the elements that overflowed.  The HOST counters still advance on
        # such a step (`_step` -> bias corrections, `_ema_updates` -> torch_ema's warm-up decay): the host does not learn of the
        # overflow until check_overflow(), and one tick of either moves the next update by < 1e-3 relative after a few hundred
        # steps.  Under DistributedOptimizer the guard forces the synchronise-then-update path (distributed.py::step).
        # main.train() switches it on; bench.py times the step without it (stated in the line).
        self.skip_step_on_overflow = False
        for p in a.plist:
            o, n = a.param_range(p)
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self._m[o:o + n].view(p.shape),
                             "exp_avg_sq": self._v[o:o + n].view(p.shape)}

    def zero_grad(self, set_to_none=False):   # grads are views of one bucket: a single memset (main.py:825)
        self.arena.zero_grad()

    def _eff_scale(self):
        return self.grad_scale / self.loss_scale

    def clip_grad_norm_(self, max_norm):
        """Global-norm clipping folded into the update scale (clip_grad_norm_, main.py:833-834), computed entirely on
        the device: sum of squares of the flat gradient bucket -> coefficient -> read by the Adam kernel.  Runs on the
        gradients as they are at call time (i.e. after the all-reduce in DP runs).  Returns a device scalar (the total
        norm) instead of a Python float: no host synchronisation in the step."""
        ops.join_side_stream()
        ss = torch.zeros(1, dtype=torch.float32, device=self.arena.grads.device)
        K.sumsq(self.arena.grads, ss)
        self._clip = K.clip_coef(ss, max_norm, self._eff_scale())
        return self._clip[1]

    # -- EMA of the parameters (torch_ema.ExponentialMovingAverage, main.py:520-525,598-616,843-844) ------------------
    def enable_ema(self, decay=0.995, state=None):
        """Keep an exponential moving average of the flat parameter bucket, updated inside the Adam kernel.
        state: optional {name: tensor} (a `checkpoint_ema.th` state_dict) to resume from."""
        a = self.arena
        self._ema = a.params.clone()
        self._ema_decay, self._ema_updates = float(decay), 0
        if state is not None:
            for name, p in a.module.named_parameters():
                if name in state:
                    o, n = a.param_range(p)
                    self._ema[o:o + n].copy_(state[name].reshape(-1))

    def ema_state_dict(self):
        """state_dict of the mapper with the averaged parameters (what `with ema.average_parameters(): net.state_dict()`
        yields, main.py:905-910)."""
        if self._ema is None:
            raise RuntimeError("EMA is not enabled")
        a = self.arena
        sd = {k: v.detach().clone() for k, v in a.module.state_dict().items()}
        for name, p in a.module.named_parameters():
            o, n = a.param_range(p)
            sd[name] = self._ema[o:o + n].view(p.shape).detach().clone()
        return sd

    @torch.no_grad()
    def step(self, closure=None, ranges=None):
        """One Adam update of the whole flat bucket — or, when `ranges` yields (start, end) element slices (each call of
        the iterator may first wait for that slice's gradient exchange), one launch per slice in that order, so the update
        of the slices already reduced runs while RCCL is still moving the last ones (distributed.py)."""
        a = self.arena
        g = self.param_groups[0]
        ops.join_side_stream()          # weight gradients are produced on the side stream
        clip, self._clip = self._clip, None
        if clip is None and self.skip_step_on_overflow and self.loss_scale != 1.0 and ranges is None:
            ss = torch.zeros(1, dtype=torch.float32, device=a.grads.device)
            K.sumsq(a.grads, ss)
            clip = K.clip_coef(ss, float("inf"), self._eff_scale())     # coefficient 1 for a finite norm, NaN otherwise
        ema_w = 0.0
        if not self._capturing:
            ema_w = self._host_tick()
        shadow = None if a.cdt == torch.float32 else a.shadow
        for s, e in (ranges if ranges is not None else ((0, a.total),)):
            K.adam(a.params[s:e], a.grads[s:e], self._m[s:e], self._v[s:e], None if shadow is None else shadow[s:e], g["lr"],
                   g["betas"][0], g["betas"][1], g["eps"], max(self._step, 1), self._eff_scale(),
                   ema=None if self._ema is None else self._ema[s:e], ema_weight=ema_w,
                   dev_scale=None if clip is None else clip[0:1], bad_count=self._bad,
                   dev_hyper=self._hyper[1] if (self._capturing and self._hyper is not None) else None)
        a.refresh(cast=False)

    def _host_tick(self):
        """Host bookkeeping of one update: step counter, EMA schedule, torch-layout `step` entries.  -> this step's ema weight."""
        self._step += 1
        ema_w = 0.0
        if self._ema is not None:       # torch_ema: decay = min(decay, (1 + n) / (10 + n)) with n counted from 1
            self._ema_updates += 1
            ema_w = 1.0 - min(self._ema_decay, (1 + self._ema_updates) / (10 + self._ema_updates))
        st_step = torch.tensor(float(self._step))
        for st in self.state.values():
            st["step"] = st_step
        return ema_w

    def enable_graph_hyper(self):
        """Allocate the device-resident scalars a captured update reads (see __init__)."""
        if self._hyper is None:
            self._hyper = (None, torch.zeros(5, dtype=torch.float32, device=self.arena.params.device))

    def graph_pre_step(self):
        """Before a replay of the captured step: do the update's host bookkeeping and send its scalars (lr after the scheduler,
        bias corrections of the new step count, gradient scale, EMA weight) to the device, in stream order."""
        ema_w = self._host_tick()
        g = self.param_groups[0]
        # a FRESH pinned block per step (caching host allocator): the host runs many steps ahead of the device in this mode, a
        # reused staging buffer would be overwritten before its copy has executed
        host = torch.tensor([g["lr"], 1.0 - g["betas"][0] ** self._step, math.sqrt(1.0 - g["betas"][1] ** self._step),
                             self._eff_scale(), ema_w], dtype=torch.float32).pin_memory()
        self._hyper[1].copy_(host, non_blocking=True)

    def check_overflow(self):
        """Dynamic loss scaling without a per-step host synchronisation.  Call where the host synchronises anyway (logging):
        if any step since the last call met a non-finite gradient (those elements were skipped by the kernel, the optimizer
        state stayed finite), halve `loss_scale`; after `scale_growth_interval` clean steps double it (up to scale_max).
        Returns the number of wavefront-level overflow events seen since the last call."""
        n = int(self._bad.item())
        if n:
            self._bad.zero_()
            self._good_steps = self._step
            if self.loss_scale > self.scale_min:
                self.loss_scale = max(self.scale_min, self.loss_scale * 0.5)
        elif self.loss_scale != 1.0 and self._step - self._good_steps >= self.scale_growth_interval:
            self._good_steps = self._step
            self.loss_scale = min(self.scale_max, self.loss_scale * 2.0)
        return n

    def state_dict(self):
        sd = super().state_dict()
        sd["ffvc"] = {"loss_scale": self.loss_scale, "good_steps": self._good_steps, "step": self._step}   # extra key: torch ignores it
        return sd

    def load_state_dict(self, state_dict):
        extra = state_dict.get("ffvc")
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
        if extra:                       # a resumed f16 run keeps its adapted loss scale (and does not double it at the first log)
            self.loss_scale = float(extra.get("loss_scale", self.loss_scale))
            self._good_steps = int(extra.get("good_steps", self._step))
        else:
            self._good_steps = self._step


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max, eta_min=0) closed form (main.py:705-706).
    base_lrs / last_epoch can be given to resume mid-schedule (the optimizer's lr is then already decayed)."""

    def __init__(self, optimizer, T_max, eta_min=0.0, base_lrs=None, last_epoch=0):
        self.opt, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base = list(base_lrs) if base_lrs is not None else [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = int(last_epoch)
        if self.last_epoch:
            self.last_epoch -= 1
            self.step()

    def step(self):
        self.last_epoch += 1
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = self.eta_min + (b - self.eta_min) * (1 + math.cos(math.pi * self.last_epoch / self.T_max)) / 2
