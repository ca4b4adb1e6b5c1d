"""ParamArena — flat HBM layout of a trainable mapper.

All parameters of the mapper live in ONE contiguous fp32 bucket (`params`), their gradients in a
second one (`grads`, `p.grad` are views), and the compute-dtype copies the MFMA kernels read in a
third (`shadow`, same offsets).  This is what makes the step cheap on MI355X:
  * one fused Adam launch updates params + m + v and rewrites the bf16 shadow in the same pass,
  * the data-parallel all-reduce runs over a handful of large contiguous bucket slices
    (RCCL over xGMI is per-link bound, so few large messages beat hundreds of small ones),
  * `zero_grad` is a single memset.
Transposed shadows (W^T, so that dgrad is a K-major GEMM) are refreshed by batched transposes.
"""
import torch

from . import kernels as K
from .ops import Weights

_ALIGN = 64  # elements; keeps every view 256-byte aligned for 16-byte vector access


class ParamArena:
    def __init__(self, module, cdt, allow_cpu=False):
        """allow_cpu=True builds only the flat params/grads buckets (no shadows, no kernels): used by the
        gloo world_size>1 tests of the data-parallel bucket logic, never by the product path."""
        self.cdt = cdt
        self.module = module
        ps = [p for p in module.parameters()]
        if not ps:
            raise ValueError("ParamArena: module has no parameters")
        dev = ps[0].device
        if dev.type != "cuda" and not allow_cpu:
            raise RuntimeError("ParamArena needs the module on a CUDA(HIP) device; there is no CPU path")
        self.offsets, off = [], 0
        for p in ps:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = off
        self.plist = ps
        self.params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = self.params if cdt == torch.float32 else torch.zeros(off, dtype=cdt, device=dev)
        for p, o in zip(ps, self.offsets):
            n = p.numel()
            self.params[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.params[o:o + n].view(p.shape)
            p.grad = self.grads[o:o + n].view(p.shape)
            p._ffvc_arena = self
        self._index = {id(p): i for i, p in enumerate(ps)}
        self._packs = []          # (Weights, param index) needing a transposed shadow
        self._padded = []         # (Weights, flat shadow view) whose GEMM-side shadow is a zero-padded copy
        self._grad_cbs = []
        self._tplan = None
        module._ffvc_arena = self

    # -- shadows ------------------------------------------------------------
    def shadow_of(self, p):
        i = self._index[id(p)]
        o = self.offsets[i]
        return self.shadow[o:o + p.numel()].view(p.shape)

    def make_weights(self, weight, bias, pad_n=0, pad_k=0):
        """Weights pack for a trainable [N, K, ...] parameter (Conv1d k=1 weights are viewed as [N, K]).

        pad_n / pad_k: the GEMM-side shadows get that many extra all-zero rows / columns (W.N, W.K are the PADDED sizes), for
        layers whose natural width leaves rows off the 16-byte grid the LDS-DMA kernels need (VitGAN: 6 heads x 170 = 1020
        channels, 3060-wide qkv).  Activations then carry zero pad columns; the master weight and its gradient keep their
        shape (ops._wgrad contracts only the real rows / columns)."""
        flat = self.shadow_of(weight).view(weight.shape[0], -1)
        if pad_n or pad_k:
            if bias is not None and pad_n:
                raise ValueError("make_weights: row padding of a layer with a bias is not supported")
            N, Kd = flat.shape
            sh = torch.zeros(N + pad_n, Kd + pad_k, dtype=self.cdt, device=flat.device)
            sht = torch.zeros(Kd + pad_k, N + pad_n, dtype=self.cdt, device=flat.device)
            W = Weights(weight, bias, sh, sht, on_grad=self._on_grad)
            self._padded.append((W, flat))
        else:
            sht = torch.empty(flat.shape[1], flat.shape[0], dtype=self.cdt, device=flat.device)
            W = Weights(weight, bias, flat, sht, on_grad=self._on_grad)
        self._packs.append(W)
        return W

    def refresh(self, cast=True):
        """Re-derive the shadows from the fp32 masters (after load_state_dict / an optimizer step)."""
        if cast and self.cdt != torch.float32:
            K.cast_into(self.params, self.shadow)
        for W, flat in self._padded:            # padded GEMM-side copies (zero rows / columns stay zero)
            N, Kd = flat.shape
            K.copy2d(flat, Kd, N, Kd, W.K, self.cdt, out=W.sh)
        if self._packs and self.shadow.is_cuda:
            if self._tplan is None or self._tplan.n != len(self._packs):
                self._tplan = K.TransposePlan([(W.sh, W.sht) for W in self._packs])
            self._tplan.run()                   # every W^T shadow in one launch

    # -- gradients ----------------------------------------------------------
    def zero_grad(self):
        if self.grads.is_cuda:
            from . import ops
            ops.discard_wgrad_groups()   # leftovers of a backward pass that never finished must not reach the fresh bucket
            ops.join_side_stream()
        self.grads.zero_()
        for p in self.plist:            # someone (e.g. optimizer.zero_grad(set_to_none=True)) may have dropped the views
            if p.grad is None or p.grad.data_ptr() != self.grads.data_ptr() + 4 * self.offsets[self._index[id(p)]]:
                o = self.offsets[self._index[id(p)]]
                p.grad = self.grads[o:o + p.numel()].view(p.shape)

    def add_grad_callback(self, fn):
        """fn(param) is called when a parameter's gradient has been fully written in backward."""
        self._grad_cbs.append(fn)

    def grad_written(self, *params):
        """A kernel accumulated these parameters' gradients straight into the bucket (autograd is bypassed)."""
        for fn in self._grad_cbs:
            for p in params:
                fn(p)

    def _on_grad(self, W):
        for fn in self._grad_cbs:
            fn(W.weight)
            if W.bias is not None:
                fn(W.bias)

    def param_range(self, p):
        i = self._index[id(p)]
        return self.offsets[i], p.numel()
