"""Differentiable ops of the hot path: torch.autograd.Function wrappers whose forward AND
backward only enqueue ffvc_* kernels (feed_forward_vqgan_clip_amd/kernels.py).

Conventions
  * activations are contiguous; `cdt` (compute dtype) is torch.bfloat16 / torch.float16 (throughput modes, fp32
    accumulate) or torch.float32 (parity mode: exact fp32 MFMA).  Residual streams of the mapper
    and of CLIP are fp32; the decoder is `cdt` end to end.
  * weights are `Weights` packs: fp32 master (`weight`, `bias`, reference layout/names) plus the
    `cdt` shadows the GEMMs read: `sh` = same layout [N, K], `sht` = transposed [K, N] (so dgrad
    is also a K-major GEMM).  For trainable packs, wgrad is written by the GEMM straight into
    `weight.grad` (a view of the flat gradient bucket) and `on_grad` is called — autograd never
    materialises or adds weight gradients.
  * "fork" norms return (normed, identity) so that the skip connection's gradient re-enters the
    norm's backward kernel as `dres` instead of being added by autograd.
  * producer -> consumer hints ride on the tensor OBJECT (autograd hands the same Python object from one
    Function to the next): `t._ffvc_lo` = bf16 copy of an fp32 gradient written by LayerNorm backward (picked up
    by `_as`), `t._ffvc_gn` = GroupNorm moments accumulated by the GEMM that produced `t` (picked up by
    `_GNForkFn`), `t._ffvc_f8` = (fp8 bytes of t, their Fp8Scale, only) written by the normalisation kernel that produced t for the
    fp8 GEMM / convolution named as its consumer (`only`: the 16-bit t itself was NOT written — a consumer that cannot take the bytes
    raises), `t._ffvc_gsc` = the e5m2 gradient scale of the fp8 convolution that produced t (its GroupNorm's backward then writes the
    gradient as fp8 bytes too).  A consumer that does not find the attribute (a view, a clone) simply does the work itself.
"""
import math
import os

import torch
from torch.autograd import Function

from . import kernels as K
from .kernels import ACT_GELU, ACT_NONE, ACT_QUICKGELU  # noqa: F401


# ---------------------------------------------------------------------------
# weight packs
# ---------------------------------------------------------------------------
class Weights:
    """fp32 master weight [N, K] (+ bias [N]) with compute-dtype shadows."""

    __slots__ = ("weight", "bias", "sh", "sht", "N", "K", "on_grad", "fp8", "group")

    def __init__(self, weight, bias, sh, sht, on_grad=None):
        self.weight, self.bias, self.sh, self.sht = weight, bias, sh, sht
        self.N, self.K = sh.shape[0], sh.shape[1]
        self.on_grad = on_grad
        self.fp8 = None
        self.group = None           # (WgradGroup, index): weight gradient deferred into a grouped launch (group_weights)

    @staticmethod
    def frozen(weight, bias, cdt, need_dgrad=True, fp8=False):
        """Build shadows once for a frozen layer; weight: fp32 [N, K] (any device -> cuda).
        fp8: also keep OCP e4m3 copies of W and W^T (one per-tensor scale, quantised from the fp32 master) plus the delayed
        scaling state of the layer's activation (e4m3) and gradient (e5m2) streams: Linear fwd / dgrad then run on the
        fp8 MFMA path (ffvc_gemm_fp8).  Needs 16-element aligned N and K."""
        w = weight.detach().reshape(weight.shape[0], -1).float().cuda().contiguous()
        b = None if bias is None else bias.detach().float().cuda().contiguous()
        sh = K.cast(w, cdt) if cdt != torch.float32 else w
        sht = K.transpose(w, cdt) if need_dgrad else None
        W = Weights(None, b, sh, sht)
        if fp8:
            if cdt not in K.LOWP or W.N % 16 or W.K % 16:
                raise ValueError(f"fp8 weights need a 16-bit compute dtype and N, K multiples of 16 (N={W.N}, K={W.K})")
            sw = K.Fp8Scale(K.E4M3, w.device)
            W.fp8 = {"w": sw, "sh": K.fp8_quant(w, sw, frozen=True),
                     "sht": K.fp8_quant(w.t().contiguous(), sw, frozen=True) if need_dgrad else None,
                     "x": K.Fp8Scale(K.E4M3, w.device), "g": K.Fp8Scale(K.E5M2, w.device)}
        return W


# ---------------------------------------------------------------------------
# side stream for weight / bias gradients
# ---------------------------------------------------------------------------
# wgrad GEMMs and bias-gradient reductions are NOT on the backward critical path: their results are consumed only by
# the gradient all-reduce / the optimizer.  They are enqueued on a second HIP stream (fenced by events) so that they
# execute concurrently with the dgrad chain of the following layers and fill MFMA / HBM bubbles of the main stream.
_SIDE = {"enabled": True, "stream": None, "dirty": False, "cb": False, "main": None}


def set_wgrad_side_stream(enabled):
    _SIDE["enabled"] = bool(enabled)


# HIP stream priority of the weight-gradient stream: -1 (high, default) | 0 (normal).  Measured (round 6, same box, two A/B rounds): high
# 127.0 / 127.2 ms, normal 127.5 / 127.6 — the full-chip weight-gradient launches finish sooner when the dispatcher favours them, and
# the main stream's kernels were going to share the chip with them anyway.
_SIDE_PRIORITY = os.environ.get("FFVC_SIDE_PRIORITY", "-1")


def _side_stream():
    if _SIDE["stream"] is None:
        if _SIDE_PRIORITY is not None:
            try:
                _SIDE["stream"] = torch.cuda.Stream(priority=int(_SIDE_PRIORITY))
            except (RuntimeError, ValueError):
                _SIDE["stream"] = torch.cuda.Stream()
        else:
            _SIDE["stream"] = torch.cuda.Stream()
    return _SIDE["stream"]


_DEBUG_MAIN = os.environ.get("FFVC_DEBUG_WGRAD_MAIN", "")      # debugging: "<rows>x<cols>,..." operand shapes whose work stays on the main stream


class _on_side:
    """Context: run the enclosed launches on the side stream after everything enqueued so far on the main stream."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.off = bool(_DEBUG_MAIN) and any(t is not None and f"{t.shape[-2]}x{t.shape[-1]}" in _DEBUG_MAIN.split(",") for t in tensors if t is not None and t.dim() >= 2)

    def __enter__(self):
        if not _SIDE["enabled"] or self.off:
            return self
        side = _side_stream()
        _SIDE["main"] = torch.cuda.current_stream()   # the stream this backward pass runs on (autograd restores the forward's stream per node)
        side.wait_stream(_SIDE["main"])
        for t in self.tensors:                       # keep the allocator from recycling operands still in use there
            if t is not None:
                t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        _SIDE["dirty"] = True
        if not _SIDE["cb"]:
            # join automatically when this backward pass ends, so `p.grad` is complete (in stream order on the
            # caller's stream) as soon as `loss.backward()` returns
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
                _SIDE["cb"] = True
            except RuntimeError:
                pass                                  # not inside a backward pass: callers join explicitly
        return self

    def __exit__(self, *exc):
        if _SIDE["enabled"] and not self.off:
            self.ctx.__exit__(*exc)
        return False


def _end_of_backward():
    # The engine runs this callback on whichever thread finishes the graph task; that thread's "current stream" is the device's
    # default stream, not necessarily the stream the step runs on (a step on a side stream, a stream capture): join the stream that
    # forked the gradient work.
    _SIDE["cb"] = False
    if _PENDING_GROUPS:
        # (on the thread that finishes the graph task: enqueue from the stream the backward pass ran on)
        with torch.cuda.stream(_SIDE["main"] if _SIDE["main"] is not None else torch.cuda.current_stream()):
            flush_wgrad_groups()
    join_side_stream(_SIDE["main"])


def join_side_stream(stream=None):
    """Make `stream` (default: the current stream) wait for all outstanding side-stream gradient work."""
    if _PENDING_GROUPS:
        flush_wgrad_groups()
    if _SIDE["dirty"] and _SIDE["stream"] is not None:
        (stream or torch.cuda.current_stream()).wait_stream(_SIDE["stream"])
        _SIDE["dirty"] = False


_DROP = {"n": 0}


def _drop_seeds(k):
    """k fresh 32-bit mask seeds for one forward call: torch's seed (torch.manual_seed reproduces a run) + a call counter."""
    base = (torch.initial_seed() * 0x9E3779B1) & 0xFFFFFFFF
    out = [(base + 0x632BE5AB * (_DROP["n"] + i + 1)) & 0xFFFFFFFF for i in range(k)]
    _DROP["n"] += k
    return out


def _grad_buf(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# A/B: channel-MLP weight gradients combine their K slices inside the launch (FFVC_F_SPLITK_INKERNEL) instead of fp32 slabs + one
# ffvc_slab_reduce.  Measured neutral on cfg2 (135.5 vs 135.6 ms: the reduce launches leave the side stream, the combine joins the
# GEMM: TN class 13.5 -> 14.4 ms), so the slab form stays the default.
_WGRAD_INKERNEL = os.environ.get("FFVC_WGRAD_INKERNEL", "0") != "0"
_FP8_FUSE = os.environ.get("FFVC_FP8_FUSE", "1") != "0"                   # A/B: fp8 operands straight from the producing GEMM's epilogue
_TM_WGRAD_INKERNEL = os.environ.get("FFVC_TOKMIX_WGRAD_INKERNEL", "0") != "0"   # A/B: one launch with the in-kernel split-K instead of slabs + reduce
_SK_TARGET = int(os.environ.get("FFVC_SK_TARGET", "768"))  # A/B: workgroups a small-output weight gradient is split into
_WGRAD_SK = int(os.environ.get("FFVC_WGRAD_SK", "0"))     # A/B: cap of the split-K factor of the 256x256-tile wgrads
_WGRAD_ATOMIC = os.environ.get("FFVC_WGRAD_ATOMIC", "0") != "0"   # A/B: K slices of the big weight gradients meet through fp32 atomics


def _split_k(n_out, k_out, red, bk, big_tiles=False):
    if big_tiles and n_out >= 1024 and k_out >= 1024 and n_out % 256 == 0 and k_out % 256 == 0 and red >= 8192:
        t256 = (n_out // 256) * (k_out // 256)          # 256x256 LDS-DMA tiles: one workgroup per CU (gemm2.hip, TT mode)
        if t256 < 256:
            if _WGRAD_SK:
                return max(1, min(_WGRAD_SK, 256 // t256, red // (32 * bk)))
            return max(1, min(256 // t256, red // (32 * bk)))
    tiles = ((n_out + 127) // 128) * ((k_out + 127) // 128)
    if tiles >= 512:
        return 1
    return max(1, min((_SK_TARGET + tiles - 1) // tiles, red // (8 * bk)))


# ---------------------------------------------------------------------------
# grouped weight gradients (round 5)
# ---------------------------------------------------------------------------
# One weight gradient of the Mixer's channel MLP (1024 x 4096 from a 16384-row reduction, mlp_mixer_pytorch.py:16-23) is 64 tiles
# of 256x256: a quarter of the chip.  Rounds 2-4 cut it 4 ways along K into fp32 slabs + one ffvc_slab_reduce (1.8x the algorithmic
# HBM traffic, 130 extra launches per step).  Weight gradients are consumed only by the all-reduce / optimizer, so nothing forces
# them out layer by layer: the same Linear kind of `size` consecutive layers is deferred until the last of them has its operands
# and goes out as ONE launch of size x 64 full-K tiles straight into the flat gradient bucket (K.gemm_grouped_wgrad).  The operands
# of the waiting layers stay alive until then (320 MB per layer at cfg2; 288 GB of HBM).  FFVC_WGRAD_GROUP=<size> (default 4;
# 0 / 1 = one launch per layer as before).
_WGRAD_GROUP = int(os.environ.get("FFVC_WGRAD_GROUP", "4"))
_PENDING_GROUPS = []


class WgradGroup:
    def __init__(self, members):
        self.members = list(members)
        self.pending = {}

    def add(self, idx, dy2d, x2d, rows, bias_done):
        if idx in self.pending:                  # a second use of the same weight before the group went out: flush what is there
            self.flush()
        if not self.pending:
            _PENDING_GROUPS.append(self)
            _SIDE["main"] = torch.cuda.current_stream()
            if not _SIDE["cb"]:                  # a backward pass that ends with a partly filled group flushes it (see _end_of_backward)
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
                    _SIDE["cb"] = True
                except RuntimeError:
                    pass                         # not inside a backward pass: join_side_stream() flushes
        self.pending[idx] = (dy2d, x2d, rows, bias_done)
        # autograd still runs the AccumulateGrad hooks of this layer's parameters when its backward returns: a gradient-ready
        # listener (distributed.DistributedOptimizer) must not take that for "written" — the report comes from flush()
        W = self.members[idx]
        W.weight._ffvc_deferred = True
        if W.bias is not None:
            W.bias._ffvc_deferred = True
        if len(self.pending) == len(self.members):
            self.flush()

    def flush(self):
        pend, self.pending = self.pending, {}
        if not pend:
            return
        if self in _PENDING_GROUPS:
            _PENDING_GROUPS.remove(self)
        idxs = sorted(pend)
        Ws = [self.members[i] for i in idxs]
        for W in Ws:
            W.weight._ffvc_deferred = False
            if W.bias is not None:
                W.bias._ffvc_deferred = False
        wgs = [_grad_buf(W.weight) for W in Ws]
        rows = pend[idxs[0]][2]
        stride = (wgs[1].data_ptr() - wgs[0].data_ptr()) // 4 if len(wgs) > 1 else 0
        regular = (len(idxs) >= 2 and stride > 0 and all(pend[i][2] == rows for i in idxs) and
                   all(wgs[j].data_ptr() - wgs[0].data_ptr() == 4 * stride * j for j in range(len(wgs))))
        if not regular:                          # a lone member / irregular layout: the per-layer launches
            for i in idxs:
                dy2d, x2d, r, bd = pend[i]
                _wgrad_now(dy2d, x2d, self.members[i], r, None, bd)
            return
        W0 = Ws[0]
        Nr = W0.weight.shape[0]
        Kr = W0.weight.numel() // Nr
        dys, xs = [pend[i][0] for i in idxs], [pend[i][1] for i in idxs]
        with _on_side(*dys, *xs):
            K.gemm_grouped_wgrad(dys, xs, wgs[0], stride, Nr, Kr, rows, W0.N, W0.K)
            for i, W in zip(idxs, Ws):
                if W.bias is not None and W.bias.requires_grad and not pend[i][3]:
                    K.colsum(pend[i][0], _grad_buf(W.bias), accumulate=True)
        for W in reversed(Ws):                   # gradient-ready reports in the order backward produces them
            if W.on_grad is not None:
                W.on_grad(W)


def wgrad_group_size():
    return min(_WGRAD_GROUP, 8) if _WGRAD_GROUP > 1 else 0


def group_weights(packs, size=None):
    """Mark consecutive runs of `size` packs (the same Linear of consecutive layers, registration order) for grouped weight
    gradients.  Only packs whose gradient takes the 256x256 LDS-DMA kernel qualify (16-bit shadows, N, K multiples of 256,
    un-padded); others are left alone."""
    size = _WGRAD_GROUP if size is None else int(size)
    if size < 2:
        return
    size = min(size, 8)
    ok = [W for W in packs if W.sh.dtype in K.LOWP and W.N % 256 == 0 and W.K % 256 == 0 and W.N >= 1024 and W.K >= 1024 and
          W.weight.shape[0] == W.N and W.weight.numel() == W.N * W.K]
    if len(ok) != len(packs):
        return
    for s0 in range(0, len(packs), size):
        run = packs[s0:s0 + size]
        if len(run) >= 2:
            g = WgradGroup(run)
            for i, W in enumerate(run):
                W.group = (g, i)


def flush_wgrad_groups():
    """Send out every partly filled group (end of a backward pass that did not reach all members of a group)."""
    for g in list(_PENDING_GROUPS):
        g.flush()


def discard_wgrad_groups():
    """Forget every queued group WITHOUT launching it.  Whatever is still queued when the gradients are about to be zeroed belongs
    to a backward pass that never finished (an exception / OOM between add() and flush()): flushing it into the fresh bucket would
    accumulate a stale gradient into the next step, the `_ffvc_deferred` marks would mute a gradient-ready listener, and the
    operands (320 MB per waiting layer at cfg2) would stay pinned.  Returns the number of dropped entries."""
    n = 0
    for g in list(_PENDING_GROUPS):
        pend, g.pending = g.pending, {}
        n += len(pend)
        for i in pend:
            W = g.members[i]
            W.weight._ffvc_deferred = False
            if W.bias is not None:
                W.bias._ffvc_deferred = False
    del _PENDING_GROUPS[:]
    _SIDE["cb"] = False
    return n


def _wgrad(dy2d, x2d, W, rows, ldy=None, bias_done=False):
    """weight.grad[N,K] += dy[rows,N]^T @ x[rows,K]; bias.grad += colsum(dy).  ldy: row stride of dy (default N).
    bias_done: the kernel that produced dy already accumulated its column sums into bias.grad."""
    if (W.group is not None and _WGRAD_GROUP > 1 and ldy in (None, W.N) and rows >= 512 and rows % 64 == 0 and
            dy2d.dtype in K.LOWP and dy2d.is_contiguous() and x2d.is_contiguous()):
        g, idx = W.group
        g.add(idx, dy2d, x2d, rows, bias_done)
        return
    _wgrad_now(dy2d, x2d, W, rows, ldy, bias_done)


def _wgrad_now(dy2d, x2d, W, rows, ldy=None, bias_done=False):
    wg = _grad_buf(W.weight)
    bg = _grad_buf(W.bias) if (W.bias is not None and W.bias.requires_grad and not bias_done) else None
    bk = 64 if dy2d.dtype in K.LOWP else 32
    # the master's own shape: W.N / W.K are the (possibly zero-padded) row strides of dy / x (ParamArena.make_weights)
    Nr = W.weight.shape[0]
    Kr = W.weight.numel() // Nr
    big = dy2d.dtype in K.LOWP and (ldy or W.N) == Nr and W.K == Kr
    sk = _split_k(Nr, Kr, rows, bk, big_tiles=big)
    # the 256x256-tile weight-gradient kernel (same conditions as csrc/gemm2.hip's dispatch) combines its K slices inside the
    # launch: no fp32 slabs, no ffvc_slab_reduce pass
    inker = (_WGRAD_INKERNEL and big and 1 < sk <= 8 and Nr >= 1024 and Kr >= 1024 and Nr % 256 == 0 and Kr % 256 == 0 and rows >= 8192 and
             (Nr // 256) * (Kr // 256) * sk >= 192)
    with _on_side(dy2d, x2d):
        K.gemm_splitk_accumulate(dy2d, x2d, wg, Nr, Kr, rows, sk, in_kernel=inker, atomic=_WGRAD_ATOMIC and big and sk > 1,
                                 ldx=ldy or W.N, ldw=W.K, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS)
        if bg is not None:
            K.colsum(dy2d, bg, accumulate=True, ld=ldy)
    if W.on_grad is not None:
        W.on_grad(W)


_LN_LO = os.environ.get("FFVC_LN_LO", "1") != "0"      # A/B switch for the fused bf16 gradient copy
_F8_PRODUCER = os.environ.get("FFVC_FP8_PRODUCER", "1") != "0"   # A/B switch: LayerNorm / GroupNorm write the fp8 operand of their consumer


def _f8_twin(t, sc):
    """The fp8 bytes of `t` in scale `sc` if its producer wrote them, else None (and a tensor that exists ONLY as such bytes must
    not reach a consumer that would read its 16-bit values)."""
    tw = getattr(t, "_ffvc_f8", None)
    if tw is not None and sc is not None and tw[1] is sc:
        return tw[0]
    if tw is not None and tw[2]:
        raise RuntimeError("a tensor whose producer wrote only its fp8 bytes reached a consumer that cannot use them "
                           "(FFVC_FP8_PRODUCER=0 disables the producer-side quantisation)")
    return None
_ACTGRAD = os.environ.get("FFVC_ACTGRAD", "1") != "0"  # A/B switch: MLPs keep act'(pre) instead of pre (16-bit modes)


def _gn_request(gn, y, images, hw, C):
    """Moments buffer for a producer whose NHWC output `y` goes into GroupNorm(32) next (None if not fusable)."""
    if not gn or y.dtype not in K.LOWP or not K.gn_sums_ok(images * hw, C, hw, y.dtype):
        return None
    sums = K.gn_sums_buffer(images, 32, y.device)
    y._ffvc_gn = sums
    return sums


def carry_gn(src, dst):
    """Views / reshapes of a producer's output keep its GroupNorm moments."""
    g = getattr(src, "_ffvc_gn", None)
    if g is not None:
        dst._ffvc_gn = g
    return dst


def _as(t, dtype):
    if t.dtype == dtype:
        return t
    lo = getattr(t, "_ffvc_lo", None)          # a producer kernel already wrote the low-precision copy (LayerNorm bwd)
    if lo is not None and lo.dtype == dtype and lo.shape == t.shape:
        return lo
    return K.cast(t, dtype)


def _contig(t):
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------
# Linear  (y = x W^T + b [+ residual])
# ---------------------------------------------------------------------------
class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, W, out_dtype, gn_hw=0, drop=0.0):
        cdt = W.sh.dtype
        x = _contig(x)
        rows = x.numel() // W.K
        y = torch.empty(*x.shape[:-1], W.N, dtype=out_dtype or cdt, device=x.device)
        ctx.drop = (float(drop), _drop_seeds(1)[0]) if drop else None
        if ctx.drop:                 # y = residual + dropout(x W^T + b)   (vitgan.py:114,133)
            K.gemm(x, W.sh, y, rows, W.N, W.K, ldx=W.K, ldw=W.K, bias=W.bias)
            if residual is not None and (residual.dtype != torch.float32 or y.dtype != torch.float32):
                raise TypeError("linear(drop>0) with a residual needs the fp32 residual stream")
            K.dropout(y, ctx.drop[0], ctx.drop[1], residual=residual, out=y)
        elif W.fp8 is not None and not gn_hw and x.dtype in K.LOWP:
            f = W.fp8                  # fp8 MFMA path of a frozen layer: quantise the activation, per-tensor delayed scale
            x8 = _f8_twin(x, f["x"])
            if x8 is None:
                x8 = K.fp8_quant(x, f["x"])
            K.gemm_fp8(x8, f["sh"], y, rows, W.N, W.K, f["x"], f["w"], lo_dtype=cdt, bias=W.bias, residual=residual)
            K.fp8_next_scale(f["x"])
            if f["sht"] is not None:
                y._ffvc_gsc = f["g"]       # this layer's dgrad runs on the fp8 path: the LayerNorm behind y may write its e5m2 operand
        else:
            _f8_twin(x, None)
            sums = _gn_request(gn_hw > 0, y, rows // gn_hw if gn_hw else 0, gn_hw, W.N)
            K.gemm(x, W.sh, y, rows, W.N, W.K, ldx=W.K, ldw=W.K, bias=W.bias, residual=residual,
                   gn_sums=None if sums is None else (sums, gn_hw, W.N // 32))
        ctx.W, ctx.rows = W, rows
        ctx.train = weight is not None and weight.requires_grad
        ctx.has_res = residual is not None
        ctx.xshape = x.shape
        ctx.save_for_backward(x if ctx.train else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        W, rows = ctx.W, ctx.rows
        cdt = W.sh.dtype
        dy = _contig(dy)
        f = W.fp8
        dy8 = None
        if f is not None and f["sht"] is not None and not ctx.drop and not ctx.train and cdt in K.LOWP:
            dy8 = _f8_twin(dy, f["g"])     # written by the LayerNorm backward that produced dy (frozen fp8 layer: nothing else reads a 16-bit dy)
        dyt = None
        if dy8 is None:
            dyt = K.dropout(dy, ctx.drop[0], ctx.drop[1], out_dtype=cdt) if ctx.drop else _as(dy, cdt)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(ctx.xshape, dtype=cdt, device=dy.device)
            if dy8 is not None or (f is not None and f["sht"] is not None and not ctx.drop and dyt.dtype in K.LOWP):
                if dy8 is None:
                    dy8 = K.fp8_quant(dyt, f["g"])                   # gradients travel as e5m2
                K.gemm_fp8(dy8, f["sht"], dx, rows, W.K, W.N, f["g"], f["w"], lo_dtype=cdt)
                K.fp8_next_scale(f["g"])
            else:
                K.gemm(dyt, W.sht, dx, rows, W.K, W.N, ldx=W.N, ldw=W.N)
        if ctx.train:
            (x,) = ctx.saved_tensors
            _wgrad(dyt, x, W, rows)
        return dx, None, None, (dy if ctx.has_res else None), None, None, None, None


def linear(x, W, residual=None, out_dtype=None, gn_hw=0, drop=0.0):
    """gn_hw > 0: the rows are NHWC pixels (gn_hw per image) feeding a GroupNorm(32) next (see conv3x3).
    drop > 0: nn.Dropout(drop) on the linear's output, before the residual is added."""
    return _LinearFn.apply(x, W.weight, W.bias, residual, W, out_dtype, gn_hw, drop)


# ---------------------------------------------------------------------------
# MLP  (y = act(x W1^T + b1) W2^T + b2 [+ residual]) — channel mixing, ViT / VitGAN / x-transformer FF
# ---------------------------------------------------------------------------
class _MLPFn(Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, W1, W2, act, out_dtype, drop=0.0):
        cdt = W1.sh.dtype
        x = _contig(x)
        rows = x.numel() // W1.K
        h_pre = torch.empty(*x.shape[:-1], W1.N, dtype=cdt, device=x.device)
        h = torch.empty_like(h_pre)
        ctx.fp8 = (W1.fp8 is not None and W2.fp8 is not None and not drop and x.dtype in K.LOWP)
        # 16-bit modes keep act'(pre) instead of pre (`h_pre` then IS the derivative): the forward forms it from the erf / exp it
        # evaluates anyway and the backward epilogue becomes a plain multiply (FFVC_F_AUX_ACTGRAD; the fp32 parity mode keeps
        # the textbook form)
        ctx.ag = K.F_AUX_ACTGRAD if (cdt in K.LOWP and _ACTGRAD and act in (ACT_GELU, ACT_QUICKGELU)) else 0
        # frozen fp8 MLP: once the hidden activation's scale exists (second step on) the first GEMM's epilogue writes it as e4m3
        # itself — no 16-bit h, no quantisation pass (csrc/gemm_common.h EPI_O_F8; FFVC_FP8_FUSE=0: separate passes)
        h8 = None
        fuse = (ctx.fp8 and _FP8_FUSE and ctx.ag and (w1 is None or not w1.requires_grad) and W1.bias is not None and
                W2.fp8["x"].ready and W1.N % 8 == 0)
        if ctx.fp8:
            f1 = W1.fp8
            x8 = _f8_twin(x, f1["x"])
            if x8 is None:
                x8 = K.fp8_quant(x, f1["x"])
            if fuse:
                h8 = torch.empty(h_pre.shape, dtype=torch.uint8, device=x.device)
                K.gemm_fp8(x8, f1["sh"], h8, rows, W1.N, W1.K, f1["x"], f1["w"], lo_dtype=cdt, bias=W1.bias, act=act, aux=h_pre,
                           ldaux=W1.N, flags=K.F_WRITE_PREACT | ctx.ag, out_scale=W2.fp8["x"])
            else:
                K.gemm_fp8(x8, f1["sh"], h, rows, W1.N, W1.K, f1["x"], f1["w"], lo_dtype=cdt, bias=W1.bias, act=act, aux=h_pre,
                           ldaux=W1.N, flags=K.F_WRITE_PREACT | ctx.ag)
            K.fp8_next_scale(f1["x"])
        else:
            _f8_twin(x, None)
            K.gemm(x, W1.sh, h, rows, W1.N, W1.K, ldx=W1.K, ldw=W1.K, bias=W1.bias, act=act, aux=h_pre, ldaux=W1.N,
                   flags=K.F_WRITE_PREACT | ctx.ag)
        y = torch.empty(*x.shape[:-1], W2.N, dtype=out_dtype or cdt, device=x.device)
        ctx.drop = (float(drop),) + tuple(_drop_seeds(2)) if drop else None
        if ctx.fp8:
            f2 = W2.fp8
            if h8 is None:
                h8 = K.fp8_quant(h, f2["x"])
            K.gemm_fp8(h8, f2["sh"], y, rows, W2.N, W2.K, f2["x"], f2["w"], lo_dtype=cdt, bias=W2.bias, residual=residual)
            K.fp8_next_scale(f2["x"])
            if f2["sht"] is not None and W1.fp8["sht"] is not None:
                y._ffvc_gsc = f2["g"]      # the LayerNorm behind y may write the e5m2 operand of this block's first dgrad
        elif ctx.drop:    # Linear, act, Dropout, Linear, Dropout (mlp_mixer_pytorch.py:16-23, vitgan.py:36-41), then + residual
            if residual is not None and (residual.dtype != torch.float32 or y.dtype != torch.float32):
                raise TypeError("mlp(drop>0) with a residual needs the fp32 residual stream")
            K.dropout(h, ctx.drop[0], ctx.drop[1], out=h)
            K.gemm(h, W2.sh, y, rows, W2.N, W2.K, ldx=W2.K, ldw=W2.K, bias=W2.bias)
            K.dropout(y, ctx.drop[0], ctx.drop[2], residual=residual, out=y)
        else:
            K.gemm(h, W2.sh, y, rows, W2.N, W2.K, ldx=W2.K, ldw=W2.K, bias=W2.bias, residual=residual)
        ctx.W1, ctx.W2, ctx.rows, ctx.act = W1, W2, rows, act
        ctx.train = w1 is not None and w1.requires_grad
        ctx.has_res = residual is not None
        ctx.save_for_backward(x if ctx.train else None, h_pre, h if ctx.train else None)
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        W1, W2, rows = ctx.W1, ctx.W2, ctx.rows
        cdt = W1.sh.dtype
        x, h_pre, h = ctx.saved_tensors
        dy = _contig(dy)
        f8_bwd = ctx.fp8 and W2.fp8["sht"] is not None and W1.fp8["sht"] is not None
        dy8 = _f8_twin(dy, W2.fp8["g"]) if (f8_bwd and not ctx.train) else None     # written by the LayerNorm backward that produced dy
        dyt = None
        if dy8 is None:
            dyt = K.dropout(dy, ctx.drop[0], ctx.drop[2], out_dtype=cdt) if ctx.drop else _as(dy, cdt)
        dh = torch.empty_like(h_pre)
        # the first Linear's bias gradient = column sums of dh: accumulated by the epilogue that writes dh
        b1_fused = (ctx.train and not ctx.drop and W1.bias is not None and W1.bias.requires_grad and
                    K.colsum_fusable(cdt, W2.K, W2.N))
        if f8_bwd:
            f1, f2 = W1.fp8, W2.fp8
            if dy8 is None:
                dy8 = K.fp8_quant(dyt, f2["g"])
            # the hidden gradient leaves the aux-multiply epilogue as e5m2 once its scale exists (nothing else reads it: frozen layer)
            fuse = _FP8_FUSE and ctx.ag and not ctx.train and ctx.needs_input_grad[0] and f1["g"].ready and W2.K % 8 == 0
            dh8 = None
            if fuse:
                dh8 = torch.empty(h_pre.shape, dtype=torch.uint8, device=dy.device)
                K.gemm_fp8(dy8, f2["sht"], dh8, rows, W2.K, W2.N, f2["g"], f2["w"], lo_dtype=cdt, aux=h_pre, ldaux=W2.K,
                           act=ctx.act, flags=K.F_MUL_ACT_GRAD | ctx.ag, out_scale=f1["g"])
            else:
                K.gemm_fp8(dy8, f2["sht"], dh, rows, W2.K, W2.N, f2["g"], f2["w"], lo_dtype=cdt, aux=h_pre, ldaux=W2.K,
                           act=ctx.act, flags=K.F_MUL_ACT_GRAD | ctx.ag)
            K.fp8_next_scale(f2["g"])
            dx = None
            if ctx.needs_input_grad[0]:
                if dh8 is None:
                    dh8 = K.fp8_quant(dh, f1["g"])
                dx = torch.empty(ctx.xshape, dtype=cdt, device=dy.device)
                K.gemm_fp8(dh8, f1["sht"], dx, rows, W1.K, W1.N, f1["g"], f1["w"], lo_dtype=cdt)
                K.fp8_next_scale(f1["g"])
            return dx, None, None, None, None, (dy if ctx.has_res else None), None, None, None, None, None
        K.gemm(dyt, W2.sht, dh, rows, W2.K, W2.N, ldx=W2.N, ldw=W2.N, aux=h_pre, ldaux=W2.K, act=ctx.act,
               flags=K.F_MUL_ACT_GRAD | ctx.ag, colsum=_grad_buf(W1.bias) if b1_fused else None)
        if ctx.drop:        # the hidden mask commutes with the element-wise act' factor the epilogue just applied
            K.dropout(dh, ctx.drop[0], ctx.drop[1], out=dh)
        if ctx.train:
            _wgrad(dyt, h, W2, rows)
            _wgrad(dh, x, W1, rows, bias_done=b1_fused)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(ctx.xshape, dtype=cdt, device=dy.device)
            K.gemm(dh, W1.sht, dx, rows, W1.K, W1.N, ldx=W1.N, ldw=W1.N)
        return dx, None, None, None, None, (dy if ctx.has_res else None), None, None, None, None, None


def mlp(x, W1, W2, act, residual=None, out_dtype=None, drop=0.0):
    """drop > 0: nn.Dropout(drop) after the activation and after the second Linear (before the residual add)."""
    return _MLPFn.apply(x, W1.weight, W1.bias, W2.weight, W2.bias, residual, W1, W2, act, out_dtype, drop)


# ---------------------------------------------------------------------------
# Token-mixing MLP (Conv1d k=1 over the token axis, mlp_mixer_pytorch.py:28,34):
#   h[b] = gelu(W1 @ xn[b] + b1[:,None]);  y[b] = W2 @ h[b] + b2[:,None] + residual[b]
# xn: [B, T, D]; W1: [O, T]; W2: [T, O].  Batched GEMMs with the activation as the TRANS operand.
# ---------------------------------------------------------------------------
class _TokenMLPFn(Function):
    @staticmethod
    def forward(ctx, xn, w1, b1, w2, b2, residual, W1, W2, out_dtype, drop=0.0):
        cdt = W1.sh.dtype
        xn = _contig(xn)
        B, T, D = xn.shape
        O = W1.N
        ctx.W1, ctx.W2 = W1, W2
        ctx.train = w1 is not None and w1.requires_grad
        ctx.has_res = residual is not None
        ctx.dims = (B, T, D, O)
        ctx.drop = (float(drop),) + tuple(_drop_seeds(2)) if drop else None
        ctx.fused = (not ctx.drop and residual is not None and residual.dtype == torch.float32 and
                     (out_dtype or cdt) == torch.float32 and W1.bias is not None and W2.bias is not None and
                     K.tokmix_supported(cdt, T, D, O))
        ctx.saved_hidden = False
        if ctx.fused and _TM_SAVE and any(ctx.needs_input_grad):
            # one launch that also writes h and act'(pre): the backward is a plain aux-multiply GEMM, no recomputation
            y, h, gact = K.tokmix_fwd_save(xn, W1.sh, W1.bias, W2.sh, W2.bias, _contig(residual))
            ctx.saved_hidden = True
            ctx.save_for_backward(xn if ctx.train else None, gact, h if ctx.train else None)
            return y
        if ctx.fused:
            # one launch, the hidden activation stays on chip; backward recomputes it (nothing saved but xn)
            ctx.save_for_backward(xn, None, None)
            return K.tokmix_fwd(xn, W1.sh, W1.bias, W2.sh, W2.bias, _contig(residual))
        h_pre = torch.empty(B, O, D, dtype=cdt, device=xn.device)
        h = torch.empty_like(h_pre)
        K.gemm(W1.sh, xn, h, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, bias=W1.bias, act=ACT_GELU, aux=h_pre, ldaux=D,
               flags=K.F_WRITE_PREACT | K.F_BIAS_ALONG_M, batch=B, wb=(T * D, 0), yb=(O * D, 0), ab=(O * D, 0))
        y = torch.empty(B, T, D, dtype=out_dtype or cdt, device=xn.device)
        if ctx.drop:
            if residual is not None and (residual.dtype != torch.float32 or y.dtype != torch.float32):
                raise TypeError("token_mlp(drop>0) with a residual needs the fp32 residual stream")
            K.dropout(h, ctx.drop[0], ctx.drop[1], out=h)
            K.gemm(W2.sh, h, y, T, D, O, ldx=O, ldw=D, w_mode=K.OP_TRANS, bias=W2.bias, flags=K.F_BIAS_ALONG_M, batch=B,
                   wb=(O * D, 0), yb=(T * D, 0))
            K.dropout(y, ctx.drop[0], ctx.drop[2], residual=None if residual is None else _contig(residual), out=y)
        else:
            K.gemm(W2.sh, h, y, T, D, O, ldx=O, ldw=D, w_mode=K.OP_TRANS, bias=W2.bias, residual=residual,
                   flags=K.F_BIAS_ALONG_M, batch=B, wb=(O * D, 0), yb=(T * D, 0), rb=(T * D, 0))
        ctx.save_for_backward(xn if ctx.train else None, h_pre, h if ctx.train else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        W1, W2 = ctx.W1, ctx.W2
        cdt = W1.sh.dtype
        B, T, D, O = ctx.dims
        xn, h_pre, h = ctx.saved_tensors
        dy = _contig(dy)
        dyt = K.dropout(dy, ctx.drop[0], ctx.drop[2], out_dtype=cdt) if ctx.drop else _as(dy, cdt)
        # db1 inside the fused kernel is available (FFVC_TOKMIX_DB1=1) but measured +1.1..2 ms/step (atomics in its chunk
        # loop, profiles/r02_bias_grad_fusion_ab.txt): the rowsum launch on the side stream stays the default
        b1_fused = (ctx.fused and not ctx.saved_hidden and ctx.train and W1.bias.requires_grad and
                    os.environ.get("FFVC_TOKMIX_DB1", "0") != "0")
        if ctx.saved_hidden:
            # dh[b] = (W2^T @ dy[b]) * act'(pre[b]), act' saved by the forward
            dh = torch.empty_like(h_pre)
            K.gemm(W2.sht, dyt, dh, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, aux=h_pre, ldaux=D, act=ACT_GELU,
                   flags=K.F_MUL_ACT_GRAD | K.F_AUX_ACTGRAD, batch=B, wb=(T * D, 0), yb=(O * D, 0), ab=(O * D, 0))
        elif ctx.fused:
            h, dh = K.tokmix_bwd_hidden(xn, dyt, W1.sh, W1.bias, W2.sht, db1=_grad_buf(W1.bias) if b1_fused else None)
        else:
            # dh_pre[b] = (W2^T @ dy[b]) * gelu'(h_pre[b])
            dh = torch.empty_like(h_pre)
            K.gemm(W2.sht, dyt, dh, O, D, T, ldx=T, ldw=D, w_mode=K.OP_TRANS, aux=h_pre, ldaux=D, act=ACT_GELU,
                   flags=K.F_MUL_ACT_GRAD, batch=B, wb=(T * D, 0), yb=(O * D, 0), ab=(O * D, 0))
            if ctx.drop:
                K.dropout(dh, ctx.drop[0], ctx.drop[1], out=dh)
        if ctx.train:
            bk = 64 if cdt in K.LOWP else 32
            seg_ok = D % bk == 0
            for (g, a, W, n_out, k_out) in ((dyt, h, W2, T, O), (dh, xn, W1, O, T)):
                wg = _grad_buf(W.weight)
                bg = _grad_buf(W.bias) if (W.bias is not None and W.bias.requires_grad) else None
                with _on_side(g, a):
                    if seg_ok:
                        # dW[n,k] = sum_{b,d} g[b][n,d] a[b][k,d]: K-major GEMM over the segmented (b,d) axis
                        sk = 1 if _TM_WGRAD_INKERNEL else _split_k(n_out, k_out, B * D, bk)
                        K.gemm_splitk_accumulate(g, a, wg, n_out, k_out, B * D, sk, ldx=D, ldw=D, kseg=D,
                                                 xkso=n_out * D, wkso=k_out * D)
                    else:
                        for b in range(B):
                            K.gemm(g[b], a[b], wg, n_out, k_out, D, ldx=D, ldw=D, flags=K.F_ACCUM_OUT)
                    if bg is not None and not (b1_fused and W is W1):
                        K.rowsum(g, bg, n_out, accumulate=True)
                if W.on_grad is not None:
                    W.on_grad(W)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, T, D, dtype=cdt, device=dy.device)
            K.gemm(W1.sht, dh, dx, T, D, O, ldx=O, ldw=D, w_mode=K.OP_TRANS, batch=B, wb=(O * D, 0), yb=(T * D, 0))
        return dx, None, None, None, None, (dy if ctx.has_res else None), None, None, None, None


def token_mlp(xn, W1, W2, residual=None, out_dtype=None, drop=0.0):
    return _TokenMLPFn.apply(xn, W1.weight, W1.bias, W2.weight, W2.bias, residual, W1, W2, out_dtype, drop)


# ---------------------------------------------------------------------------
# LayerNorm (fork form)
# ---------------------------------------------------------------------------
class _LNForkFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, out_dtype, eps, f8_for=None):
        x = _contig(x)
        g, b = gamma.detach(), beta.detach()
        f8 = None
        if (f8_for is not None and _F8_PRODUCER and getattr(f8_for, "fp8", None) is not None and out_dtype in K.LOWP and
                x.shape[-1] % 4 == 0 and f8_for.fp8["x"].ready):
            f8 = f8_for.fp8["x"]       # the one consumer is a frozen fp8 linear: its operand leaves this kernel as e4m3 bytes, y is not written
        if f8 is not None:
            y, mean, rstd, y8 = K.layernorm_fwd(x, g, b, out_dtype, eps, f8=f8, f8_only=True)
            y._ffvc_f8 = (y8, f8, True)
        else:
            y, mean, rstd = K.layernorm_fwd(x, g, b, out_dtype, eps)
        ctx.save_for_backward(x, g, mean, rstd)
        # x (the residual stream) came out of an fp8 linear / mlp whose dgrad wants this node's gradient as e5m2 bytes
        ctx.gsc = getattr(x, "_ffvc_gsc", None) if _F8_PRODUCER else None
        ctx.train = gamma.requires_grad
        # parameters that live in a ParamArena get their gradients accumulated straight into the flat bucket
        ctx.params = (gamma, beta) if (ctx.train and getattr(gamma, "_ffvc_arena", None) is not None and
                                       getattr(beta, "_ffvc_arena", None) is gamma._ffvc_arena) else None
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, g, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None
        dy = _contig(dy)
        if dres is not None:
            dres = _as(_contig(dres), x.dtype)
        # fp32 residual stream under bf16 compute: the gradient leaving here feeds a bf16 GEMM next -> emit its copy now
        lo = _LN_LO and x.dtype == torch.float32 and dy.dtype in K.LOWP
        if ctx.params is not None:
            gamma, beta = ctx.params
            dx = K.layernorm_bwd_acc(dy, x, g, mean, rstd, _grad_buf(gamma), _grad_buf(beta), dres=dres, want_lo=lo)
            gamma._ffvc_arena.grad_written(gamma, beta)
            return dx, None, None, None, None, None
        gsc = ctx.gsc
        if gsc is not None and gsc.ready and lo and not ctx.train and x.shape[-1] % 4 == 0:
            # the 16-bit copy's only reader would be that dgrad's quantiser: write its fp8 bytes instead
            dx, dx8 = K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, f8=gsc)
            dx._ffvc_f8 = (dx8, gsc, False)
            return dx, None, None, None, None, None
        dx, dg, db = K.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, want_param_grads=ctx.train, want_lo=lo)
        return dx, dg, db, None, None, None


def layernorm_fork(x, gamma, beta, out_dtype, eps=1e-5, f8_for=None):
    """-> (LN(x) in out_dtype, identity alias of x whose gradient is fused into LN's backward).
    f8_for: the frozen fp8 `Weights` pack of the ONE layer that consumes LN(x) (linear / mlp): once its activation scale exists the
    kernel writes that layer's e4m3 operand itself and skips the 16-bit output."""
    return _LNForkFn.apply(x, gamma, beta, out_dtype, eps, f8_for)


def layernorm(x, gamma, beta, out_dtype, eps=1e-5):
    return _LNForkFn.apply(x, gamma, beta, out_dtype, eps)[0]


# ---------------------------------------------------------------------------
# GroupNorm(32, eps 1e-6) [+ swish] on NHWC (fork form), frozen affine
# ---------------------------------------------------------------------------
class _GNForkFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, swish, f8_for=None, grad_sole=False):
        x = _contig(x)
        sums = getattr(x, "_ffvc_gn", None)          # moments accumulated by the GEMM that produced x
        if sums is not None and tuple(sums.shape) != (x.shape[0], 32, 2):
            sums = None
        f8 = None
        if (f8_for is not None and _F8_PRODUCER and f8_for.fp8 is not None and x.dtype in K.LOWP and x.dim() == 4 and
                f8_for.fp8["x"].ready and K.conv_fp8_ok(x.shape[0], x.shape[1], x.shape[2], x.shape[3], f8_for.Cout)):
            f8 = f8_for.fp8["x"]       # the one consumer is an fp8 convolution: its e4m3 operand leaves this kernel, y is not written
        if f8 is not None:
            y, mean, rstd, y8 = K.groupnorm_fwd(x, gamma, beta, 32, 1e-6, swish, sums=sums, f8=f8, f8_only=True)
            y._ffvc_f8 = (y8, f8, True)
        else:
            y, mean, rstd = K.groupnorm_fwd(x, gamma, beta, 32, 1e-6, swish, sums=sums)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.swish = swish
        if x.dtype in K.LOWP and x.dim() == 4 and f8 is None:
            # the 3x3 convolution that consumes y can fold this node's BACKWARD statistics into its dgrad epilogue (round 6):
            # what it needs rides on y (see _Conv3x3Fn)
            y._ffvc_gnb = (x, gamma, beta, mean, rstd, bool(swish))
        # x came out of an fp8 convolution whose dgrad wants this node's gradient as e5m2 bytes
        ctx.gsc = getattr(x, "_ffvc_gsc", None) if _F8_PRODUCER else None
        ctx.grad_sole = bool(grad_sole)
        ctx.set_materialize_grads(False)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None
        _f8_twin(dy, None)
        dy = _as(_contig(dy), x.dtype)
        if dres is not None:
            _f8_twin(dres, None)
            dres = _as(_contig(dres), x.dtype)
        gsc = ctx.gsc
        gnb_sums = getattr(dy, "_ffvc_gnb_sums", None)  # accumulated by the dgrad convolution that produced dy for THIS node
        if gnb_sums is not None and gnb_sums[1].data_ptr() == mean.data_ptr() and not (gsc is not None and gsc.ready):
            dx = K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=dres, G=32, swish=ctx.swish, sums=gnb_sums[0])
            return dx, None, None, None, None, None
        if gsc is not None and gsc.ready and x.dtype in K.LOWP:
            only = ctx.grad_sole and dres is None        # nothing but that dgrad reads the gradient: skip its 16-bit form
            dx, dx8 = K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=dres, G=32, swish=ctx.swish, f8=gsc, f8_only=only)
            dx._ffvc_f8 = (dx8, gsc, only)
        else:
            dx = K.groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=dres, G=32, swish=ctx.swish)
        return dx, None, None, None, None, None


def groupnorm_fork(x, gamma, beta, swish=True, f8_for=None, grad_sole=False):
    """f8_for: the `ConvWeights` of the ONE 3x3 convolution that consumes the normalised tensor — when that launch will run on the fp8
    path the kernel writes its e4m3 operand and skips the 16-bit output.  grad_sole: x has no consumer but this node and the
    identity output is not used, so (x being the output of an fp8 convolution) the gradient may leave as e5m2 bytes only."""
    return _GNForkFn.apply(x, gamma, beta, swish, f8_for, grad_sole)


# ---------------------------------------------------------------------------
# 3x3 conv (implicit GEMM, NHWC, frozen weights), optional fused nearest-2x upsample of the input
# ---------------------------------------------------------------------------
class ConvWeights:
    """Frozen 3x3 conv: w [Cout, 3,3,Cin] and the dgrad filter wd [Cin, 3,3,Cout] (flipped taps), both K-major."""

    __slots__ = ("w", "wd", "bias", "Cin", "Cout", "wd_small", "fp8")

    def __init__(self, weight_oihw, bias, cdt, fp8=False):
        """fp8: also keep OCP e4m3 copies of the filter and of the dgrad filter (per-tensor scale from the fp32 master) and the
        delayed-scaling state of the layer's activation (e4m3) / gradient (e5m2) streams: launches whose geometry the fp8 row
        kernel covers (kernels.conv_fp8_ok) then run on v_mfma_f32_32x32x64_f8f6f4 (BASELINE configs[4])."""
        w = weight_oihw.detach().float().cuda()
        self.Cout, self.Cin = w.shape[0], w.shape[1]
        w_k = w.permute(0, 2, 3, 1).contiguous()                         # [Cout, kh, kw, Cin]
        wd_k = w.flip(2, 3).permute(1, 2, 3, 0).contiguous()             # [Cin, kh', kw', Cout]
        self.w = _as(w_k.view(self.Cout, 9 * self.Cin), cdt)
        self.bias = None if bias is None else bias.detach().float().cuda().contiguous()
        bk = 64 if cdt in K.LOWP else 32
        self.wd = self.wd_small = None
        if self.Cout % bk == 0:
            self.wd = _as(wd_k.view(self.Cin, 9 * self.Cout), cdt)
        else:
            # small Cout (conv_out -> RGB): dgrad runs as a plain GEMM over an explicit im2col of dy
            kp = ((9 * self.Cout + 31) // 32) * 32
            wds = torch.zeros(self.Cin, kp, dtype=torch.float32, device=w.device)
            wds[:, :9 * self.Cout] = wd_k.view(self.Cin, 9 * self.Cout)
            self.wd_small = _as(wds, cdt)
        self.fp8 = None
        if fp8 and cdt in K.LOWP and self.Cin % 128 == 0 and self.Cout % 128 == 0:
            sw = K.Fp8Scale(K.E4M3, w.device)
            self.fp8 = {"w": sw, "w8": K.fp8_quant(w_k.view(self.Cout, 9 * self.Cin), sw, frozen=True),
                        "wd8": K.fp8_quant(wd_k.view(self.Cin, 9 * self.Cout), sw, frozen=True),
                        "x": K.Fp8Scale(K.E4M3, w.device), "g": K.Fp8Scale(K.E5M2, w.device)}


class _Conv3x3Fn(Function):
    @staticmethod
    def forward(ctx, x, residual, P, upsample, out_dtype, gn):
        x = _contig(x)
        B, Hin, Win, Cin = x.shape
        H, W = (2 * Hin, 2 * Win) if upsample else (Hin, Win)
        y = torch.empty(B, H, W, P.Cout, dtype=out_dtype or x.dtype, device=x.device)
        sums = _gn_request(gn, y, B, H * W, P.Cout)
        ctx.f8 = (P.fp8 is not None and x.dtype in K.LOWP and y.dtype == x.dtype and K.conv_fp8_ok(B, H, W, Cin, P.Cout) and
                  (residual is None or residual.dtype == x.dtype))
        if ctx.f8:
            f = P.fp8                      # e4m3 activation (per-tensor delayed scale) x e4m3 filter on the fp8 row kernel
            x8 = _f8_twin(x, f["x"])       # written by the GroupNorm in front (groupnorm_fork(f8_for=P)), else quantised here
            if x8 is None:
                x8 = K.fp8_quant(x, f["x"])
            K.gemm_fp8(x8, f["w8"], y, B * H * W, P.Cout, 9 * Cin, f["x"], f["w"], lo_dtype=x.dtype, bias=P.bias, residual=residual,
                       conv=(H, W, Cin), flags=K.F_UPSAMPLE2X if upsample else 0,
                       gn_sums=None if sums is None else (sums, H * W, P.Cout // 32))
            K.fp8_next_scale(f["x"])
            if P.wd is not None and K.conv_fp8_ok(B, H, W, P.Cout, Cin):
                y._ffvc_gsc = f["g"]       # this launch's dgrad will run on the fp8 path too: see _GNForkFn.backward
        else:
            _f8_twin(x, None)
            K.gemm(x, P.w, y, B * H * W, P.Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, bias=P.bias,
                   residual=residual, conv=(H, W, Cin), flags=K.F_UPSAMPLE2X if upsample else 0,
                   gn_sums=None if sums is None else (sums, H * W, P.Cout // 32))
        ctx.P, ctx.upsample, ctx.geom, ctx.cdt = P, upsample, (B, H, W, Cin), x.dtype
        ctx.has_res = residual is not None
        # x = act(GroupNorm(.)): that node's backward statistics can be folded into this convolution's dgrad (no upsample in between)
        ctx.gnb = None if (upsample or ctx.f8) else getattr(x, "_ffvc_gnb", None)
        return y

    @staticmethod
    def backward(ctx, dy):
        P = ctx.P
        B, H, W, Cin = ctx.geom
        dy = _contig(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dxu = torch.empty(B, H, W, Cin, dtype=ctx.cdt, device=dy.device)
            if P.wd is not None and ctx.f8 and K.conv_fp8_ok(B, H, W, P.Cout, Cin):
                f = P.fp8                  # gradients travel as e5m2
                dy8 = _f8_twin(dy, f["g"])
                if dy8 is None:
                    dy8 = K.fp8_quant(_as(dy, ctx.cdt), f["g"])
                elif ctx.has_res and dy._ffvc_f8[2]:
                    raise RuntimeError("conv3x3 backward: the skip connection needs the 16-bit gradient, but only its fp8 bytes were written")
                K.gemm_fp8(dy8, f["wd8"], dxu, B * H * W, Cin, 9 * P.Cout, f["g"], f["w"], lo_dtype=ctx.cdt, conv=(H, W, P.Cout))
                K.fp8_next_scale(f["g"])
            elif P.wd is not None:
                _f8_twin(dy, None)
                dyt = _as(dy, ctx.cdt)
                gnb = ctx.gnb
                if gnb is not None and gnb[0].dtype == ctx.cdt and tuple(gnb[0].shape) == (B, H, W, Cin) and \
                        K.conv_gnb_ok(dyt, P.wd, dxu, gnb[0], gnb[3], gnb[4], gnb[1], gnb[2], B, H, W, P.Cout, Cin):
                    # the gradient this dgrad stores is the one GroupNorm's backward starts from: accumulate its statistics here
                    # (2 of that backward's 5 reads and its first launch), hand them over on the tensor
                    xg, gam, bet, mean, rstd, sw = gnb
                    sums = K.gn_sums_buffer(B, 32, dy.device)
                    K.gemm(dyt, P.wd, dxu, B * H * W, Cin, 9 * P.Cout, ldw=9 * P.Cout, x_mode=K.OP_CONV3X3, conv=(H, W, P.Cout),
                           gnb=(xg, mean, rstd, gam, bet, sums, sw, H * W, Cin // 32))
                    dxu._ffvc_gnb_sums = (sums, mean)
                else:
                    K.gemm(dyt, P.wd, dxu, B * H * W, Cin, 9 * P.Cout, ldw=9 * P.Cout, x_mode=K.OP_CONV3X3,
                           conv=(H, W, P.Cout))
            else:
                kp = P.wd_small.shape[1]
                cols = K.im2col3x3(dy, ctx.cdt, kp)
                K.gemm(cols, P.wd_small, dxu, B * H * W, Cin, kp, ldx=kp, ldw=kp)
            dx = K.sumpool2x2(dxu) if ctx.upsample else dxu
        dres = None
        if ctx.has_res:
            dres = dy
        return dx, dres, None, None, None, None


def conv3x3(x, P, residual=None, upsample=False, out_dtype=None, gn=False):
    """gn=True: the output feeds a GroupNorm(32) next -> its moments are accumulated by this GEMM's epilogue."""
    return _Conv3x3Fn.apply(x, residual, P, upsample, out_dtype, gn)


# ---------------------------------------------------------------------------
# Multi-head attention from a packed qkv tensor [B, T, 3*H*dh] (q | k | v blocks, heads inside)
# ---------------------------------------------------------------------------
def _pad8(n):
    return (n + 7) // 8 * 8


class _AttentionFn(Function):
    @staticmethod
    def forward(ctx, qkv, heads, scale, causal, f8_for=None):
        qkv = _contig(qkv)
        B, T, D3 = qkv.shape
        D = D3 // 3
        dh = D // heads
        if K.attn_small_ok(qkv, heads, causal):          # ViT-B/32: 50 tokens -> one wave per (cutout, head)
            ctx.save_for_backward(qkv)
            ctx.cfg = (heads, scale)
            ctx.small = True
            return K.attn_small_fwd(qkv, heads, scale)
        ctx.small = False
        if not ctx.needs_input_grad[0] and K.attn_text_ok(qkv, heads):
            # forward-only fp32 attention of a short sequence (the frozen text tower): one launch instead of GEMM + softmax + GEMM
            return K.attn_text_fwd(qkv, heads, scale, causal)
        ctx.flash = K.attn_flash_ok(qkv, heads)
        if ctx.flash:       # any length, head dim 64: online-softmax kernel, scores never reach HBM, causal blocks skipped
            f8 = None
            if f8_for is not None and _F8_PRODUCER and getattr(f8_for, "fp8", None) is not None and f8_for.fp8["x"].ready:
                f8 = f8_for.fp8["x"]   # the fp8 out_proj behind: its e4m3 operand leaves the attention epilogue (o stays: the backward reads it)
            if f8 is not None:
                o, lse, o8 = K.attn_flash_fwd(qkv, heads, scale, causal, f8=f8)
                o._ffvc_f8 = (o8, f8, False)
            else:
                o, lse = K.attn_flash_fwd(qkv, heads, scale, causal)
            ctx.save_for_backward(qkv, o, lse)
            ctx.cfg = (heads, scale, causal)
            return o
        Tp = _pad8(T)
        cdt = qkv.dtype
        BH = B * heads
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        S = torch.empty(BH, T, Tp, dtype=torch.float32, device=qkv.device)
        bs = (T * D3, dh)
        K.gemm(q, k, S, T, T, dh, ldx=D3, ldw=D3, batch=BH, batch_inner=heads, xb=bs, wb=bs,
               yb=(heads * T * Tp, T * Tp), y_map=(0, 0, Tp))
        P = torch.empty(BH, T, Tp, dtype=cdt, device=qkv.device)
        K.softmax_fwd(S, P, BH * T, T, Tp, Tp, scale=scale, causal=causal, q_len=T)
        o = torch.empty(B, T, D, dtype=cdt, device=qkv.device)
        K.gemm(P, v, o, T, dh, T, ldx=Tp, ldw=D3, w_mode=K.OP_TRANS, batch=BH, batch_inner=heads,
               xb=(heads * T * Tp, T * Tp), wb=bs, yb=(T * D, dh), y_map=(0, 0, D))
        ctx.save_for_backward(qkv, P)
        ctx.cfg = (B, T, D, heads, dh, Tp, scale)
        return o

    @staticmethod
    def backward(ctx, do):
        if ctx.small:
            (qkv,) = ctx.saved_tensors
            heads, scale = ctx.cfg
            return K.attn_small_bwd(qkv, _as(_contig(do), qkv.dtype), heads, scale), None, None, None, None
        if ctx.flash:
            qkv, o, lse = ctx.saved_tensors
            heads, scale, causal = ctx.cfg
            return K.attn_flash_bwd(qkv, o, _as(_contig(do), qkv.dtype), lse, heads, scale, causal), None, None, None, None
        qkv, P = ctx.saved_tensors
        B, T, D, heads, dh, Tp, scale = ctx.cfg
        D3, BH, cdt = 3 * D, B * heads, qkv.dtype
        do = _as(_contig(do), cdt)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
        bs = (T * D3, dh)
        pb = (heads * T * Tp, T * Tp)
        ob = (T * D, dh)
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:]
        # dP = dO V^T
        dP = torch.empty(BH, T, Tp, dtype=torch.float32, device=do.device)
        K.gemm(do, v, dP, T, T, dh, ldx=D, ldw=D3, batch=BH, batch_inner=heads, xb=ob, wb=bs, yb=pb, y_map=(0, 0, Tp))
        # dV = P^T dO
        K.gemm(P, do, dv, T, dh, T, ldx=Tp, ldw=D, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS, batch=BH, batch_inner=heads,
               xb=pb, wb=ob, yb=bs, y_map=(0, 0, D3))
        dS = torch.empty_like(P)
        K.softmax_bwd(P, dP, dS, BH * T, T, Tp, Tp, scale=scale)
        # dQ = dS K ; dK = dS^T Q
        K.gemm(dS, k, dq, T, dh, T, ldx=Tp, ldw=D3, w_mode=K.OP_TRANS, batch=BH, batch_inner=heads, xb=pb, wb=bs,
               yb=bs, y_map=(0, 0, D3))
        K.gemm(dS, q, dk, T, dh, T, ldx=Tp, ldw=D3, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS, batch=BH,
               batch_inner=heads, xb=pb, wb=bs, yb=bs, y_map=(0, 0, D3))
        return dqkv, None, None, None, None


def attention(qkv, heads, scale, causal=False, f8_for=None):
    """f8_for: the frozen fp8 `Weights` pack of the projection that consumes the result (see layernorm_fork)."""
    return _AttentionFn.apply(qkv, heads, scale, causal, f8_for)


class _AttentionTinyFn(Function):
    """softmax(scale q k^T) v for a handful of tokens and any head width, straight from the projection's own column order."""

    @staticmethod
    def forward(ctx, qkv, heads, dh, scale, layout, out_ld):
        qkv = _contig(qkv)
        ctx.save_for_backward(qkv)
        ctx.cfg = (heads, dh, scale, layout)
        return K.attn_tiny_fwd(qkv, heads, dh, scale, layout, out_ld)

    @staticmethod
    def backward(ctx, do):
        (qkv,) = ctx.saved_tensors
        heads, dh, scale, layout = ctx.cfg
        return K.attn_tiny_bwd(qkv, _as(_contig(do), qkv.dtype), heads, dh, scale, layout), None, None, None, None, None


def attention_tiny(qkv, heads, dh, scale, layout="dkh", out_ld=None):
    """qkv [B, T, >= 3*heads*dh] in the reference's '(d k h)' column order (vitgan.py:81-82) or '(k h d)' -> [B, T, out_ld]."""
    return _AttentionTinyFn.apply(qkv, heads, dh, scale, layout, out_ld)


# ---------------------------------------------------------------------------
# Glue: clamp-with-grad, VQ (straight-through), cutouts, patch embedding, loss
# ---------------------------------------------------------------------------
class _ClampFn(Function):
    @staticmethod
    def forward(ctx, x, mul, add, lo, hi, out_dtype):
        x = _contig(x)
        ctx.save_for_backward(x)
        ctx.p = (mul, add, lo, hi)
        return K.clamp_fwd(x, out_dtype or x.dtype, mul, add, lo, hi)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return K.clamp_bwd(x, _contig(g), *ctx.p), None, None, None, None, None


def clamp_with_grad(x, lo, hi, mul=1.0, add=0.0, out_dtype=None):
    """ClampWithGrad (main.py:118-132) of u = x*mul + add."""
    return _ClampFn.apply(x, float(mul), float(add), float(lo), float(hi), out_dtype)


# A/B: the fused token-mix forward saves h / act' so that the backward needs no recomputation.  Measured (profiles/r04_experiments.txt):
# 228 + 110 us per layer against 121 + 134 for the recomputing pair (the backward kernel's 242 us inside the step is the weight-
# gradient stream sharing the CUs, not its own cost), step +3 ms -> off
_TM_SAVE = os.environ.get("FFVC_TOKMIX_SAVE", "0") != "0"
_VQ_SPLIT = os.environ.get("FFVC_VQ_SPLIT", "1") != "0"   # A/B: 16-bit modes take the z . codebook^T products through split-precision f16


class _VQFn(Function):
    @staticmethod
    def forward(ctx, z, codebook, cnorm, out_dtype, force_idx, cb3=None):
        z = _contig(z)
        C = z.shape[-1]
        rows = z.numel() // C
        n = codebook.shape[0]
        if force_idx is None:
            if cb3 is not None and _VQ_SPLIT and K.vq_fused_ok(cb3.dtype, n, 3 * C):
                # the same split-precision products, the argmin folded into the GEMM's epilogue: the 1 GB distance matrix of a
                # 16384-row batch is neither written nor read back (r6: 0.97 -> 0.42 ms, bit-identical indices)
                idx = K.vq_argmin_fused(K.split3(z.view(rows, C), cb3.dtype), cb3, K.rownorm_sq(z), cnorm)
            else:
                dot = torch.empty(rows, n, dtype=torch.float32, device=z.device)
                if cb3 is not None and _VQ_SPLIT:
                    # fp32-grade products on the f16 matrix pipes (ffvc_split3: [hi|lo|hi] . [hi|hi|lo], every term but lo x lo,
                    # ~2^-22 relative): 16384 x 16384 x 256 in exact-fp32 MFMA is 1.4 ms of the step, this form a third of it
                    K.gemm(K.split3(z.view(rows, C)), cb3, dot, rows, n, 3 * C, ldx=3 * C, ldw=3 * C)
                else:
                    K.gemm(z, codebook, dot, rows, n, C, ldx=C, ldw=C)
                idx = K.vq_argmin(dot, K.rownorm_sq(z), cnorm)
        else:
            idx = force_idx.reshape(rows).to(torch.int64).contiguous()
        ctx.zdtype = z.dtype
        ctx.mark_non_differentiable(idx)
        return K.gather_rows(codebook, idx.view(z.shape[:-1]), out_dtype), idx

    @staticmethod
    def backward(ctx, g, _):
        return _as(_contig(g), ctx.zdtype), None, None, None, None, None   # straight-through (main.py:105-116,138)


def vector_quantize(z, codebook, cnorm, out_dtype, force_idx=None, cb3=None):
    """z: (..., C) fp32 -> (z_q in out_dtype, indices). Nearest code, STE gradient (main.py:134-138).
    force_idx: take these codes instead of the argmin (parity instrumentation: the argmin is a discontinuity, so
    stage-wise precision checks hand the reference's codes to the low-precision decoder).
    cb3: K.split3(codebook, weight_order=True) — the 16-bit modes' split-precision distance GEMM (None: exact fp32 MFMA)."""
    return _VQFn.apply(z, codebook, cnorm, out_dtype, force_idx, cb3)


class _CutoutsFn(Function):
    @staticmethod
    def forward(ctx, xr, noise, facs, cut, cutn, patch, mean, std, out_dtype):
        xr = _contig(xr)
        ctx.save_for_backward(xr)
        ctx.cfg = (cut, cutn, patch, std)
        return K.cutouts_fwd(xr, cut, cutn, patch, mean, std, out_dtype, noise=noise, facs=facs)

    @staticmethod
    def backward(ctx, g):
        (xr,) = ctx.saved_tensors
        cut, cutn, patch, std = ctx.cfg
        return K.cutouts_bwd(xr, _contig(g), cut, cutn, patch, std), None, None, None, None, None, None, None, None


def cutouts(xr, cut, cutn, patch, mean, std, out_dtype, noise=None, facs=None):
    return _CutoutsFn.apply(xr, noise, facs, cut, cutn, patch, mean, std, out_dtype)


class _AugmentFn(Function):
    """Fused augmentation chain on the pooled image (kernels.augment_fwd / augment_bwd)."""

    @staticmethod
    def forward(ctx, pooled, noise, facs, pinv, ainv, cmat, erase, cutn, patch, mean, std, out_dtype, coff=None, out_size=None, cj=None,
                seq=False):
        pooled = _contig(pooled)
        # the colour jitter is not linear: its backward re-evaluates the forward up to the jitter (needs the source image)
        ctx.save_for_backward(pinv, ainv, cmat, erase, pooled if cj is not None else None, coff if cj is not None else None, cj)
        ctx.cfg = (pooled.shape[0], out_size or pooled.shape[2], pooled.shape[2], cutn, patch, std, bool(seq))
        return K.augment_fwd(pooled, pinv, ainv, cmat, erase, cutn, patch, mean, std, out_dtype, noise=noise, facs=facs,
                             coff=coff, out_size=out_size, cj=cj, seq=bool(seq))

    @staticmethod
    def backward(ctx, g):
        pinv, ainv, cmat, erase, pooled, coff, cj = ctx.saved_tensors
        B, S, Ss, cutn, patch, std, seq = ctx.cfg
        return (K.augment_bwd(_contig(g), pinv, ainv, cmat, erase, B, S, cutn, patch, std, src_size=Ss, pooled=pooled, coff=coff,
                              cj=cj, seq=seq),) + (None,) * 15


def augment(pooled, params, cutn, patch, mean, std, out_dtype, noise=None, facs=None, out_size=None):
    """pooled: (B,3,Ss,Ss) fp32 -> ViT patch rows (cutn*B, (S/patch)^2, 3*patch^2), S = out_size or Ss; params from
    augment.draw_params / augment.plan (drawn for that source / output size pair)."""
    return _AugmentFn.apply(pooled, noise, facs, params["pinv"], params["ainv"], params["cmat"], params["erase"], cutn,
                            patch, mean, std, out_dtype, params.get("coff"), out_size, params.get("cj"), bool(params.get("seq", 0)))


class _SharpnessFn(Function):
    """kornia RandomSharpness on the cutout batch (N,3,S,S) fp32 (main.py:169), per-sample factor / on flags."""

    @staticmethod
    def forward(ctx, x, factor, on):
        x = _contig(x)
        ctx.save_for_backward(x, factor, on)
        return K.sharpness_fwd(x, factor, on)

    @staticmethod
    def backward(ctx, g):
        x, factor, on = ctx.saved_tensors
        return K.sharpness_bwd(_contig(g), x, factor, on), None, None


def sharpness(x, factor, on):
    return _SharpnessFn.apply(x, factor, on)


class _WarpGridFn(Function):
    """Bilinear resampling of the cutout batch at a dense coordinate field (RandomElasticTransform / RandomThinPlateSpline,
    main.py:179,181); the field does not depend on the image, so only the image gets a gradient."""

    @staticmethod
    def forward(ctx, x, grid, on):
        ctx.save_for_backward(grid, on)
        return K.warp_grid_fwd(_contig(x), grid, on)

    @staticmethod
    def backward(ctx, g):
        grid, on = ctx.saved_tensors
        return K.warp_grid_bwd(_contig(g), grid, on), None, None


def warp_grid(x, grid, on):
    return _WarpGridFn.apply(x, grid, on)


class _AvgPoolPatchesFn(Function):
    """MakeCutouts(interpolate=True): adaptive average pooling of the augmented batch, then mean/std + patch rows."""

    @staticmethod
    def forward(ctx, x, out_size, patch, mean, std, out_dtype):
        x = _contig(x)
        ctx.cfg = (x.shape[0], x.shape[2], out_size, patch, std)
        return K.avgpool_patches_fwd(x, out_size, patch, mean, std, out_dtype)

    @staticmethod
    def backward(ctx, g):
        N, S, So, patch, std = ctx.cfg
        return K.avgpool_patches_bwd(_contig(g), N, S, So, patch, std), None, None, None, None, None


def avgpool_patches(x, out_size, patch, mean, std, out_dtype):
    return _AvgPoolPatchesFn.apply(x, out_size, patch, mean, std, out_dtype)


class _PatchEmbedFn(Function):
    """tokens[n, 1+p, :] = patches[n, p, :] @ Wc^T + pos[1+p]; tokens[n, 0] = cls + pos[0]  (cloob.py:237-244)."""

    @staticmethod
    def forward(ctx, patches, W, cls_pos0, pos):
        N, Pn, Kd = patches.shape
        width = W.N
        tok = torch.empty(N, Pn + 1, width, dtype=torch.float32, device=patches.device)
        K.gemm(patches, W.sh, tok[:, 1:], N * Pn, width, Kd, ldx=Kd, ldw=Kd, y_map=(Pn, (Pn + 1) * width, width),
               residual=pos[1:], r_map=(Pn, 0, width))
        K.copy_rows(cls_pos0, 0, tok, (Pn + 1) * width, N, width)
        ctx.W, ctx.dims, ctx.cdt = W, (N, Pn, Kd, width), patches.dtype
        return tok

    @staticmethod
    def backward(ctx, dtok):
        W = ctx.W
        N, Pn, Kd, width = ctx.dims
        dt = _as(_contig(dtok), ctx.cdt)
        dp = torch.empty(N, Pn, Kd, dtype=ctx.cdt, device=dtok.device)
        K.gemm(dt[:, 1:], W.sht, dp, N * Pn, Kd, width, ldx=width, ldw=width, x_map=(Pn, (Pn + 1) * width))
        return dp, None, None, None


def patch_embed(patches, W, cls_pos0, pos):
    return _PatchEmbedFn.apply(patches, W, cls_pos0, pos)


class _SphericalLossFn(Function):
    @staticmethod
    def forward(ctx, embed, feats, coef):
        loss, dembed = K.spherical_loss(_contig(embed), _contig(feats), coef, want_grad=True)
        ctx.save_for_backward(dembed)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dembed,) = ctx.saved_tensors
        return K.mul_dev_scalar(dembed, _contig(g).float()), None, None


def spherical_loss(embed, feats, coef=1.0):
    """main.py:801-811 (repeat = 1): embed [cutn*B, D] fp32, feats [B, D] fp32 (no grad) -> scalar."""
    return _SphericalLossFn.apply(embed, feats, float(coef))


class _MeanSqFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(x)
        ctx.save_for_backward(x)
        return K.mean_sq(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return K.mean_sq_bwd(x, _contig(g).float())


def mean_sq(x):
    """l2 regulariser (main.py:758-762): mean(x^2) of a contiguous fp32 tensor."""
    return _MeanSqFn.apply(x)


class _TVLossFn(Function):
    @staticmethod
    def forward(ctx, x):
        x = _contig(x)
        ctx.save_for_backward(x)
        return K.tv_loss_fwd(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return K.tv_loss_bwd(x, _contig(g).float())


def tv_loss_nhwc(x):
    """tv_loss (main.py:423-428) of an NHWC fp32 image batch (B, H, W, C)."""
    return _TVLossFn.apply(x)


def default_scale(dh):
    return 1.0 / math.sqrt(dh)


# ---------------------------------------------------------------------------
# layout / dtype plumbing with gradients
# ---------------------------------------------------------------------------
class _BroadcastRowsFn(Function):
    """p [T, D] (fp32 parameter) -> [B, T, D] (+ x [B, T, D]): the learned position table added to / handed on as the token stream
    (vitgan.py:255,300).  One row-broadcast copy (+ one axpby); backward: column sums over the batch into the table's gradient."""

    @staticmethod
    def forward(ctx, p, x, B, scale):
        p = _contig(p)
        n = p.numel()
        out = torch.empty(B, *p.shape, dtype=torch.float32, device=p.device)
        K.copy_rows(p.view(1, n), 0, out.view(B, n), n, B, n)
        if x is not None:
            K.axpby(_contig(x), out, 1.0, float(scale))          # out = x + scale * p
        elif scale != 1.0:
            raise ValueError("broadcast_rows: a scale needs an x to add to")
        ctx.has_x, ctx.scale = x is not None, float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        g = _contig(g)
        B = g.shape[0]
        n = g.numel() // B
        dp = torch.empty(n, dtype=torch.float32, device=g.device)
        K.colsum(g.view(B, n), dp)
        if ctx.scale != 1.0:
            K.axpby(dp, dp, ctx.scale, 0.0)
        return dp.view(g.shape[1:]), (g if ctx.has_x else None), None, None


def broadcast_rows(p, B, x=None, scale=1.0):
    """[T, D] -> [B, T, D]: x + scale * p with p broadcast over the batch (fp32); without x, B copies of p."""
    return _BroadcastRowsFn.apply(p, x, B, scale)


class _TransposeFn(Function):
    @staticmethod
    def forward(ctx, x, out_dtype):
        ctx.in_dtype = x.dtype
        return K.transpose(_contig(x), out_dtype or x.dtype)

    @staticmethod
    def backward(ctx, g):
        return K.transpose(_contig(g), ctx.in_dtype), None


def transpose_last2(x, out_dtype=None):
    """(..., R, C) -> (..., C, R) contiguous, optional dtype conversion (einops Rearrange, mlp_mixer_pytorch.py:31)."""
    return _TransposeFn.apply(x, out_dtype)


class _CastFn(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.in_dtype = x.dtype
        return K.cast(_contig(x), dtype)

    @staticmethod
    def backward(ctx, g):
        return K.cast(_contig(g), ctx.in_dtype), None


def cast(x, dtype):
    return x if x.dtype == dtype else _CastFn.apply(x, dtype)


# ---------------------------------------------------------------------------
# VitGAN / x-transformer specific ops
# ---------------------------------------------------------------------------
class _SLNForkFn(Function):
    """Self-modulated LayerNorm (vitgan.py:8-21), fork form: -> (gamma_s*w*LN(hl) + beta_s*w, identity alias of hl)."""

    @staticmethod
    def forward(ctx, hl, w, gamma, beta, gs, bs, out_dtype, share=None, last=False):
        ctx.share, ctx.last = share, last
        hl, w = _contig(hl), _contig(w)
        g, b, gsd, bsd = gamma.detach(), beta.detach(), gs.detach().reshape(1), bs.detach().reshape(1)
        y, mean, rstd = K.sln_fwd(hl, w, g, b, gsd, bsd, out_dtype)
        ctx.save_for_backward(hl, w, g, b, gsd, bsd, mean, rstd)
        ctx.sshape = gs.shape
        arena = getattr(gamma, "_ffvc_arena", None)
        ctx.params = (gamma, beta, gs, bs) if (arena is not None and gamma.requires_grad and all(
            getattr(p, "_ffvc_arena", None) is arena for p in (beta, gs, bs))) else None
        ctx.set_materialize_grads(False)
        return y, hl.view_as(hl)

    @staticmethod
    def backward(ctx, dy, dres):
        hl, w, g, b, gsd, bsd, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None, None, None, None
        if dres is not None:
            dres = _as(_contig(dres), torch.float32)
        if ctx.params is not None and not _SLN_INPLACE:   # gradients straight into the flat bucket (see _LNForkFn); the default form
            gamma, beta, gs, bs = ctx.params
            sc = torch.zeros(2, dtype=torch.float32, device=hl.device)
            dhl, dw = K.sln_bwd_acc(_contig(dy), hl, w, g, b, gsd, bsd, mean, rstd, _grad_buf(gamma), _grad_buf(beta), sc,
                                    dres=dres)
            _grad_buf(gs).view(-1).add_(sc[0:1])          # the two scalar parameters live apart in the bucket
            _grad_buf(bs).view(-1).add_(sc[1:2])
            gamma._ffvc_arena.grad_written(gamma, beta, gs, bs)
            return dhl, dw, None, None, None, None, None, None, None
        if ctx.params is not None:              # FFVC_SLN_INPLACE=1: no scratch, no scalar adds, optionally one running sum for w's gradient
            gamma, beta, gs, bs = ctx.params
            sh = ctx.share
            # (the two scalar parameters live apart in the bucket: ffvc_sln_bwd_acc2 takes one address each)
            dhl, dw = K.sln_bwd_acc2(_contig(dy), hl, w, g, b, gsd, bsd, mean, rstd, _grad_buf(gamma), _grad_buf(beta),
                                     _grad_buf(gs).view(-1), _grad_buf(bs).view(-1), dres=dres, dw=None if sh is None else sh.buf)
            gamma._ffvc_arena.grad_written(gamma, beta, gs, bs)
            if sh is not None:                  # one running sum for the modulation input; the consumer whose backward runs last hands it on
                sh.buf = None if ctx.last else dw
                if not ctx.last:
                    dw = None
            return dhl, dw, None, None, None, None, None, None, None
        dhl, dw, dg, db, dgs, dbs = K.sln_bwd(_contig(dy), hl, w, g, b, gsd, bsd, mean, rstd, dres=dres)
        return dhl, dw, dg, db, dgs.view(ctx.sshape), dbs.view(ctx.sshape), None, None, None


# FFVC_SLN_INPLACE=1: ffvc_sln_bwd_acc2 writes the two scalar SLN gradients straight to their bucket slots and (with a SharedGrad holder,
# FFVC_SLN_SHARE=1) keeps ONE running sum for the shared modulation input: 65 fills + 130 scalar adds + 64 tensor adds per cfg3 step less.
# OFF by default: with that kernel instantiation in the backward pass 3-17 % of the passes of a 9-block generator come out with ONE sample's
# gradients changed at f16-rounding level (tools/r6/vitgan_determinism_old.py: 0 of 220 with the default kernel, 3 / 11 / 16 of 100 with
# this one on the same boxes; more with the grouped / padded weight-gradient launches on top).  Every kernel involved is bit-reproducible
# on its own (tools/r6/vit_ops_stress.py, skfix_stress.py), the mechanism was not found, and the step time does not depend on it
#  — so the round-5 form stays the product path.
_SLN_INPLACE = os.environ.get("FFVC_SLN_INPLACE", "0") != "0"


class SharedGrad:
    """Running gradient sum of a tensor that many nodes of one forward pass consume (the modulation input of every SLN in a VitGAN
    generator, vitgan.py:132,256).  A fresh holder per forward pass; the consumers' backward kernels add into `buf`, and the one
    that autograd runs LAST (`last=True`: the first consumer of the forward pass, which everything later depends on) returns it."""

    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


def sln_fork(hl, w, ln_weight, ln_bias, gamma_s, beta_s, out_dtype, share=None, last=False):
    return _SLNForkFn.apply(hl, w, ln_weight, ln_bias, gamma_s, beta_s, out_dtype, share, last)


class _TransposePadFn(Function):
    """(..., R, C) -> (..., C, Rp) with zero padding of the new last dim (VitGAN '(d k h)' regroup, vitgan.py:82)."""

    @staticmethod
    def forward(ctx, x, pad_to):
        x = _contig(x)
        ctx.R, ctx.C = x.shape[-2], x.shape[-1]
        return K.transpose(x, pad_to=pad_to)

    @staticmethod
    def backward(ctx, g):
        g = _contig(g)
        R, C, Rp = ctx.R, ctx.C, g.shape[-1]
        gt = K.transpose(g)                                    # (..., Rp, C)
        if Rp == R:
            return gt, None
        batch = g.numel() // (C * Rp)
        out = K.copy2d(gt, Rp * C, batch, R * C, R * C, g.dtype)
        return out.view(*g.shape[:-2], R, C), None


def transpose_pad(x, pad_to):
    return _TransposePadFn.apply(x, pad_to)


class _Copy2dFn(Function):
    """rows x cols block copy between leading dims (drop / add per-head zero padding)."""

    @staticmethod
    def forward(ctx, x, rows, cols, src_ld, dst_cols):
        x = _contig(x)
        if x.numel() != rows * src_ld:
            raise ValueError(f"copy2d: {tuple(x.shape)} is not {rows} rows of {src_ld}")
        ctx.cfg = (rows, cols, src_ld, dst_cols, x.shape)
        return K.copy2d(x, src_ld, rows, cols, dst_cols, x.dtype)

    @staticmethod
    def backward(ctx, g):
        rows, cols, src_ld, dst_cols, xshape = ctx.cfg
        gx = K.copy2d(_contig(g), dst_cols, rows, min(cols, dst_cols), src_ld, g.dtype)
        return gx.view(xshape), None, None, None, None          # the input may have been any view of the rows x src_ld block


def copy2d(x, rows, cols, src_ld, dst_cols):
    return _Copy2dFn.apply(x, rows, cols, src_ld, dst_cols)


class _QKV3Fn(Function):
    """qkv[..., 0:D | D:2D | 2D:3D] = x Wq^T | x Wk^T | x Wv^T (three bias-free Linears of x-transformers' Attention)."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, Wq, Wk, Wv):
        cdt = Wq.sh.dtype
        x = _contig(x)
        D, Kd = Wq.N, Wq.K
        rows = x.numel() // Kd
        qkv = torch.empty(*x.shape[:-1], 3 * D, dtype=cdt, device=x.device)
        for i, W in enumerate((Wq, Wk, Wv)):
            K.gemm(x, W.sh, qkv.view(-1)[i * D:], rows, D, Kd, ldx=Kd, ldw=Kd, y_map=(0, 0, 3 * D))
        ctx.Ws, ctx.rows = (Wq, Wk, Wv), rows
        ctx.train = wq.requires_grad
        ctx.save_for_backward(x)
        return qkv

    @staticmethod
    def backward(ctx, dqkv):
        (x,) = ctx.saved_tensors
        Wq = ctx.Ws[0]
        cdt, D, Kd, rows = Wq.sh.dtype, Wq.N, Wq.K, ctx.rows
        dqkv = _as(_contig(dqkv), cdt)
        dx = torch.empty_like(x)
        for i, W in enumerate(ctx.Ws):
            g = dqkv.view(-1)[i * D:]
            K.gemm(g, W.sht, dx, rows, Kd, D, ldx=3 * D, ldw=D, residual=dx if i else None)
            if ctx.train:
                _wgrad(g, x, W, rows, ldy=3 * D)
        return dx, None, None, None, None, None, None


def qkv3(x, Wq, Wk, Wv):
    return _QKV3Fn.apply(x, Wq.weight, Wk.weight, Wv.weight, Wq, Wk, Wv)
