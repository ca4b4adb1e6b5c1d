"""Thin Python launchers over the C ABI (one function per `ffvc_*` entry point).

Every launcher takes torch CUDA tensors (torch is only the allocator / stream owner),
passes raw device pointers + explicit dims to libffvc_hip.so and enqueues on torch's
current HIP stream.  Nothing here computes with torch ops.
"""
import ctypes
from ctypes import byref, c_int32, c_int64

import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LRELU, ACT_NONE, ACT_QUICKGELU, ACT_TANH, BF16, F_AUX_ACTGRAD, F16, F32, F_ACCUM_OUT, F_ATOMIC_OUT, F_BIAS_ALONG_M,  # noqa: F401
                   F_MUL_ACT_GRAD, F_OUT_F32, F_RES_F32, F_TR_SAFE, F_UPSAMPLE2X, F_WRITE_PREACT,
                   OP_CONV3X3, OP_KMAJOR, OP_TRANS, GemmDesc)


LOWP = (torch.bfloat16, torch.float16)      # 16-bit storage formats (fp32 accumulate); same kernels, same MFMA rate


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


# Optional per-launch timing (bench.py roofline leg): when PROFILE is a list, every ffvc_gemm launch is
# bracketed by HIP events recorded on the launch stream and (kernel class, algorithmic flops, events) is appended.
PROFILE = None
_GEMM_CLASS = {(OP_KMAJOR, OP_KMAJOR): "gemm_nt", (OP_CONV3X3, OP_KMAJOR): "conv3x3", (OP_TRANS, OP_TRANS): "gemm_tn",
               (OP_KMAJOR, OP_TRANS): "gemm_nn", (OP_TRANS, OP_KMAJOR): "gemm_tk"}


# Optional capture for replay (bench.py `roofline.attainable_ms`): when REPLAY is a list, every ffvc_gemm launch appends
# (class, key, descriptor, tensors kept alive) so that the SAME launch — same pointers, epilogue, flags — can be re-issued alone,
# back to back, after the step (replay_gemm): its isolated duration next to the in-step one.
REPLAY = None


def replay_gemm(desc, n=1):
    """Re-issue a captured launch n times: `desc` is a GemmDesc (plain ffvc_gemm) or a zero-argument callable that performs the
    whole launch group (row-split + skinny remainder, the fp8 entry points)."""
    if callable(desc):
        for _ in range(n):
            desc()
        return
    lib = _lib.load()
    for _ in range(n):
        _lib.check(lib.ffvc_gemm(byref(desc), stream_ptr()), "ffvc_gemm (replay)")


# Optional per-launch timing of the HBM-bound kernels (bench.py `hbm_kernels`): when HBM_PROFILE is a list, the launchers
# below bracket their kernel with HIP events on the launch stream and append (name, algorithmic bytes, events).
HBM_PROFILE = None


class _hbm:
    """nbytes = algorithmic bytes of the kernel AS BUILT; min_bytes = what any implementation has to move (every operand read
    once, every result written once) — bench.py's roofline.hw_bound_ms prices the kernel at min_bytes."""

    def __init__(self, name, nbytes, min_bytes=None):
        self.name, self.nbytes = name, nbytes
        self.min_bytes = nbytes if min_bytes is None else min_bytes

    def __enter__(self):
        if HBM_PROFILE is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if HBM_PROFILE is not None:
            self.e1.record()
            HBM_PROFILE.append((self.name, float(self.nbytes), self.e0, self.e1))
            HBM_MIN_BYTES.append(float(self.min_bytes))
        return False


# The remaining kernels of a step (fused token-mixing MLP, attention, augmentation, reductions, glue): when AUX_PROFILE is a list
# their launchers append (name, algorithmic FLOPs, minimal bytes, events) — bench.py's hardware bound prices each at
# max(FLOPs / best sustained GEMM rate, bytes / 6.3 TB/s) next to its measured time.
AUX_PROFILE = None
HBM_MIN_BYTES = []         # parallel to HBM_PROFILE (kept separate: the 4-tuple layout of HBM_PROFILE is read in several places)


class _aux:
    def __init__(self, name, flops=0.0, nbytes=0.0):
        self.name, self.flops, self.nbytes = name, float(flops), float(nbytes)

    def __enter__(self):
        if AUX_PROFILE is not None:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if AUX_PROFILE is not None:
            self.e1.record()
            AUX_PROFILE.append((self.name, self.flops, self.nbytes, self.e0, self.e1))
        return False


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    if dt == torch.float16:
        return F16
    raise TypeError(f"unsupported compute dtype {dt}")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.FFVCError("ffvc kernels need CUDA(HIP) tensors; there is no CPU fallback")


def _req(dtype, *ts, contiguous=True):
    """Launcher-side contract of the raw-pointer ABI: device tensor, expected dtype, contiguous (None is allowed)."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.FFVCError("ffvc kernels need CUDA(HIP) tensors; there is no CPU fallback")
        if t.dtype != dtype:
            raise TypeError(f"ffvc launcher: expected a {dtype} tensor, got {t.dtype}")
        if contiguous and not t.is_contiguous():
            raise TypeError("ffvc launcher: expected a contiguous tensor")


def _req_f32(*ts):
    _req(torch.float32, *ts)


def gemm(x, w, y, M, N, K, *, ldx=0, ldw=0, x_mode=OP_KMAJOR, w_mode=OP_KMAJOR, bias=None,
         residual=None, aux=None, ldaux=0, act=ACT_NONE, flags=0, split_k=1, alpha=1.0,
         kseg=0, xkso=0, wkso=0, y_map=None, r_map=None, batch=1, batch_inner=1,
         xb=(0, 0), wb=(0, 0), yb=(0, 0), rb=(0, 0), ab=(0, 0), conv=None, x_map=None, slab_stride=0, gn_sums=None,
         colsum=None, gnb=None, probe_only=False, vq=None):
    """Enqueue `ffvc_gemm`. See include/ffvc.h for the index maps.

    y_map / r_map = (mi, so, sm): row offset(m) = (m // mi) * so + (m % mi) * sm (mi = 0: m * sm).
    ?b = (outer, inner) batch strides; conv = (H, W, Cin) of the OUTPUT grid for OP_CONV3X3.
    """
    _need_cuda(x, w, y, bias, residual, aux)
    d = GemmDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.bias, d.residual, d.aux = _ptr(bias), _ptr(residual), _ptr(aux)
    d.M, d.N, d.K = M, N, K
    d.x_mode, d.w_mode = x_mode, w_mode
    d.in_dtype = dtype_code(x.dtype)
    if w.dtype != x.dtype:
        raise TypeError(f"gemm operand dtypes differ: {x.dtype} vs {w.dtype}")
    if vq is not None:
        pass                                      # nothing is stored: y is the packed (distance, index) word per row
    elif y.dtype == torch.float32 and x.dtype != torch.float32:
        flags |= F_OUT_F32
    elif y.dtype != x.dtype:
        raise TypeError(f"gemm output dtype {y.dtype} incompatible with input {x.dtype}")
    if x.dtype == torch.float32:
        flags |= F_OUT_F32
    if residual is not None:
        if residual.dtype == torch.float32:
            flags |= F_RES_F32
        elif residual.dtype != x.dtype:
            raise TypeError("gemm residual dtype must be fp32 or the input dtype")
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("gemm bias must be fp32")
    if aux is not None and aux.dtype != x.dtype:
        raise TypeError("gemm aux dtype must equal the input dtype")
    d.act, d.flags, d.split_k, d.alpha = act, flags, split_k, alpha
    d.ldx, d.ldw, d.ldaux = ldx, ldw, ldaux
    d.kseg, d.xkso, d.wkso = kseg, xkso, wkso
    ym = y_map if y_map is not None else (0, 0, N)
    rm = r_map if r_map is not None else (0, 0, N)
    d.y_mi, d.y_so, d.y_sm = ym
    d.r_mi, d.r_so, d.r_sm = rm
    d.batch, d.batch_inner = batch, batch_inner
    d.xbo, d.xbi = xb
    d.wbo, d.wbi = wb
    d.ybo, d.ybi = yb
    d.rbo, d.rbi = rb
    d.abo, d.abi = ab
    if conv is not None:
        d.conv_H, d.conv_W, d.conv_Cin = conv
    if x_map is not None:
        d.x_mi, d.x_so = x_map
    d.slab_stride = slab_stride
    if gn_sums is not None:                       # (fp64 [images, groups, 2] zeroed buffer, pixels per image, channels per group)
        buf, hw, cpg = gn_sums
        if buf.dtype != torch.float64 or not buf.is_cuda or not buf.is_contiguous():
            raise TypeError("gemm: gn_sums buffer must be a contiguous fp64 device tensor")
        d.gn_sums, d.gn_hw, d.gn_cpg = buf.data_ptr(), hw, cpg
        d.flags |= _lib.F_GN_SUMS
    if gnb is not None:
        # (x, mean, rstd, gamma, beta, sums fp64 [images, groups, 2] zeroed, swish, pixels per image, channels per group): the stored
        # output is the gradient of act(GroupNorm(x)) -> the launch also accumulates that node's backward statistics (F_GNB_SUMS)
        gx, gmean, grstd, ggamma, gbeta, gsums, gswish, ghw, gcpg = gnb
        d.gnb_x, d.gnb_mean, d.gnb_rstd = gx.data_ptr(), gmean.data_ptr(), grstd.data_ptr()
        d.gnb_gamma, d.gnb_beta, d.gnb_sums, d.gnb_swish = ggamma.data_ptr(), gbeta.data_ptr(), gsums.data_ptr(), int(bool(gswish))
        d.gn_hw, d.gn_cpg = ghw, gcpg
        d.flags |= _lib.F_GNB_SUMS
    if vq is not None:                            # (xn fp32 [M], cn fp32 [N], packed int64 [M] = all ones): FFVC_F_VQ_ARGMIN
        vxn, vcn, vout = vq
        _req_f32(vxn, vcn)
        if vout.dtype != torch.int64 or not vout.is_contiguous() or vout.numel() != M or vxn.numel() != M or vcn.numel() != N:
            raise TypeError("gemm: vq = (xn fp32 [M], cn fp32 [N], packed int64 [M])")
        d.vq_xn, d.vq_cn, d.vq_out = vxn.data_ptr(), vcn.data_ptr(), vout.data_ptr()
        d.flags |= _lib.F_VQ_ARGMIN
    if probe_only:
        return bool(_lib.load().ffvc_gemm_gnb_probe(byref(d), stream_ptr()))
    if colsum is not None:                        # fp32 [N]: += column sums of the stored output (colsum_fusable() first)
        if colsum.dtype != torch.float32 or not colsum.is_cuda or not colsum.is_contiguous() or colsum.numel() != N:
            raise TypeError("gemm: colsum must be a contiguous fp32 [N] device tensor")
        d.colsum = colsum.data_ptr()
        d.flags |= _lib.F_COLSUM
    lib = _lib.load()
    # 64 x 257 rows (ViT-L/14 at 64 cutouts) = 64 whole 256-row tiles + 64 rows: inside one launch the remainder costs a nearly empty extra
    # round (tools/f16_rows_bench.py: 16448 vs 16384 rows 62 vs 36 us at N=1024 K=1024, 152 vs 108 at N=1024 K=4096); the plain-epilogue
    # Linear kinds therefore run as 16384 rows on the tiled kernel + the remainder on ffvc_gemm_skinny (K split across the waves of a
    # workgroup).  Row-separable by construction: bias / residual / store only.
    M0 = M - M % 256
    tail = M - M0
    skinny = (_GEMM_ROWSPLIT and x.dtype in LOWP and x_mode == OP_KMAJOR and w_mode == OP_KMAJOR and M0 >= 8192 and 0 < tail <= 64 and
              batch == 1 and split_k == 1 and act == ACT_NONE and aux is None and colsum is None and gn_sums is None and conv is None and vq is None and
              x_map is None and y_map is None and r_map is None and kseg == 0 and slab_stride == 0 and alpha == 1.0 and
              (flags & ~(F_OUT_F32 | F_RES_F32)) == 0 and (ldx in (0, K)) and (ldw in (0, K)) and
              bool(lib.ffvc_gemm_skinny_ok(tail, N, K)))
    if skinny:
        d.M = M0

    def issue():
        _lib.check(lib.ffvc_gemm(byref(d), stream_ptr()), "ffvc_gemm")
        if skinny:
            _lib.check(lib.ffvc_gemm_skinny(x.data_ptr() + M0 * K * x.element_size(), w.data_ptr(), dtype_code(x.dtype),
                                            y.data_ptr() + M0 * N * y.element_size(), dtype_code(y.dtype), _ptr(bias),
                                            (residual.data_ptr() + M0 * N * residual.element_size()) if residual is not None else None,
                                            dtype_code(residual.dtype) if residual is not None else 0, tail, N, K, stream_ptr()),
                       "ffvc_gemm_skinny")

    if REPLAY is not None:       # EVERY profiled launch has a replay entry (bench.attainable_leg aligns the two lists index by index)
        REPLAY.append((_GEMM_CLASS[(x_mode, w_mode)] + {torch.float32: "_f32", torch.float16: "_f16"}.get(x.dtype, "_bf16"),
                       (M, N, K, max(1, batch), split_k, int(d.flags), int(act)), issue if skinny else d,
                       (x, w, y, bias, residual, aux, colsum, gn_sums, gnb, vq)))
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    issue()
    if PROFILE is not None:
        e1.record()
        PROFILE.append((_GEMM_CLASS[(x_mode, w_mode)] + {torch.float32: "_f32", torch.float16: "_f16"}.get(x.dtype, "_bf16"),
                        2.0 * M * N * K * max(1, batch), e0, e1, (M, N, K, max(1, batch), split_k)))
    return y


def gemm_skinny(x, w, y, M, N, K, bias=None, residual=None):
    """y[M <= 64, N] = X W^T (+ bias) (+ residual), 16-bit K-major operands, on the K-split-across-waves kernel (see gemm's row split)."""
    _need_cuda(x, w, y, bias, residual)
    _call("ffvc_gemm_skinny", x.data_ptr(), w.data_ptr(), dtype_code(x.dtype), y.data_ptr(), dtype_code(y.dtype), _ptr(bias), _ptr(residual),
          dtype_code(residual.dtype) if residual is not None else 0, M, N, K, stream_ptr())
    return y


def device_info():
    lib = _lib.load()
    ncu, clk, mem = c_int32(), c_int32(), c_int64()
    _lib.check(lib.ffvc_device_info(byref(ncu), byref(clk), byref(mem)), "ffvc_device_info")
    return {"n_cu": ncu.value, "clock_khz": clk.value, "hbm_bytes": mem.value}


def probe_tr16():
    out = torch.empty(256, dtype=torch.int16, device="cuda")
    _lib.check(_lib.load().ffvc_probe_tr16(out.data_ptr(), stream_ptr()), "ffvc_probe_tr16")
    torch.cuda.synchronize()
    return out.cpu().view(64, 4)


# ---------------------------------------------------------------------------
# norm / softmax
# ---------------------------------------------------------------------------
def _call(name, *args):
    lib = _lib.load()
    _lib.check(getattr(lib, name)(*args), name)


# the `f8_only` outputs of the normalisation kernels are buffers the kernel does NOT write (their one consumer takes the
# fp8 bytes).  FFVC_DEBUG_F8_NAN=1 fills them with NaN, so that anything that does read one — a view or clone that lost the
# `_ffvc_f8` attribute, an autograd accumulation — poisons the loss visibly instead of consuming uninitialised memory
# (tests/test_models_gpu.py runs the fp8 models under it).
DEBUG_F8_NAN = os.environ.get("FFVC_DEBUG_F8_NAN", "0") != "0"


def _unwritten(t, only):
    if only and DEBUG_F8_NAN:
        t.fill_(float("nan"))
    return t


def layernorm_fwd(x, gamma, beta, out_dtype, eps=1e-5, f8=None, f8_only=False):
    """x: (..., dim) fp32|bf16 contiguous -> (y[out_dtype], mean, rstd).
    f8 (an initialised Fp8Scale, 16-bit out_dtype, dim % 4 == 0): -> (y, mean, rstd, y8) with y8 = the fp8 bytes fp8_quant(y, f8) would
    give; f8_only: y is NOT written (uninitialised) — for the one consumer that takes y8."""
    _req_f32(gamma, beta)
    _need_cuda(x, gamma, beta)
    dim = x.shape[-1]
    rows = x.numel() // dim
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if f8 is not None:
        if out_dtype not in LOWP or not f8.ready or dim % 4:
            raise TypeError("layernorm_fwd: fp8 output needs a 16-bit out_dtype, dim % 4 == 0 and an initialised Fp8Scale")
        fp8_begin(f8)
        y8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        with _hbm("layernorm_fwd", x.numel() * (x.element_size() + 1 + (0 if f8_only else y.element_size()))):
            _call("ffvc_layernorm_fwd_f8", x.data_ptr(), dtype_code(x.dtype), gamma.data_ptr(), beta.data_ptr(),
                  0 if f8_only else y.data_ptr(), dtype_code(out_dtype), y8.data_ptr(), f8.state.data_ptr(), f8.fmt, mean.data_ptr(),
                  rstd.data_ptr(), rows, dim, eps, stream_ptr())
        return _unwritten(y, f8_only), mean, rstd, y8
    with _hbm("layernorm_fwd", x.numel() * (x.element_size() + y.element_size())):
        _call("ffvc_layernorm_fwd", x.data_ptr(), dtype_code(x.dtype), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
              dtype_code(out_dtype), mean.data_ptr(), rstd.data_ptr(), rows, dim, eps, stream_ptr())
    return y, mean, rstd


def _dx_lo(x, want_lo, lo_dtype=torch.bfloat16):
    return torch.empty(x.shape, dtype=lo_dtype, device=x.device) if (want_lo and x.dtype == torch.float32) else None


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, want_param_grads=False, want_lo=False, f8=None):
    """-> (dx [x.dtype], dgamma|None, dbeta|None).  want_lo: also write a bf16 copy of an fp32 dx (`dx._ffvc_lo`).
    f8 (an initialised Fp8Scale; frozen layer, fp32 x, 16-bit dy): -> (dx, dx8) with dx8 = the fp8 bytes fp8_quant would make of that
    16-bit copy, which is then not written."""
    _req_f32(gamma, mean, rstd)
    _need_cuda(dy, x, gamma)
    dim = x.shape[-1]
    rows = x.numel() // dim
    dx = torch.empty_like(x)
    if dres is not None and dres.dtype != x.dtype:
        raise TypeError("layernorm_bwd: dres dtype must equal x dtype")
    if f8 is not None:
        if want_param_grads or x.dtype != torch.float32 or dy.dtype not in LOWP or dim % 4 or not f8.ready:
            raise TypeError("layernorm_bwd: the fp8 form needs a frozen layer, fp32 x, 16-bit dy, dim % 4 == 0 and an initialised Fp8Scale")
        fp8_begin(f8)
        dx8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        with _hbm("layernorm_bwd", x.numel() * (dy.element_size() + 8 + (4 if dres is not None else 0) + 1)):
            _call("ffvc_layernorm_bwd_f8", dy.data_ptr(), dtype_code(dy.dtype), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), _ptr(dres), dx.data_ptr(), dx8.data_ptr(), f8.state.data_ptr(), f8.fmt, rows, dim, stream_ptr())
        return dx, dx8
    pg = pb = None
    lo = _dx_lo(x, want_lo, dy.dtype)
    if want_param_grads:
        nb = _lib.load().ffvc_layernorm_bwd_blocks(rows)
        pg = torch.empty(nb, dim, dtype=torch.float32, device=x.device)
        pb = torch.empty(nb, dim, dtype=torch.float32, device=x.device)
    _call("ffvc_layernorm_bwd", dy.data_ptr(), dtype_code(dy.dtype), x.data_ptr(), dtype_code(x.dtype),
          gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dx.data_ptr(), _ptr(pg), _ptr(pb),
          _ptr(lo), rows, dim, stream_ptr())
    if lo is not None:
        dx._ffvc_lo = lo
    if not want_param_grads:
        return dx, None, None
    dg = torch.empty(dim, dtype=torch.float32, device=x.device)
    db = torch.empty(dim, dtype=torch.float32, device=x.device)
    colsum(pg, dg)
    colsum(pb, db)
    return dx, dg, db


def layernorm_bwd_acc(dy, x, gamma, mean, rstd, dgamma, dbeta, dres=None, want_lo=False):
    """-> dx; dgamma / dbeta (fp32 [dim], contiguous) are accumulated in place."""
    _req_f32(gamma, mean, rstd, dgamma, dbeta)
    _need_cuda(dy, x, gamma, dgamma, dbeta)
    dim = x.shape[-1]
    rows = x.numel() // dim
    dx = torch.empty_like(x)
    if dres is not None and dres.dtype != x.dtype:
        raise TypeError("layernorm_bwd: dres dtype must equal x dtype")
    if dgamma.dtype != torch.float32 or dbeta.dtype != torch.float32 or not dgamma.is_contiguous() or not dbeta.is_contiguous():
        raise TypeError("layernorm_bwd_acc: gradients must be contiguous fp32")
    lo = _dx_lo(x, want_lo, dy.dtype)
    nb = x.numel() * (dy.element_size() + 2 * x.element_size() + (x.element_size() if dres is not None else 0) +
                      (2 if lo is not None else 0))
    with _hbm("layernorm_bwd", nb):
        _call("ffvc_layernorm_bwd_acc", dy.data_ptr(), dtype_code(dy.dtype), x.data_ptr(), dtype_code(x.dtype),
              gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dx.data_ptr(), dgamma.data_ptr(),
              dbeta.data_ptr(), _ptr(lo), rows, dim, stream_ptr())
    if lo is not None:
        dx._ffvc_lo = lo
    return dx


# ---------------------------------------------------------------------------------------------------------------
# fp8 path of the frozen towers (cfg5): per-tensor delayed scaling + ffvc_gemm_fp8
# ---------------------------------------------------------------------------------------------------------------
E4M3, E5M2 = 0, 1
FP8_MARGIN = 2.0          # headroom of the delayed scale: next step's values may exceed this step's amax by this factor


_GEMM_ROWSPLIT = os.environ.get("FFVC_GEMM_ROWSPLIT", "1") != "0"  # A/B: 16-bit 64 x 257-row Linear GEMMs as 16384 rows + ffvc_gemm_skinny
_FP8_SKINNY = os.environ.get("FFVC_FP8_SKINNY", "1") != "0"        # A/B: the 64-row remainder on ffvc_gemm_fp8_skinny
_FP8_ROWSPLIT = os.environ.get("FFVC_FP8_ROWSPLIT", "1") != "0"    # A/B: 64 x 257-row fp8 GEMMs as 16384 + 64 rows
_F8_POOLS = {}        # device -> list of [buf [1024, 4] fp32, rows handed out]
_F8_PENDING = []      # scales whose update (amax -> next scale) has not been enqueued yet


class Fp8Scale:
    """Device-resident scale of one tensor stream: state = [scale, running amax, 1/scale, format code].  The first use measures the
    tensor itself (current scaling); afterwards the amax seen while quantising step t sets the scale of step t+1
    (delayed scaling: no extra pass over the tensor, no host synchronisation).  States live in a pool so that ONE launch
    (fp8_flush_updates, called at the top of a train step) refreshes every stream that saw a tensor."""

    __slots__ = ("state", "fmt", "ready", "pending")

    def __init__(self, fmt, device):
        device = torch.device(device)
        pools = _F8_POOLS.setdefault(device, [])
        if not pools or pools[-1][1] == pools[-1][0].shape[0]:
            pools.append([torch.zeros(1024, 4, dtype=torch.float32, device=device), 0])
        buf, n = pools[-1]
        self.state = buf[n]
        pools[-1][1] = n + 1
        self.state[3] = float(fmt)
        self.fmt, self.ready, self.pending = fmt, False, False

    @property
    def inv(self):
        return self.state[2:3]


def fp8_begin(sc):
    """Before a kernel writes fp8 bytes in sc's scale: if the update that folds the previous use's amax into the scale has not been
    enqueued yet (no fp8_flush_updates since), do it for this stream now."""
    if sc.pending:
        _call("ffvc_fp8_update", sc.state.data_ptr(), sc.fmt, FP8_MARGIN, stream_ptr())
        sc.pending = False
        try:
            _F8_PENDING.remove(sc)
        except ValueError:
            pass


def fp8_flush_updates():
    """ONE launch per pool: every stream whose running amax is set gets its next scale.  Call where no fp8 tensor is between its producer
    and its consumer (the top of a step); streams nobody flushes are updated lazily by fp8_begin."""
    if not _F8_PENDING:
        return
    st = stream_ptr()
    for pools in _F8_POOLS.values():
        for buf, n in pools:
            if n:
                _call("ffvc_fp8_update_many", buf.data_ptr(), n, FP8_MARGIN, st)
    for sc in _F8_PENDING:
        sc.pending = False
    _F8_PENDING.clear()


def fp8_quant(x, sc, frozen=False):
    """x (16-bit or fp32, numel % 8 == 0) -> uint8 tensor of fp8 bytes in sc.fmt, scaled by sc's current scale.
    frozen: a weight, quantised once with its own amax (margin 1)."""
    _need_cuda(x)
    x = x if x.is_contiguous() else x.contiguous()
    n = x.numel()
    st = stream_ptr()
    if not sc.ready:
        _call("ffvc_fp8_amax", x.data_ptr(), dtype_code(x.dtype), sc.state.data_ptr(), n, st)
        _call("ffvc_fp8_update", sc.state.data_ptr(), sc.fmt, 1.0 if frozen else FP8_MARGIN, st)
        sc.ready = True
    fp8_begin(sc)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    _call("ffvc_fp8_quant", x.data_ptr(), dtype_code(x.dtype), out.data_ptr(), sc.fmt, sc.state.data_ptr(), n, st)
    if frozen:
        sc.state[1].zero_()     # a weight keeps the scale its bytes were written with: leave no amax behind for fp8_flush_updates to fold in
    return out


def fp8_next_scale(sc):
    """The GEMM that consumed this step's bytes has been enqueued: the amax accumulated while they were written may become the scale
    of the next step (enqueued by fp8_flush_updates for all streams at once, or by the stream's next producer)."""
    if not sc.pending:
        sc.pending = True
        _F8_PENDING.append(sc)


def conv_fp8_ok(B, H, W, Cin, Cout):
    """Geometry the fp8 3x3 convolution (conv_row_f8_kernel) covers: whole 256-pixel row tiles, 128-channel K steps, a full chip."""
    return ((W == 64 or W == 128 or (W >= 256 and W % 256 == 0)) and (H * W) % 256 == 0 and Cout % 128 == 0 and Cin % 128 == 0 and
            (B * H * W) % 256 == 0 and (B * H * W // 256) * (Cout // 128) >= 256 and os.environ.get("FFVC_CONV_FP8", "1") != "0")


def gemm_fp8(x8, w8, y, M, N, K, sx, sw, *, lo_dtype, bias=None, residual=None, aux=None, ldaux=0, act=ACT_NONE, flags=0,
             colsum=None, out_scale=None, conv=None, gn_sums=None):
    """y[M,N] = act(sx.inv * sw.inv * X8 W8^T + bias) (+ residual): x8 [M,K], w8 [N,K] uint8 fp8 bytes (x in sx.fmt).
    out_scale (an initialised Fp8Scale): y is a uint8 tensor that receives the result as fp8 bytes in out_scale.fmt, scaled by its
    current scale, and out_scale's running amax is updated — the operand of the next fp8 GEMM straight from the epilogue (the two
    MLP kinds of the frozen towers only, see include/ffvc.h y8_state)."""
    _need_cuda(x8, w8, y, bias, residual, aux)
    if x8.dtype != torch.uint8 or w8.dtype != torch.uint8:
        raise TypeError("gemm_fp8: operands must be uint8 tensors of fp8 bytes")
    d = GemmDesc()
    d.x, d.w, d.y = x8.data_ptr(), w8.data_ptr(), y.data_ptr()
    d.bias, d.residual, d.aux = _ptr(bias), _ptr(residual), _ptr(aux)
    d.M, d.N, d.K = M, N, K
    d.x_mode, d.w_mode = OP_KMAJOR, OP_KMAJOR
    if conv is not None:          # 3x3 convolution on an NHWC fp8 tensor: conv = (H, W, Cin) of the OUTPUT grid (F_UPSAMPLE2X in flags)
        d.x_mode = OP_CONV3X3
        d.conv_H, d.conv_W, d.conv_Cin = conv
    if gn_sums is not None:
        buf, hw, cpg = gn_sums
        d.gn_sums, d.gn_hw, d.gn_cpg = buf.data_ptr(), hw, cpg
        flags |= _lib.F_GN_SUMS
    if out_scale is not None:
        if y.dtype != torch.uint8 or not out_scale.ready:
            raise TypeError("gemm_fp8: fp8 output needs a uint8 y and an initialised Fp8Scale")
        fp8_begin(out_scale)
        d.y8_state, d.y8_fmt = out_scale.state.data_ptr(), out_scale.fmt
    elif y.dtype == torch.float32:
        flags |= F_OUT_F32
    elif y.dtype != lo_dtype:
        raise TypeError("gemm_fp8: y must be fp32 or lo_dtype")
    if residual is not None:
        if residual.dtype == torch.float32:
            flags |= F_RES_F32
        elif residual.dtype != lo_dtype:
            raise TypeError("gemm_fp8: residual must be fp32 or lo_dtype")
    if aux is not None and aux.dtype != lo_dtype:
        raise TypeError("gemm_fp8: aux must be lo_dtype")
    if colsum is not None:
        d.colsum = colsum.data_ptr()
        flags |= _lib.F_COLSUM
    d.act, d.flags, d.split_k, d.alpha = act, flags, 1, 1.0
    d.ldx, d.ldw, d.ldaux = K, K, ldaux
    d.y_mi, d.y_so, d.y_sm = 0, 0, N
    d.r_mi, d.r_so, d.r_sm = 0, 0, N
    d.batch, d.batch_inner = 1, 1
    lib = _lib.load()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    # ViT-L/14 at 64 cutouts: 16448 = 64 x 257 rows = 64 whole 256-row tiles + 64 rows.  With the tail inside the launch every tile
    # choice pays for a nearly empty extra round (260 tiles of 256 x 256 on 256 CUs); as two launches the 16384 rows fill whole rounds.
    # The 64 rows go to the skinny kernel (K split across the waves of a workgroup: ~6 us) when the epilogue is one it has — bias,
    # residual, plain store — and otherwise to the tiled kernel (13-26 us of K-loop latency: tools/fp8_rows_bench.py; then the split only
    # pays for narrow outputs and the 4096-wide one: N=1024 K=1024 42.4 -> 39.0 us, N=1024 K=4096 105.1 -> 98.9, N=4096 K=1024 106.5 ->
    # 103.7, but N=3072 K=1024 85.2 -> 95.2).  Every epilogue option is row-separable (column sums / amax accumulate).
    M0 = M - M % 256
    tail = M - M0
    simple = (act == ACT_NONE and aux is None and colsum is None and out_scale is None and gn_sums is None and
              (flags & ~(F_OUT_F32 | F_RES_F32)) == 0 and
              (residual is None or residual.dtype == torch.float32 or residual.dtype == y.dtype))
    skinny = bool(_FP8_SKINNY and conv is None and simple and 0 < tail <= 64 and lib.ffvc_gemm_fp8_skinny_ok(tail, N, K))
    if conv is None and _FP8_ROWSPLIT and M0 >= 8192 and 0 < tail <= 64 and (skinny or N <= 1024 or N >= 4096):
        segs = ((0, M0), (M0, tail))
    else:
        segs = ((0, M),)
        skinny = False
    base = (d.x, d.y, d.residual, d.aux)

    def issue():
        for r0, rows in segs:
            if skinny and r0:
                _lib.check(lib.ffvc_gemm_fp8_skinny(base[0] + r0 * K, d.w, base[1] + r0 * N * y.element_size(), dtype_code(y.dtype), _ptr(bias),
                                                    (base[2] + r0 * N * residual.element_size()) if residual is not None else None,
                                                    dtype_code(residual.dtype) if residual is not None else 0, rows, N, K, sx.fmt,
                                                    sx.inv.data_ptr(), sw.inv.data_ptr(), stream_ptr()), "ffvc_gemm_fp8_skinny")
                continue
            d.M = rows
            d.x = base[0] + r0 * K
            d.y = base[1] + r0 * N * y.element_size()
            if residual is not None:
                d.residual = base[2] + r0 * N * residual.element_size()
            if aux is not None:
                d.aux = base[3] + r0 * ldaux * aux.element_size()
            _lib.check(lib.ffvc_gemm_fp8(byref(d), sx.fmt, dtype_code(lo_dtype), sx.inv.data_ptr(), sw.inv.data_ptr(), stream_ptr()),
                       "ffvc_gemm_fp8")

    if REPLAY is not None:
        REPLAY.append(("conv3x3_fp8" if conv is not None else "gemm_nt_fp8", (M, N, K, 1, 1, int(flags), int(act)), issue,
                       (x8, w8, y, bias, residual, aux, colsum, sx, sw, out_scale)))
    issue()
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("conv3x3_fp8" if conv is not None else "gemm_nt_fp8", 2.0 * M * N * K, e0, e1, (M, N, K, 1, 1)))
    return y


def gemm_fp8_skinny(x8, w8, y, M, N, K, sx, sw, bias=None, residual=None):
    """y[M <= 64, N] = sx.inv * sw.inv * X8 W8^T (+ bias) (+ residual) on the K-split-across-waves kernel (see gemm_fp8's row split)."""
    _need_cuda(x8, w8, y, bias, residual)
    _call("ffvc_gemm_fp8_skinny", x8.data_ptr(), w8.data_ptr(), y.data_ptr(), dtype_code(y.dtype), _ptr(bias), _ptr(residual),
          dtype_code(residual.dtype) if residual is not None else 0, M, N, K, sx.fmt, sx.inv.data_ptr(), sw.inv.data_ptr(), stream_ptr())
    return y


def _gn_ws(B, HW, G, dev):
    n = _lib.load().ffvc_groupnorm_ws_bytes(B, HW, G)
    return torch.empty((n + 7) // 8, dtype=torch.float64, device=dev)


def gn_sums_ok(M, N, HW, dtype, G=32):
    """Can the GEMM that produces an (M = images*HW, N = C) NHWC tensor also accumulate its GroupNorm moments?"""
    return (dtype in LOWP and HW % 256 == 0 and M % HW == 0 and N % G == 0 and (N // G) % 4 == 0 and
            os.environ.get("FFVC_GN_FUSE", "1") != "0")


def colsum_fusable(dtype, N, K, *lds):
    """Can a K-major x K-major GEMM of this shape also accumulate the column sums of its output (FFVC_F_COLSUM)?  Needs the
    LDS-DMA kernels with the row-store epilogue: 16-bit operands, 8-element aligned rows everywhere."""
    return (dtype in LOWP and N % 8 == 0 and K % 8 == 0 and all(int(v) % 8 == 0 for v in lds) and
            os.environ.get("FFVC_COLSUM_FUSE", "1") != "0" and os.environ.get("FFVC_GEMM2_BM", "1") != "0" and
            os.environ.get("FFVC_EPI_ROWS", "1") != "0")


_SUMS_POOL = {}
_SUMS_POOL_ELEMS = int(os.environ.get("FFVC_SUMS_POOL", str(1 << 18)))      # fp64 elements per pool block (2 MiB); 0: one fill per buffer


def gn_sums_buffer(images, G, device):
    """Zeroed fp64 [images, G, 2] accumulator for GroupNorm moments / backward statistics.  A decoder pass needs ~60 of them per step
    (16 KiB each at batch 32): they are cut from a block that ONE fill zeroes, a fresh block when it is used up (the slices keep
    their block alive for as long as a backward pass needs them).  Under stream capture every buffer gets its own fill, so that a
    replay zeroes it again."""
    n = images * G * 2
    device = torch.device(device)
    if _SUMS_POOL_ELEMS <= 0 or n > _SUMS_POOL_ELEMS // 4 or torch.cuda.is_current_stream_capturing():
        return torch.zeros(images, G, 2, dtype=torch.float64, device=device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    ent = _SUMS_POOL.get(key)
    if ent is None or ent[1] + n > ent[0].numel():
        ent = [torch.zeros(_SUMS_POOL_ELEMS, dtype=torch.float64, device=device), 0]
        _SUMS_POOL[key] = ent
    o = ent[1]
    ent[1] = o + (n + 31) // 32 * 32                 # 256-byte aligned slices
    return ent[0][o:o + n].view(images, G, 2)


def groupnorm_fwd(x, gamma, beta, G=32, eps=1e-6, swish=True, sums=None, f8=None, f8_only=False):
    """x: NHWC (B, H, W, C) -> (y, mean[B,G], rstd[B,G]).  sums: moments already accumulated by the producer GEMM.
    f8 (an initialised Fp8Scale, 16-bit x): -> (y, mean, rstd, y8) where y8 holds the output as fp8 bytes in f8's scale (what fp8_quant(y, f8)
    would give); f8_only: y is NOT written (an uninitialised tensor is returned in its place — only for a caller that hands y8 to the
    one consumer)."""
    _req_f32(gamma, beta)
    _need_cuda(x)
    _need_cuda(x, gamma, beta)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    mean = torch.empty(B, G, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, G, dtype=torch.float32, device=x.device)
    if f8 is not None:
        if x.dtype not in LOWP or not f8.ready:
            raise TypeError("groupnorm_fwd: fp8 output needs a 16-bit tensor and an initialised Fp8Scale")
        if sums is not None and (sums.dtype != torch.float64 or tuple(sums.shape) != (B, G, 2)):
            raise TypeError("groupnorm_fwd: sums must be fp64 [B, G, 2]")
        fp8_begin(f8)
        y8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        ws = None if sums is not None else _gn_ws(B, HW, G, x.device)
        nb = x.numel() * x.element_size()
        with _hbm("groupnorm_fwd", (nb if f8_only else 2 * nb) + x.numel() + (0 if sums is not None else nb)):
            _call("ffvc_groupnorm_fwd_f8", x.data_ptr(), 0 if f8_only else y.data_ptr(), y8.data_ptr(), f8.state.data_ptr(), f8.fmt,
                  gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(ws), _ptr(sums), B, HW, C, G, eps, int(swish),
                  dtype_code(x.dtype), stream_ptr())
        return _unwritten(y, f8_only), mean, rstd, y8
    if sums is not None:
        if sums.dtype != torch.float64 or tuple(sums.shape) != (B, G, 2):
            raise TypeError("groupnorm_fwd: sums must be fp64 [B, G, 2]")
        with _hbm("groupnorm_fwd", 2 * x.numel() * x.element_size()):
            _call("ffvc_groupnorm_fwd_sums", x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                  rstd.data_ptr(), sums.data_ptr(), B, HW, C, G, eps, int(swish), dtype_code(x.dtype), stream_ptr())
        return y, mean, rstd
    ws = _gn_ws(B, HW, G, x.device)
    _call("ffvc_groupnorm_fwd", x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
          rstd.data_ptr(), ws.data_ptr(), B, HW, C, G, eps, int(swish), dtype_code(x.dtype), stream_ptr())
    return y, mean, rstd


_GNB_OK = {}
GNB_FUSE = os.environ.get("FFVC_GNB_FUSE", "1") != "0"      # A/B: GroupNorm-backward statistics inside the dgrad convolution


def conv_gnb_ok(dy, wd, dx, x_gn, mean, rstd, gamma, beta, B, H, W, Cout_of_conv, Cin_of_conv):
    """Will the dgrad convolution dx[B,H,W,Cin] = conv3x3(dy[B,H,W,Cout], wd) take the kernel that folds the GroupNorm-backward
    statistics of the node that produced the convolution's input (ffvc_gemm_gnb_probe)?  Asked once per shape."""
    if not GNB_FUSE or dy.dtype not in LOWP or x_gn.dtype != dy.dtype or Cin_of_conv % 32:
        return False
    key = (dy.dtype, B, H, W, Cout_of_conv, Cin_of_conv)
    ok = _GNB_OK.get(key)
    if ok is None:
        sums = gn_sums_buffer(B, 32, dy.device)
        ok = gemm(dy, wd, dx, B * H * W, Cin_of_conv, 9 * Cout_of_conv, ldw=9 * Cout_of_conv, x_mode=OP_CONV3X3, conv=(H, W, Cout_of_conv),
                  gnb=(x_gn, mean, rstd, gamma, beta, sums, True, H * W, Cin_of_conv // 32), probe_only=True)
        _GNB_OK[key] = ok
    return ok


def groupnorm_bwd(dy, x, gamma, beta, mean, rstd, dres=None, G=32, swish=True, f8=None, f8_only=False, sums=None):
    """f8 (an initialised Fp8Scale, 16-bit tensors): -> (dx, dx8), the gradient also as fp8 bytes in f8's scale (what fp8_quant(dx, f8)
    would give); f8_only: dx is not written (uninitialised)."""
    _req_f32(gamma, beta, mean, rstd)
    _need_cuda(dy, x, dres)
    _need_cuda(dy, x)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    dx = torch.empty_like(x)
    ws = _gn_ws(B, HW, G, x.device)
    if f8 is not None:
        if x.dtype not in LOWP or not f8.ready:
            raise TypeError("groupnorm_bwd: fp8 output needs 16-bit tensors and an initialised Fp8Scale")
        fp8_begin(f8)
        dx8 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        nb = x.numel() * x.element_size()
        with _hbm("groupnorm_bwd", nb * (4 + (0 if f8_only else 1) + (1 if dres is not None else 0)) + x.numel()):
            _call("ffvc_groupnorm_bwd_f8", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  _ptr(dres), 0 if f8_only else dx.data_ptr(), dx8.data_ptr(), f8.state.data_ptr(), f8.fmt, ws.data_ptr(), B, HW, C, G,
                  int(swish), dtype_code(x.dtype), stream_ptr())
        return _unwritten(dx, f8_only), dx8
    if sums is not None:
        # statistics already accumulated by the dgrad convolution that produced dy (gemm(gnb=...)): the apply pass only
        nb1 = x.numel() * x.element_size() * (3 + (1 if dres is not None else 0))
        with _hbm("groupnorm_bwd", nb1, min_bytes=nb1):
            _call("ffvc_groupnorm_bwd_sums", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  _ptr(dres), dx.data_ptr(), sums.data_ptr(), B, HW, C, G, int(swish), dtype_code(x.dtype), stream_ptr())
        return dx
    # algorithmic bytes of the two-pass backward: statistics (dy, x) + apply (dy, x, dres, dx)
    with _hbm("groupnorm_bwd", x.numel() * x.element_size() * (5 + (1 if dres is not None else 0)),
              min_bytes=x.numel() * x.element_size() * (3 + (1 if dres is not None else 0))):
        _call("ffvc_groupnorm_bwd", dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
              rstd.data_ptr(), _ptr(dres), dx.data_ptr(), ws.data_ptr(), B, HW, C, G, int(swish), dtype_code(x.dtype),
              stream_ptr())
    return dx


def softmax_fwd(s, p, rows, cols, lds, ldp, scale=1.0, causal=False, q_len=0):
    _req_f32(s)
    _need_cuda(p)
    _call("ffvc_softmax_fwd", s.data_ptr(), p.data_ptr(), dtype_code(p.dtype), rows, cols, lds, ldp, scale, int(causal),
          q_len, stream_ptr())
    return p


def softmax_bwd(p, dp, ds, rows, cols, ldp, lddp, scale=1.0):
    _req_f32(dp)
    _need_cuda(p, ds)
    _call("ffvc_softmax_bwd", p.data_ptr(), dp.data_ptr(), ds.data_ptr(), dtype_code(p.dtype), rows, cols, ldp, lddp,
          scale, stream_ptr())
    return ds


# ---------------------------------------------------------------------------
# glue
# ---------------------------------------------------------------------------
def cast(src, dtype):
    _need_cuda(src)
    if src.dtype == dtype:
        return src
    _need_cuda(src)
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    _call("ffvc_cast", src.data_ptr(), dtype_code(src.dtype), dst.data_ptr(), dtype_code(dtype), src.numel(),
          stream_ptr())
    return dst


def cast_into(src, dst):
    _need_cuda(src, dst)
    _call("ffvc_cast", src.data_ptr(), dtype_code(src.dtype), dst.data_ptr(), dtype_code(dst.dtype), src.numel(),
          stream_ptr())
    return dst


def transpose(src, out_dtype=None, out=None, pad_to=0):
    """src: (..., R, C) contiguous -> (..., C, max(R, pad_to)) (batched over leading dims; pad columns are zero)."""
    _need_cuda(src, out)
    _need_cuda(src)
    R, C = src.shape[-2], src.shape[-1]
    Rp = max(R, pad_to)
    batch = src.numel() // (R * C)
    if out is None:
        out = torch.empty(*src.shape[:-2], C, Rp, dtype=out_dtype or src.dtype, device=src.device)
    _call("ffvc_transpose", src.data_ptr(), dtype_code(src.dtype), out.data_ptr(), dtype_code(out.dtype), batch, R, C,
          R * C, C * Rp, Rp, stream_ptr())
    return out


class TransposePlan:
    """Device-side table for `ffvc_transpose_multi`: pairs of contiguous 2-D tensors (src [R, C] -> dst [C, R])."""

    def __init__(self, pairs):
        import numpy as np
        if not pairs:
            raise ValueError("TransposePlan: no matrices")
        dt = pairs[0][0].dtype
        items = np.zeros(len(pairs), dtype=[("src", "<u8"), ("dst", "<u8"), ("rows", "<i4"), ("cols", "<i4")])
        prefix = np.zeros(len(pairs), dtype=np.int32)
        total = 0
        for i, (src, dst) in enumerate(pairs):
            _need_cuda(src, dst)
            R, C = src.shape
            if src.dtype != dt or dst.dtype != dt or tuple(dst.shape) != (C, R) or not src.is_contiguous() or not dst.is_contiguous():
                raise TypeError("TransposePlan: need contiguous [R, C] -> [C, R] pairs of one dtype")
            items[i] = (src.data_ptr(), dst.data_ptr(), R, C)
            prefix[i] = total
            total += ((R + 63) // 64) * ((C + 63) // 64)
        dev = pairs[0][0].device
        self.items = torch.from_numpy(items.view(np.uint8).copy()).to(dev)
        self.prefix = torch.from_numpy(prefix).to(dev)
        self.n, self.total, self.dtype = len(pairs), total, dtype_code(dt)
        self._keep = pairs                                   # the table holds raw pointers: keep the tensors alive

    def run(self):
        _call("ffvc_transpose_multi", self.items.data_ptr(), self.prefix.data_ptr(), self.n, self.total, self.dtype,
              stream_ptr())


def copy2d(src, src_ld, rows, cols, dst_cols, out_dtype, dst_ld=None, out=None):
    """dst[r, c] = src[r, c] (c < cols) else 0, for c < dst_cols.  out: write into the first `rows` rows of an existing
    [>= rows, dst_ld] buffer instead of allocating one."""
    dst_ld = dst_ld or dst_cols
    if out is not None:
        _need_cuda(src, out)
        if out.dtype != out_dtype or not out.is_contiguous() or out.numel() < rows * dst_ld:
            raise TypeError("copy2d: `out` must be a contiguous buffer of the output dtype with >= rows * dst_ld elements")
    dst = out if out is not None else torch.empty(rows, dst_ld, dtype=out_dtype, device=src.device)
    _call("ffvc_copy2d", src.data_ptr(), dtype_code(src.dtype), src_ld, dst.data_ptr(), dtype_code(out_dtype), dst_ld, rows,
          cols, dst_cols, stream_ptr())
    return dst


def sln_fwd(hl, w, gamma, beta, gs, bs, out_dtype, eps=1e-5):
    dim = hl.shape[-1]
    rows = hl.numel() // dim
    y = torch.empty(hl.shape, dtype=out_dtype, device=hl.device)
    mean = torch.empty(rows, dtype=torch.float32, device=hl.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=hl.device)
    _call("ffvc_sln_fwd", hl.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), gs.data_ptr(), bs.data_ptr(),
          y.data_ptr(), dtype_code(out_dtype), mean.data_ptr(), rstd.data_ptr(), rows, dim, eps, stream_ptr())
    return y, mean, rstd


def sln_bwd(dy, hl, w, gamma, beta, gs, bs, mean, rstd, dres=None):
    """-> (dhl, dw, dgamma, dbeta, dgamma_s, dbeta_s)"""
    dim = hl.shape[-1]
    rows = hl.numel() // dim
    dev = hl.device
    nb = _lib.load().ffvc_layernorm_bwd_blocks(rows)
    dhl, dw = torch.empty_like(hl), torch.empty_like(w)
    pg = torch.empty(nb, dim, dtype=torch.float32, device=dev)
    pb = torch.empty(nb, dim, dtype=torch.float32, device=dev)
    ps = torch.empty(nb, 2, dtype=torch.float32, device=dev)
    _call("ffvc_sln_bwd", dy.data_ptr(), dtype_code(dy.dtype), hl.data_ptr(), w.data_ptr(), gamma.data_ptr(),
          beta.data_ptr(), gs.data_ptr(), bs.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dhl.data_ptr(),
          dw.data_ptr(), pg.data_ptr(), pb.data_ptr(), ps.data_ptr(), rows, dim, stream_ptr())
    dg = torch.empty(dim, dtype=torch.float32, device=dev)
    db = torch.empty(dim, dtype=torch.float32, device=dev)
    dsc = torch.empty(2, dtype=torch.float32, device=dev)
    colsum(pg, dg)
    colsum(pb, db)
    colsum(ps, dsc)
    return dhl, dw, dg, db, dsc[0:1], dsc[1:2]


def sln_bwd_acc(dy, hl, w, gamma, beta, gs, bs, mean, rstd, dgamma, dbeta, dscalars, dres=None):
    """-> (dhl, dw); dgamma / dbeta ([dim]) and dscalars ([2] = {dgamma_s, dbeta_s}) are accumulated in place."""
    _req_f32(hl, w, gamma, beta, gs, bs, mean, rstd, dgamma, dbeta, dscalars, dres)
    _need_cuda(dy)
    dim = hl.shape[-1]
    rows = hl.numel() // dim
    dhl, dw = torch.empty_like(hl), torch.empty_like(w)
    _call("ffvc_sln_bwd_acc", dy.data_ptr(), dtype_code(dy.dtype), hl.data_ptr(), w.data_ptr(), gamma.data_ptr(),
          beta.data_ptr(), gs.data_ptr(), bs.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dhl.data_ptr(),
          dw.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), dscalars.data_ptr(), rows, dim, stream_ptr())
    return dhl, dw


def sln_bwd_acc2(dy, hl, w, gamma, beta, gs, bs, mean, rstd, dgamma, dbeta, dgs, dbs, dres=None, dw=None):
    """-> (dhl, dw); dgamma / dbeta ([dim]) and the scalar gradients dgs / dbs ([1] each, anywhere) are accumulated in place; with
    `dw` given the gradient of the modulation input is added to it (and it is returned)."""
    _req_f32(hl, w, gamma, beta, gs, bs, mean, rstd, dgamma, dbeta, dgs, dbs, dres, dw)
    _need_cuda(dy)
    dim = hl.shape[-1]
    rows = hl.numel() // dim
    dhl = torch.empty_like(hl)
    acc = dw is not None
    if acc and (dw.shape != w.shape or not dw.is_contiguous()):
        raise ValueError("sln_bwd_acc2: dw must be a contiguous tensor of w's shape")
    if not acc:
        dw = torch.empty_like(w)
    _call("ffvc_sln_bwd_acc2", dy.data_ptr(), dtype_code(dy.dtype), hl.data_ptr(), w.data_ptr(), gamma.data_ptr(),
          beta.data_ptr(), gs.data_ptr(), bs.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _ptr(dres), dhl.data_ptr(),
          dw.data_ptr(), int(acc), dgamma.data_ptr(), dbeta.data_ptr(), dgs.data_ptr(), dbs.data_ptr(), rows, dim, stream_ptr())
    return dhl, dw


def split3(x, dtype=torch.float16, weight_order=False, out=None):
    """fp32 (..., K) -> 16-bit (..., 3K): [hi | lo | hi] (activations) or [hi | hi | lo] (weights), see ffvc_split3."""
    _req_f32(x)
    _need_cuda(x)
    Kd = x.shape[-1]
    rows = x.numel() // Kd
    if not x.is_contiguous():
        raise ValueError("split3: contiguous input expected")
    if out is None:
        out = torch.empty(*x.shape[:-1], 3 * Kd, dtype=dtype, device=x.device)
    with _aux("split3", 0.0, x.numel() * 4 + out.numel() * out.element_size()):
        _call("ffvc_split3", x.data_ptr(), out.data_ptr(), dtype_code(out.dtype), rows, Kd, Kd, int(weight_order), stream_ptr())
    return out


def colsum(x, out, accumulate=False, ld=None):
    _req_f32(out)
    _need_cuda(x)
    if out.numel() < x.shape[-1]:
        raise ValueError('colsum: out is shorter than the number of columns')
    rows, cols = x.shape[0], x.shape[-1]
    rows = x.numel() // cols
    with _aux("colsum", 0.0, rows * cols * x.element_size()):
        _call("ffvc_colsum", x.data_ptr(), dtype_code(x.dtype), out.data_ptr(), rows, cols, ld or cols, int(accumulate),
              stream_ptr())
    return out


def clamp_fwd(x, out_dtype, mul, add, lo, hi):
    _need_cuda(x)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _call("ffvc_clamp_fwd", x.data_ptr(), dtype_code(x.dtype), y.data_ptr(), dtype_code(out_dtype), x.numel(), mul, add,
          lo, hi, stream_ptr())
    return y


def clamp_bwd(x, g, mul, add, lo, hi):
    _need_cuda(x, g)
    dx = torch.empty_like(x)
    _call("ffvc_clamp_bwd", x.data_ptr(), dtype_code(x.dtype), g.data_ptr(), dtype_code(g.dtype), dx.data_ptr(),
          x.numel(), mul, add, lo, hi, stream_ptr())
    return dx


def sumpool2x2(src):
    _need_cuda(src)
    B, H2, W2, C = src.shape
    dst = torch.empty(B, H2 // 2, W2 // 2, C, dtype=src.dtype, device=src.device)
    _call("ffvc_sumpool2x2", src.data_ptr(), dst.data_ptr(), dtype_code(src.dtype), B, H2 // 2, W2 // 2, C, stream_ptr())
    return dst


def rownorm_sq(x):
    _req_f32(x)
    rows, dim = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty(rows, dtype=torch.float32, device=x.device)
    _call("ffvc_rownorm_sq", x.data_ptr(), out.data_ptr(), rows, dim, stream_ptr())
    return out


VQ_FUSE = os.environ.get("FFVC_VQ_FUSE", "1") != "0"       # A/B: the argmin inside the distance GEMM (FFVC_F_VQ_ARGMIN)


def vq_fused_ok(dtype, ncodes, depth):
    """Can the distance GEMM of `vector_quantize` keep the argmin itself?  16-bit K-major operands on the 256x256 LDS-DMA kernel."""
    return VQ_FUSE and dtype in LOWP and depth % 8 == 0 and ncodes % 4 == 0 and os.environ.get("FFVC_GEMM2_BM", "1") != "0"


def vq_argmin_fused(x, cb, xnorm, cnorm):
    """idx[r] = argmin_j (xnorm[r] + cnorm[j]) - 2 x[r] . cb[j]  (main.py:133-139; first minimum) without the [rows, codes] distance
    matrix: x [rows, K], cb [codes, K] 16-bit K-major (e.g. split3 operands).  Bit-identical to gemm(fp32 out) + vq_argmin."""
    rows, depth = x.shape
    ncodes = cb.shape[0]
    packed = torch.full((rows,), -1, dtype=torch.int64, device=x.device)
    gemm(x, cb, packed, rows, ncodes, depth, ldx=depth, ldw=depth, vq=(xnorm, cnorm, packed))
    return packed & 0xFFFFFFFF


def vq_argmin(dot, xnorm, cnorm):
    _req_f32(dot, xnorm, cnorm)
    rows, ncodes = dot.shape
    idx = torch.empty(rows, dtype=torch.int64, device=dot.device)
    with _aux("vq_argmin", 0.0, dot.numel() * 4):
        _call("ffvc_vq_argmin", dot.data_ptr(), xnorm.data_ptr(), cnorm.data_ptr(), idx.data_ptr(), rows, ncodes, ncodes,
              stream_ptr())
    return idx


def gather_rows(table, idx, out_dtype, pos=None, period=0):
    rows, dim = idx.numel(), table.shape[-1]
    out = torch.empty(*idx.shape, dim, dtype=out_dtype, device=table.device)
    _call("ffvc_gather_rows", table.data_ptr(), idx.data_ptr(), _ptr(pos), period, out.data_ptr(), dtype_code(out_dtype),
          rows, dim, stream_ptr())
    return out


def eot_gather(x, tokens):
    _need_cuda(x, tokens)
    B, L, D = x.shape
    out = torch.empty(B, D, dtype=torch.float32, device=x.device)
    _call("ffvc_eot_gather", x.data_ptr(), dtype_code(x.dtype), tokens.data_ptr(), out.data_ptr(), B, L, D, stream_ptr())
    return out


def cutouts_fwd(xr, cut, cutn, patch, mean, std, out_dtype, noise=None, facs=None):
    _req_f32(xr, noise, facs)
    B, H, W, _ = xr.shape
    g = cut // patch
    out = torch.empty(cutn * B, g * g, 3 * patch * patch, dtype=out_dtype, device=xr.device)
    _call("ffvc_cutouts_fwd", xr.data_ptr(), _ptr(noise), _ptr(facs), out.data_ptr(), dtype_code(out_dtype), B, H, W, cut,
          cutn, patch, mean[0], mean[1], mean[2], std[0], std[1], std[2], stream_ptr())
    return out


def cutouts_bwd(xr, gout, cut, cutn, patch, std):
    _req_f32(xr)
    _need_cuda(gout)
    B, H, W, _ = xr.shape
    dxr = torch.empty_like(xr)
    with _aux("cutouts_bwd", 0.0, gout.numel() * gout.element_size() + 2 * xr.numel() * 4):
        _call("ffvc_cutouts_bwd", xr.data_ptr(), gout.data_ptr(), dtype_code(gout.dtype), dxr.data_ptr(), B, H, W, cut, cutn,
              patch, std[0], std[1], std[2], stream_ptr())
    return dxr


def spherical_loss(embed, feats, coef=1.0, want_grad=True):
    N, D = embed.shape
    B = feats.shape[0]
    rowloss = torch.empty(N, dtype=torch.float32, device=embed.device)
    loss = torch.empty((), dtype=torch.float32, device=embed.device)
    dembed = torch.empty_like(embed) if want_grad else None
    _call("ffvc_spherical_loss", embed.data_ptr(), feats.data_ptr(), rowloss.data_ptr(), loss.data_ptr(), _ptr(dembed), N,
          B, D, coef, stream_ptr())
    return loss, dembed


def adam(p, g, m, v, shadow, lr, beta1, beta2, eps, step, grad_scale=1.0, ema=None, ema_weight=0.0, dev_scale=None,
         bad_count=None, dev_hyper=None):
    """ema (fp32, same layout as p): torch_ema update folded into the pass, ema -= ema_weight * (ema - p_new).
    dev_scale (fp32 device scalar): multiplied into grad_scale on the device (clip_grad_norm_ coefficient).
    bad_count (int32 device scalar): elements with a non-finite scaled gradient are skipped and counted (per wavefront)."""
    _req_f32(p, g, m, v, ema, dev_scale, dev_hyper)
    _need_cuda(shadow, bad_count)
    if bad_count is not None and bad_count.dtype != torch.int32:
        raise TypeError("adam: bad_count must be an int32 device scalar")
    with _hbm("adam", p.numel() * (28 + (shadow.element_size() if shadow is not None else 0) + (8 if ema is not None else 0))):
        _call("ffvc_adam", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(shadow),
              dtype_code(shadow.dtype) if shadow is not None else F32, p.numel(), lr, beta1, beta2, eps, step, grad_scale,
              _ptr(ema), float(ema_weight), _ptr(dev_scale), _ptr(bad_count), _ptr(dev_hyper), stream_ptr())


def clip_coef(sumsq_buf, max_norm, grad_scale):
    """-> fp32 [2] device tensor: (clip coefficient, total gradient norm)."""
    _req_f32(sumsq_buf)
    out = torch.empty(2, dtype=torch.float32, device=sumsq_buf.device)
    _call("ffvc_clip_coef", sumsq_buf.data_ptr(), float(max_norm), float(grad_scale), out.data_ptr(), stream_ptr())
    return out


def dropout(x, p, seed, residual=None, out=None, out_dtype=None):
    """out = (residual) + mask * x / (1 - p), mask regenerated from `seed` (same call on the gradient = backward)."""
    _need_cuda(x, out)
    _req_f32(residual)
    if not x.is_contiguous():
        raise TypeError("dropout: expected a contiguous tensor")
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    _call("ffvc_dropout", x.data_ptr(), dtype_code(x.dtype), _ptr(residual), out.data_ptr(), dtype_code(out.dtype), x.numel(),
          float(p), int(seed) & 0xFFFFFFFF, stream_ptr())
    return out


def mean_sq(x):
    _req_f32(x)
    out = torch.empty((), dtype=torch.float32, device=x.device)
    _call("ffvc_mean_sq", x.data_ptr(), out.data_ptr(), x.numel(), stream_ptr())
    return out


def mean_sq_bwd(x, g):
    _req_f32(x, g)
    dx = torch.empty_like(x)
    _call("ffvc_mean_sq_bwd", x.data_ptr(), g.data_ptr(), dx.data_ptr(), x.numel(), stream_ptr())
    return dx


def tv_loss_fwd(x):
    """x: NHWC fp32 (B, H, W, C) -> scalar tv_loss (main.py:423-428)."""
    _req_f32(x)
    B, H, W, C = x.shape
    out = torch.empty((), dtype=torch.float32, device=x.device)
    _call("ffvc_tv_loss_fwd", x.data_ptr(), out.data_ptr(), B, H, W, C, stream_ptr())
    return out


def tv_loss_bwd(x, g):
    _req_f32(x, g)
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    _call("ffvc_tv_loss_bwd", x.data_ptr(), g.data_ptr(), dx.data_ptr(), B, H, W, C, stream_ptr())
    return dx


def clock_sample():
    """-> int64 [8, 2] device tensor: per XCD (shader-clock ticks, 100 MHz ticks) sampled on the current stream."""
    out = torch.empty(8, 2, dtype=torch.int64, device="cuda")
    _call("ffvc_clock_sample", out.data_ptr(), stream_ptr())
    return out


def effective_clock_mhz(c0, c1):
    """Average engine clock between two clock_sample() results: mean over the XCDs present in both whose counters both advanced (the
    shader-clock counter of an XCD that was power-gated between the samples restarts: such an XCD says nothing about the region)."""
    a, b = c0.cpu(), c1.cpu()
    ok = (a[:, 1] > 0) & (b[:, 1] > a[:, 1]) & (b[:, 0] > a[:, 0])
    if not bool(ok.any()):
        return float("nan")
    d = (b - a)[ok].double()
    return float((d[:, 0] / d[:, 1]).mean() * 100.0)


def sumsq(x, out):
    _req_f32(x, out)
    _call("ffvc_sumsq", x.data_ptr(), out.data_ptr(), x.numel(), stream_ptr())
    return out


def axpby(x, y, a, b):
    _req_f32(x, y)
    _call("ffvc_axpby", x.data_ptr(), y.data_ptr(), x.numel(), a, b, stream_ptr())
    return y


def rowsum(x, out, period, accumulate=False):
    _req_f32(out)
    _need_cuda(x)
    if out.numel() < period:
        raise ValueError('rowsum: out is shorter than the period')
    cols = x.shape[-1]
    with _aux("rowsum", 0.0, x.numel() * x.element_size()):
        _call("ffvc_rowsum", x.data_ptr(), dtype_code(x.dtype), out.data_ptr(), x.numel() // cols, cols, period,
              int(accumulate), stream_ptr())
    return out


def copy_rows(src, src_stride, dst, dst_stride, rows, cols):
    _req(torch.float32, src, dst, contiguous=False)      # row-strided views by design (explicit strides)
    _call("ffvc_copy_rows", src.data_ptr(), src_stride, dst.data_ptr(), dst_stride, rows, cols, stream_ptr())
    return dst


def im2col3x3(x, out_dtype, Kp):
    B, H, W, C = x.shape
    out = torch.empty(B * H * W, Kp, dtype=out_dtype, device=x.device)
    _call("ffvc_im2col3x3", x.data_ptr(), dtype_code(x.dtype), out.data_ptr(), dtype_code(out_dtype), B, H, W, C, Kp,
          stream_ptr())
    return out


def mul_dev_scalar(x, s):
    _req_f32(x, s)
    y = torch.empty_like(x)
    _call("ffvc_mul_dev_scalar", x.data_ptr(), s.data_ptr(), y.data_ptr(), x.numel(), stream_ptr())
    return y


def gemm_splitk_accumulate(x, w, out, M, N, K, split_k, in_kernel=False, atomic=False, **kw):
    """out[M,N] (fp32, contiguous) += X W^T with the K range cut into `split_k` slices whose partial tiles go to
    slabs (plain stores) and are combined by one ffvc_slab_reduce pass — or, with in_kernel=True, are combined inside the
    launch by the last slice to arrive on each tile (FFVC_F_SPLITK_INKERNEL: no slabs, no reduce launch; the caller must know
    the shape takes a kernel that implements it)."""
    if split_k <= 1:
        return gemm(x, w, out, M, N, K, flags=kw.pop("flags", 0) | F_ACCUM_OUT, **kw)
    if in_kernel:
        return gemm(x, w, out, M, N, K, split_k=split_k, flags=kw.pop("flags", 0) | F_ACCUM_OUT | _lib.F_SPLITK_INKERNEL, **kw)
    if atomic:           # every K slice adds its partial tile straight into `out` with fp32 atomics: no slabs, no reduce pass
        return gemm(x, w, out, M, N, K, split_k=split_k, flags=kw.pop("flags", 0) | F_ATOMIC_OUT, **kw)
    slabs = torch.empty(split_k, M, N, dtype=torch.float32, device=out.device)
    gemm(x, w, slabs, M, N, K, split_k=split_k, slab_stride=M * N, **kw)
    with _aux("slab_reduce", 0.0, 4.0 * M * N * (split_k + 2)):
        _call("ffvc_slab_reduce", slabs.data_ptr(), out.data_ptr(), M * N, split_k, 1, stream_ptr())
    return out


def gemm_grouped_wgrad(dys, xs, out0, ystride, M, N, K, ldx, ldw):
    """ONE launch for the weight gradients of len(dys) layers of the same kind (include/ffvc.h grp_*):
        out_z[M, N] (fp32, at out0 + z * ystride elements) += dys[z][K, ldx>=M]^T @ xs[z][K, ldw>=N],   z = 0 .. len(dys) - 1,
    every 256x256 tile with its full reduction (no split-K, no slabs, no reduce pass).  dys / xs: lists of 16-bit [rows, ld]
    tensors (separate allocations; addressed through signed element offsets from the first one)."""
    G = len(dys)
    if not (2 <= G <= 8) or len(xs) != G:
        raise ValueError("gemm_grouped_wgrad: 2..8 operand pairs")
    _need_cuda(out0, *dys, *xs)
    d = GemmDesc()
    es = dys[0].element_size()
    d.x, d.w, d.y = dys[0].data_ptr(), xs[0].data_ptr(), out0.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.x_mode, d.w_mode = OP_TRANS, OP_TRANS
    d.in_dtype = dtype_code(dys[0].dtype)
    d.act, d.flags, d.split_k, d.alpha = ACT_NONE, F_OUT_F32 | F_ACCUM_OUT, 1, 1.0
    d.ldx, d.ldw = ldx, ldw
    d.y_mi, d.y_so, d.y_sm = 0, 0, N
    d.r_mi, d.r_so, d.r_sm = 0, 0, N
    d.batch, d.batch_inner = G, 1
    d.ybo = ystride
    d.grp_n = G
    for i in range(G):
        if dys[i].dtype != dys[0].dtype or xs[i].dtype != dys[0].dtype:
            raise TypeError("gemm_grouped_wgrad: operand dtypes differ")
        dx, dw = dys[i].data_ptr() - d.x, xs[i].data_ptr() - d.w
        if dx % es or dw % es:
            raise ValueError("gemm_grouped_wgrad: misaligned operand")
        d.grp_xoff[i], d.grp_woff[i] = dx // es, dw // es
    if REPLAY is not None:
        REPLAY.append(("gemm_tn" + {torch.float16: "_f16"}.get(dys[0].dtype, "_bf16"), (M, N, K, G, 1, int(d.flags), 0), d,
                       (tuple(dys), tuple(xs), out0)))
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.check(_lib.load().ffvc_gemm(byref(d), stream_ptr()), "ffvc_gemm (grouped)")
    if PROFILE is not None:
        e1.record()
        PROFILE.append(("gemm_tn" + {torch.float16: "_f16"}.get(dys[0].dtype, "_bf16"), 2.0 * M * N * K * G, e0, e1, (M, N, K, G, 1)))


class RcclComm:
    """The library's own RCCL communicator (csrc/comm.hip): `unique_id()` on rank 0, the 128 bytes travel to the other ranks
    through the host's side channel, `RcclComm(id, rank, world)` everywhere (collective), `allreduce(t, stream)` per bucket."""

    def __init__(self, unique_id, rank, world):
        if len(unique_id) != 128:
            raise ValueError("RcclComm: the unique id is 128 bytes")
        self._id = (ctypes.c_ubyte * 128).from_buffer_copy(bytes(unique_id))
        h = ctypes.c_void_p()
        _call("ffvc_rccl_comm_create", ctypes.cast(self._id, ctypes.c_void_p), int(rank), int(world), byref(h))
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def available():
        return int(_lib.load().ffvc_rccl_available())

    @staticmethod
    def unique_id():
        buf = (ctypes.c_ubyte * 128)()
        _call("ffvc_rccl_unique_id", ctypes.cast(buf, ctypes.c_void_p))
        return bytes(buf)

    def allreduce(self, t, stream=None):
        """In-place sum of the contiguous device tensor `t` over the ranks, enqueued on `stream` (default: current stream)."""
        _need_cuda(t)
        if not t.is_contiguous():
            raise TypeError("RcclComm.allreduce: contiguous tensor expected")
        sp = stream.cuda_stream if stream is not None else stream_ptr()
        _call("ffvc_allreduce_bucket", self._h, t.data_ptr(), t.numel(), dtype_code(t.dtype), sp)
        return t

    def destroy(self):
        if self._h:
            _call("ffvc_rccl_comm_destroy", self._h)
            self._h = ctypes.c_void_p()


def attn_small_ok(qkv, heads, causal):
    """Shapes the fused short-sequence attention kernel covers."""
    B, T, D3 = qkv.shape
    return (qkv.dtype in LOWP and not causal and T <= 64 and D3 == 3 * heads * 64 and
            os.environ.get("FFVC_ATTN_SMALL", "1") != "0")


def attn_small_fwd(qkv, heads, scale):
    _req(qkv.dtype if qkv.dtype in LOWP else torch.bfloat16, qkv)
    B, T, D3 = qkv.shape
    o = torch.empty(B, T, D3 // 3, dtype=qkv.dtype, device=qkv.device)
    with _aux("attn_small_fwd", 4.0 * B * heads * T * T * 64, (qkv.numel() + o.numel()) * qkv.element_size()):
        _call("ffvc_attn_small_fwd", qkv.data_ptr(), o.data_ptr(), dtype_code(qkv.dtype), B, T, heads, 64, float(scale),
              stream_ptr())
    return o


def attn_small_bwd(qkv, do, heads, scale):
    _req(qkv.dtype if qkv.dtype in LOWP else torch.bfloat16, qkv, do)
    if do.shape[:2] != qkv.shape[:2] or do.shape[2] * 3 != qkv.shape[2]:
        raise ValueError('attn_small_bwd: dout shape does not match qkv')
    B, T, D3 = qkv.shape
    dqkv = torch.empty_like(qkv)
    with _aux("attn_small_bwd", 10.0 * B * heads * T * T * 64, (2 * qkv.numel() + do.numel()) * qkv.element_size()):
        _call("ffvc_attn_small_bwd", qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dtype_code(qkv.dtype), B, T, heads, 64,
              float(scale),
              stream_ptr())
    return dqkv


def _tiny_strides(layout, T, heads, dh, row_ld):
    """Element strides (sample, token, which-of-q/k/v, head, channel) of a [B, T, row_ld] projection whose first 3*heads*dh
    columns hold q, k, v as '(d k h)' (vitgan.py:81-82) or '(k h d)'."""
    if layout == "dkh":
        return (T * row_ld, row_ld, heads, 1, 3 * heads)
    if layout == "khd":
        return (T * row_ld, row_ld, heads * dh, dh, 1)
    raise ValueError(f"attn_tiny: unknown layout {layout!r}")


def attn_tiny_ok(T, dh):
    """A handful of tokens x any head width: one workgroup per (sample, head), everything in LDS (csrc/attn_tiny.hip)."""
    return os.environ.get("FFVC_ATTN_TINY", "1") != "0" and bool(_lib.load().ffvc_attn_tiny_supported(int(T), int(dh)))


def attn_tiny_fwd(qkv, heads, dh, scale, layout="dkh", out_ld=None):
    """qkv [B, T, row_ld] (row_ld >= 3*heads*dh) -> out [B, T, out_ld] with heads*dh valid columns (pad zeroed)."""
    _need_cuda(qkv)
    if not qkv.is_contiguous():
        raise TypeError("attn_tiny_fwd: contiguous qkv expected")
    B, T, row_ld = qkv.shape
    if row_ld < 3 * heads * dh:
        raise ValueError("attn_tiny_fwd: rows shorter than 3 * heads * head_dim")
    out_ld = heads * dh if out_ld is None else int(out_ld)
    o = torch.empty(B, T, out_ld, dtype=qkv.dtype, device=qkv.device)
    sb, st, sk, sh, sd = _tiny_strides(layout, T, heads, dh, row_ld)
    _call("ffvc_attn_tiny_fwd", qkv.data_ptr(), o.data_ptr(), dtype_code(qkv.dtype), B, T, heads, dh, sb, st, sk, sh, sd, out_ld,
          float(scale), stream_ptr())
    return o


def attn_tiny_bwd(qkv, do, heads, dh, scale, layout="dkh"):
    """-> dqkv, same shape / layout as qkv (columns beyond 3*heads*dh zeroed)."""
    _need_cuda(qkv, do)
    if not (qkv.is_contiguous() and do.is_contiguous()) or do.dtype != qkv.dtype or do.shape[:2] != qkv.shape[:2]:
        raise TypeError("attn_tiny_bwd: contiguous qkv / dout of one dtype and matching [B, T] expected")
    B, T, row_ld = qkv.shape
    out_ld = do.shape[2]
    dqkv = torch.empty_like(qkv)
    sb, st, sk, sh, sd = _tiny_strides(layout, T, heads, dh, row_ld)
    _call("ffvc_attn_tiny_bwd", qkv.data_ptr(), do.data_ptr(), dqkv.data_ptr(), dtype_code(qkv.dtype), B, T, heads, dh, sb, st, sk,
          sh, sd, out_ld, 3 * heads * dh, float(scale), stream_ptr())
    return dqkv


def attn_flash_ok(qkv, heads):
    """Shapes the flash-style attention kernels cover: 16-bit storage, head dim 64, any length, causal or not."""
    B, T, D3 = qkv.shape
    return (qkv.dtype in LOWP and D3 == 3 * heads * 64 and B * heads <= 65535 and
            os.environ.get("FFVC_ATTN_FLASH", "1") != "0")


_ATTN_TEXT = os.environ.get("FFVC_ATTN_TEXT", "1") != "0"     # A/B: one-launch fp32 attention of the text tower


def attn_text_ok(qkv, heads):
    return (_ATTN_TEXT and qkv.dtype == torch.float32 and qkv.is_cuda and qkv.dim() == 3 and qkv.shape[1] <= 128 and
            qkv.shape[2] == 3 * heads * 64)


def attn_text_fwd(qkv, heads, scale, causal):
    """fp32 qkv [B, T <= 128, 3*heads*64] -> fp32 [B, T, heads*64]: exact-fp32 attention in one launch (forward only)."""
    _need_cuda(qkv)
    B, T, D3 = qkv.shape
    o = torch.empty(B, T, D3 // 3, dtype=torch.float32, device=qkv.device)
    with _aux("attn_text_fwd", 0.0, (qkv.numel() + o.numel()) * 4):      # exact-fp32 VALU attention: priced by its bytes
        _call("ffvc_attn_text_fwd", qkv.data_ptr(), o.data_ptr(), B, T, heads, 64, float(scale), int(bool(causal)), stream_ptr())
    return o


def attn_flash_fwd(qkv, heads, scale, causal, f8=None):
    """-> (out [B,T,heads*64], lse fp32 [B*heads,T]).  f8 (an initialised Fp8Scale): -> (out, lse, out8), the output also as the fp8
    bytes fp8_quant(out, f8) would give."""
    _req(qkv.dtype if qkv.dtype in LOWP else torch.bfloat16, qkv)
    B, T, D3 = qkv.shape
    o = torch.empty(B, T, D3 // 3, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B * heads, T, dtype=torch.float32, device=qkv.device)
    if f8 is not None:
        if not f8.ready:
            raise TypeError("attn_flash_fwd: fp8 output needs an initialised Fp8Scale")
        fp8_begin(f8)
        o8 = torch.empty(o.shape, dtype=torch.uint8, device=qkv.device)
        _call("ffvc_attn_flash_fwd_f8", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), o8.data_ptr(), f8.state.data_ptr(), f8.fmt,
              dtype_code(qkv.dtype), B, T, heads, 64, float(scale), int(bool(causal)), stream_ptr())
        return o, lse, o8
    with _aux("attn_flash_fwd", 4.0 * B * heads * T * T * 64 * (0.5 if causal else 1.0), (qkv.numel() + o.numel()) * qkv.element_size()):
        _call("ffvc_attn_flash_fwd", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), dtype_code(qkv.dtype), B, T, heads, 64,
              float(scale), int(bool(causal)), stream_ptr())
    return o, lse


def attn_flash_bwd(qkv, o, do, lse, heads, scale, causal):
    _req(qkv.dtype if qkv.dtype in LOWP else torch.bfloat16, qkv, o, do)
    _req_f32(lse)
    B, T, D3 = qkv.shape
    if tuple(do.shape) != (B, T, D3 // 3) or tuple(o.shape) != (B, T, D3 // 3) or tuple(lse.shape) != (B * heads, T):
        raise ValueError("attn_flash_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty_like(lse)
    with _aux("attn_flash_bwd", 10.0 * B * heads * T * T * 64 * (0.5 if causal else 1.0), (2 * qkv.numel() + 2 * o.numel()) * qkv.element_size()):
        _call("ffvc_attn_flash_bwd", qkv.data_ptr(), o.data_ptr(), do.data_ptr(), lse.data_ptr(), delta.data_ptr(),
              dqkv.data_ptr(), dtype_code(qkv.dtype), B, T, heads, 64, float(scale), int(bool(causal)), stream_ptr())
    return dqkv


def tokmix_supported(dtype, T, D, O):
    """Shapes / dtypes the fused token-mixing kernels cover (FFVC_TOKMIX=0 switches them off for A/B runs)."""
    if dtype not in LOWP or os.environ.get("FFVC_TOKMIX", "1") == "0":
        return False
    return bool(_lib.load().ffvc_tokmix_supported(dtype_code(dtype), T, D, O))


def tokmix_fwd(xn, w1, b1, w2, b2, residual):
    """y[b] = W2 @ gelu(W1 @ xn[b] + b1) + b2 + residual[b]; xn (B,T,D) 16-bit, residual / y fp32."""
    _req(xn.dtype, xn, w1, w2)
    _req_f32(b1, b2, residual)
    B, T, D = xn.shape
    O = w1.shape[0]
    if tuple(w1.shape) != (O, T) or tuple(w2.shape) != (T, O) or tuple(residual.shape) != (B, T, D):
        raise ValueError("tokmix_fwd: shape mismatch")
    y = torch.empty(B, T, D, dtype=torch.float32, device=xn.device)
    with _aux("tokmix_fwd", 4.0 * B * O * T * D, B * T * D * (xn.element_size() + 8)):
        _call("ffvc_tokmix_fwd", xn.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), residual.data_ptr(),
              y.data_ptr(), dtype_code(xn.dtype), B, T, D, O, stream_ptr())
    return y


def tokmix_fwd_save(xn, w1, b1, w2, b2, residual):
    """tokmix_fwd that also returns h = gelu(W1 xn + b1) and gact = gelu'(W1 xn + b1), (B,O,D) in xn's dtype: -> (y, h, gact)."""
    _req(xn.dtype, xn, w1, w2)
    _req_f32(b1, b2, residual)
    B, T, D = xn.shape
    O = w1.shape[0]
    if tuple(w1.shape) != (O, T) or tuple(w2.shape) != (T, O) or tuple(residual.shape) != (B, T, D):
        raise ValueError("tokmix_fwd_save: shape mismatch")
    y = torch.empty(B, T, D, dtype=torch.float32, device=xn.device)
    h = torch.empty(B, O, D, dtype=xn.dtype, device=xn.device)
    g = torch.empty_like(h)
    _call("ffvc_tokmix_fwd_save", xn.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), residual.data_ptr(),
          y.data_ptr(), h.data_ptr(), g.data_ptr(), dtype_code(xn.dtype), B, T, D, O, stream_ptr())
    return y, h, g


def tokmix_bwd_hidden(xn, dy, w1, b1, w2t, db1=None):
    """-> (h, dh) as (B,O,D) in xn's dtype: h = gelu(W1 xn + b1), dh = (W2^T dy) * gelu'(W1 xn + b1); w2t = W2^T (O,T).
    db1 (fp32 [O]): += sum_{b,d} dh, the first Conv1d's bias gradient, accumulated while dh is written."""
    _req(xn.dtype, xn, dy, w1, w2t)
    _req_f32(b1, db1)
    B, T, D = xn.shape
    O = w1.shape[0]
    if tuple(w1.shape) != (O, T) or tuple(w2t.shape) != (O, T) or tuple(dy.shape) != (B, T, D):
        raise ValueError("tokmix_bwd_hidden: shape mismatch")
    h = torch.empty(B, O, D, dtype=xn.dtype, device=xn.device)
    dh = torch.empty_like(h)
    if db1 is not None and (db1.numel() != O or not db1.is_contiguous()):
        raise ValueError("tokmix_bwd_hidden: db1 must be a contiguous [O] tensor")
    with _aux("tokmix_bwd_hidden", 4.0 * B * O * T * D, (2 * B * T * D + 2 * B * O * D) * xn.element_size()):
        _call("ffvc_tokmix_bwd_hidden", xn.data_ptr(), dy.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2t.data_ptr(), h.data_ptr(),
              dh.data_ptr(), _ptr(db1), dtype_code(xn.dtype), B, T, D, O, stream_ptr())
    return h, dh


def set_option(name, value):
    """Kernel-selection override (tests / A-B runs): see ffvc_set_option in include/ffvc.h."""
    _call("ffvc_set_option", name.encode(), int(value))


def augment_fwd(pooled, pinv, ainv, cmat, erase, cutn, patch, mean, std, out_dtype, noise=None, facs=None, coff=None,
                out_size=None, cj=None, seq=False):
    """pooled (B,3,Ss,Ss) fp32 -> patch rows of cutn*B cutouts of side out_size (default Ss).  cj (N,8): kornia ColorJitter
    parameters per cutout (augment.py), applied after cmat / coff.  seq: the affine slot is its own bilinear resample at the
    integer pixels of an intermediate image the homography slot then samples (kornia's two sequential warps, ffvc_augment_seq_fwd)."""
    _req_f32(pooled, pinv, ainv, cmat, noise, facs, coff, cj)
    _req(torch.int32, erase)
    B, _, Ss, _ = pooled.shape
    S = out_size or Ss
    g = S // patch
    out = torch.empty(cutn * B, g * g, 3 * patch * patch, dtype=out_dtype, device=pooled.device)
    if seq and S != Ss:
        raise ValueError("augment_fwd(seq=True): source and output side must be equal")
    with _aux("augment_fwd", 0.0, pooled.numel() * 4 + out.numel() * out.element_size() + (noise.numel() * 4 if noise is not None else 0)):
        _call("ffvc_augment_seq_fwd" if seq else "ffvc_augment_fwd", pooled.data_ptr(), pinv.data_ptr(), ainv.data_ptr(), cmat.data_ptr(), _ptr(coff), _ptr(cj),
              erase.data_ptr(), _ptr(noise), _ptr(facs), out.data_ptr(), dtype_code(out_dtype), B, S, Ss, cutn, patch, mean[0], mean[1],
              mean[2], std[0], std[1], std[2], stream_ptr())
    return out


def augment_bwd(gout, pinv, ainv, cmat, erase, B, S, cutn, patch, std, src_size=None, pooled=None, coff=None, cj=None, seq=False):
    _req_f32(pinv, ainv, cmat, pooled, coff, cj)
    _req(torch.int32, erase)
    _need_cuda(gout)
    Ss = src_size or S
    dpooled = torch.empty(B, 3, Ss, Ss, dtype=torch.float32, device=gout.device)
    with _aux("augment_bwd", 0.0, gout.numel() * gout.element_size() + dpooled.numel() * 4 + (pooled.numel() * 4 if pooled is not None else 0)):
        _call("ffvc_augment_seq_bwd" if seq else "ffvc_augment_bwd", gout.data_ptr(), dtype_code(gout.dtype), pinv.data_ptr(), ainv.data_ptr(), cmat.data_ptr(),
              erase.data_ptr(), _ptr(pooled), _ptr(coff), _ptr(cj), dpooled.data_ptr(), B, S, Ss, cutn, patch, std[0], std[1], std[2],
              stream_ptr())
    return dpooled


def sharpness_fwd(x, factor, on):
    """kornia sharpness on x (N,3,S,S) fp32 with per-sample factor / on flags (fp32 (N,))."""
    _req_f32(x, factor, on)
    N, _, S, _ = x.shape
    y = torch.empty_like(x)
    _call("ffvc_sharpness_fwd", x.data_ptr(), factor.data_ptr(), on.data_ptr(), y.data_ptr(), N, S, stream_ptr())
    return y


def sharpness_bwd(g, x, factor, on):
    _req_f32(g, x, factor, on)
    N, _, S, _ = x.shape
    dx = torch.empty_like(x)
    _call("ffvc_sharpness_bwd", g.data_ptr(), x.data_ptr(), factor.data_ptr(), on.data_ptr(), dx.data_ptr(), N, S, stream_ptr())
    return dx


def warp_grid_fwd(x, grid, on):
    """bilinear grid_sample(align_corners=False, zeros) of x (N,3,S,S) at the normalised coordinates grid (N,S,S,2)."""
    _req_f32(x, grid, on)
    N, _, S, _ = x.shape
    y = torch.empty_like(x)
    _call("ffvc_warp_grid_fwd", x.data_ptr(), grid.data_ptr(), on.data_ptr(), y.data_ptr(), N, S, stream_ptr())
    return y


def warp_grid_bwd(g, grid, on):
    _req_f32(g, grid, on)
    N, _, S, _ = g.shape
    dx = torch.empty_like(g)
    _call("ffvc_warp_grid_bwd", g.data_ptr(), grid.data_ptr(), on.data_ptr(), dx.data_ptr(), N, S, stream_ptr())
    return dx


def tps_grid(tps, S):
    """tps (N,26) fp32 (augment.tps_params) -> sampling grid (N,S,S,2)."""
    _req_f32(tps)
    N = tps.shape[0]
    grid = torch.empty(N, S, S, 2, dtype=torch.float32, device=tps.device)
    _call("ffvc_tps_grid", tps.data_ptr(), grid.data_ptr(), N, S, stream_ptr())
    return grid


def elastic_grid(noise, ksize=63, sigma=32.0, alpha=(1.0, 1.0)):
    """noise (N,2,S,S) fp32 in [-1,1] -> sampling grid (N,S,S,2) of kornia's elastic_transform2d."""
    _req_f32(noise)
    N, _, S, _ = noise.shape
    tmp, disp = torch.empty_like(noise), torch.empty_like(noise)
    grid = torch.empty(N, S, S, 2, dtype=torch.float32, device=noise.device)
    _call("ffvc_elastic_grid", noise.data_ptr(), tmp.data_ptr(), disp.data_ptr(), grid.data_ptr(), N, S, int(ksize), float(sigma),
          float(alpha[0]), float(alpha[1]), stream_ptr())
    return grid


def avgpool_patches_fwd(x, out_size, patch, mean, std, out_dtype):
    """x (N,3,S,S) fp32 -> adaptive average pool to out_size, mean/std, ViT patch rows (main.py:226-228 + :797)."""
    _req_f32(x)
    N, _, S, _ = x.shape
    g = out_size // patch
    out = torch.empty(N, g * g, 3 * patch * patch, dtype=out_dtype, device=x.device)
    _call("ffvc_avgpool_patches_fwd", x.data_ptr(), out.data_ptr(), dtype_code(out_dtype), N, S, out_size, patch, mean[0],
          mean[1], mean[2], std[0], std[1], std[2], stream_ptr())
    return out


def avgpool_patches_bwd(gout, N, S, out_size, patch, std):
    _need_cuda(gout)
    dx = torch.empty(N, 3, S, S, dtype=torch.float32, device=gout.device)
    _call("ffvc_avgpool_patches_bwd", gout.data_ptr(), dtype_code(gout.dtype), dx.data_ptr(), N, S, out_size, patch, std[0],
          std[1], std[2], stream_ptr())
    return dx
