"""Thin Python launchers over the C ABI (one function per `ffvc_*` entry point).

Every launcher takes torch CUDA tensors (torch is only the allocator / stream owner),
passes raw device pointers + explicit dims to libffvc_hip.so and enqueues on torch's
current HIP stream.  Nothing here computes with torch ops.
"""
import ctypes
from ctypes import byref, c_int32, c_int64

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, BF16, F32, F_ATOMIC_OUT, F_BIAS_ALONG_M,  # noqa: F401
                   F_MUL_ACT_GRAD, F_OUT_F32, F_RES_F32, F_TR_SAFE, F_UPSAMPLE2X, F_WRITE_PREACT,
                   OP_CONV3X3, OP_KMAJOR, OP_TRANS, GemmDesc)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt):
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float32:
        return F32
    raise TypeError(f"unsupported compute dtype {dt}")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.FFVCError("ffvc kernels need CUDA(HIP) tensors; there is no CPU fallback")


def gemm(x, w, y, M, N, K, *, ldx=0, ldw=0, x_mode=OP_KMAJOR, w_mode=OP_KMAJOR, bias=None,
         residual=None, aux=None, ldaux=0, act=ACT_NONE, flags=0, split_k=1, alpha=1.0,
         kseg=0, xkso=0, wkso=0, y_map=None, r_map=None, batch=1, batch_inner=1,
         xb=(0, 0), wb=(0, 0), yb=(0, 0), rb=(0, 0), ab=(0, 0), conv=None):
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
    if y.dtype == torch.float32 and x.dtype != torch.float32:
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
    lib = _lib.load()
    _lib.check(lib.ffvc_gemm(byref(d), stream_ptr()), "ffvc_gemm")
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
