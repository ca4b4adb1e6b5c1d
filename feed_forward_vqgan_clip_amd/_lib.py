"""ctypes binding of libffvc_hip.so (C ABI declared in include/ffvc.h).

The library is the product: if it is missing or a call fails we raise — there is
no CPU / eager fallback anywhere in this package (the oracle under oracle/ is test
infrastructure only and is never imported from here).
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int16, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FFVC_LIB") or os.path.join(_HERE, "lib", "libffvc_hip.so")   # FFVC_LIB: A/B builds of the same ABI

BF16, F32, F16 = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_QUICKGELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3, 4
OP_KMAJOR, OP_TRANS, OP_CONV3X3 = 0, 1, 2
F_BIAS_ALONG_M = 1
F_WRITE_PREACT = 2
F_MUL_ACT_GRAD = 4
F_ATOMIC_OUT = 8
F_RES_F32 = 16
F_OUT_F32 = 32
F_TR_SAFE = 64
F_UPSAMPLE2X = 128
F_ACCUM_OUT = 256
F_GN_SUMS = 512
F_COLSUM = 1024
F_SPLITK_INKERNEL = 4096
F_GNB_SUMS = 8192
F_VQ_ARGMIN = 16384
F_AUX_ACTGRAD = 2048


class GemmDesc(Structure):
    """Mirror of `ffvc_gemm_desc` (include/ffvc.h)."""

    _fields_ = [
        ("x", c_void_p),
        ("w", c_void_p),
        ("y", c_void_p),
        ("bias", c_void_p),
        ("residual", c_void_p),
        ("aux", c_void_p),
        ("M", c_int32),
        ("N", c_int32),
        ("K", c_int32),
        ("x_mode", c_int32),
        ("w_mode", c_int32),
        ("in_dtype", c_int32),
        ("act", c_int32),
        ("flags", c_int32),
        ("split_k", c_int32),
        ("alpha", c_float),
        ("ldx", c_int64),
        ("ldw", c_int64),
        ("ldaux", c_int64),
        ("kseg", c_int32),
        ("xkso", c_int64),
        ("wkso", c_int64),
        ("y_mi", c_int32),
        ("y_so", c_int64),
        ("y_sm", c_int64),
        ("r_mi", c_int32),
        ("r_so", c_int64),
        ("r_sm", c_int64),
        ("batch", c_int32),
        ("batch_inner", c_int32),
        ("xbo", c_int64),
        ("xbi", c_int64),
        ("wbo", c_int64),
        ("wbi", c_int64),
        ("ybo", c_int64),
        ("ybi", c_int64),
        ("rbo", c_int64),
        ("rbi", c_int64),
        ("abo", c_int64),
        ("abi", c_int64),
        ("conv_H", c_int32),
        ("conv_W", c_int32),
        ("conv_Cin", c_int32),
        ("x_mi", c_int32),
        ("x_so", c_int64),
        ("slab_stride", c_int64),
        ("gn_sums", c_void_p),
        ("gn_hw", c_int32),
        ("gn_cpg", c_int32),
        ("colsum", c_void_p),
        ("sk_ws", c_void_p),
        ("sk_cnt", c_void_p),
        ("sk_full", c_int32),
        ("sk_slices", c_int32),
        ("y8_state", c_void_p),
        ("y8_fmt", c_int32),
        ("grp_n", c_int32),
        ("grp_xoff", c_int64 * 8),
        ("grp_woff", c_int64 * 8),
        ("gnb_x", c_void_p),
        ("gnb_mean", c_void_p),
        ("gnb_rstd", c_void_p),
        ("gnb_gamma", c_void_p),
        ("gnb_beta", c_void_p),
        ("gnb_sums", c_void_p),
        ("gnb_swish", c_int32),
        ("vq_xn", c_void_p),
        ("vq_cn", c_void_p),
        ("vq_out", c_void_p),
    ]


_lib = None

# name -> (restype, argtypes).  tests/test_abi.py checks every symbol declared in
# include/ffvc.h appears here and is exported by the shared object.
_SIGNATURES = {
    "ffvc_gemm": (c_int, [POINTER(GemmDesc), c_void_p]),
    "ffvc_gemm_gnb_probe": (c_int, [POINTER(GemmDesc), c_void_p]),
    "ffvc_actgrad_inplace": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "ffvc_gemm_fp8": (c_int, [POINTER(GemmDesc), c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ffvc_fp8_quant": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "ffvc_fp8_amax": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p]),
    "ffvc_fp8_update": (c_int, [c_void_p, c_int, c_float, c_void_p]),
    "ffvc_gemm_skinny_ok": (c_int, [c_int, c_int, c_int]),
    "ffvc_gemm_skinny": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_gemm_fp8_skinny_ok": (c_int, [c_int, c_int, c_int]),
    "ffvc_gemm_fp8_skinny": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p]),
    "ffvc_fp8_update_many": (c_int, [c_void_p, c_int, c_float, c_void_p]),
    "ffvc_layernorm_fwd_f8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_int64, c_int, c_float, c_void_p]),
    "ffvc_layernorm_bwd_f8": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_int64, c_int, c_void_p]),
    "ffvc_attn_flash_fwd_f8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_int, c_void_p]),
    "ffvc_groupnorm_fwd_f8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "ffvc_groupnorm_bwd_f8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_layernorm_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                   c_int, c_float, c_void_p]),
    "ffvc_attn_text_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ffvc_attn_small_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ffvc_attn_small_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ffvc_rccl_available": (c_int, []),
    "ffvc_rccl_load": (c_int, [c_char_p]),
    "ffvc_rccl_unique_id": (c_int, [c_void_p]),
    "ffvc_rccl_comm_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "ffvc_allreduce_bucket": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ffvc_rccl_comm_destroy": (c_int, [c_void_p]),
    "ffvc_attn_tiny_supported": (c_int, [c_int, c_int]),
    "ffvc_attn_tiny_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int64,
                                   c_int64, c_int64, c_float, c_void_p]),
    "ffvc_attn_tiny_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64,
                                   c_int64, c_int64, c_int64, c_int64, c_float, c_void_p]),
    "ffvc_attn_flash_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ffvc_attn_flash_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_int, c_float, c_int, c_void_p]),
    "ffvc_transpose_multi": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ffvc_layernorm_bwd_blocks": (c_int, [c_int64]),
    "ffvc_layernorm_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ffvc_layernorm_bwd_acc": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ffvc_groupnorm_ws_bytes": (c_int64, [c_int, c_int, c_int]),
    "ffvc_groupnorm_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "ffvc_groupnorm_fwd_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_int, c_int, c_float, c_int, c_int, c_void_p]),
    "ffvc_groupnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_groupnorm_bwd_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_softmax_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_float, c_int, c_int,
                                 c_void_p]),
    "ffvc_softmax_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "ffvc_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_void_p]),
    "ffvc_split3": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int64, c_int, c_void_p]),
    "ffvc_transpose": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int, c_void_p]),
    "ffvc_copy2d": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int64, c_int, c_int, c_void_p]),
    "ffvc_sln_fwd": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "ffvc_sln_bwd": (c_int, [c_void_p, c_int] + [c_void_p] * 14 + [c_int64, c_int, c_void_p]),
    "ffvc_sln_bwd_acc": (c_int, [c_void_p, c_int] + [c_void_p] * 14 + [c_int64, c_int, c_void_p]),
    "ffvc_sln_bwd_acc2": (c_int, [c_void_p, c_int] + [c_void_p] * 11 + [c_int] + [c_void_p] * 4 + [c_int64, c_int, c_void_p]),
    "ffvc_colsum": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int64, c_int, c_void_p]),
    "ffvc_clamp_fwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int64, c_float, c_float, c_float, c_float, c_void_p]),
    "ffvc_clamp_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                               c_void_p]),
    "ffvc_sumpool2x2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_rownorm_sq": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "ffvc_vq_argmin": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "ffvc_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "ffvc_eot_gather": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ffvc_cutouts_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_float, c_float, c_float, c_float, c_float, c_float, c_void_p]),
    "ffvc_cutouts_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                 c_float, c_float, c_void_p]),
    "ffvc_augment_fwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_int, c_int] + [c_float] * 6 + [c_void_p]),
    "ffvc_augment_seq_fwd": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_int, c_int] + [c_float] * 6 + [c_void_p]),
    "ffvc_augment_seq_bwd": (c_int, [c_void_p, c_int] + [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                     c_float, c_void_p]),
    "ffvc_avgpool_patches_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_float] * 6 + [c_void_p]),
    "ffvc_avgpool_patches_bwd": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int] + [c_float] * 3 + [c_void_p]),
    "ffvc_augment_bwd": (c_int, [c_void_p, c_int] + [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                                 c_float, c_void_p]),
    "ffvc_sharpness_fwd": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "ffvc_sharpness_bwd": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "ffvc_warp_grid_fwd": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "ffvc_warp_grid_bwd": (c_int, [c_void_p] * 4 + [c_int, c_int, c_void_p]),
    "ffvc_tps_grid": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ffvc_elastic_grid": (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "ffvc_spherical_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                    c_void_p]),
    "ffvc_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_float, c_float, c_float,
                          c_float, c_int, c_float, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ffvc_clock_sample": (c_int, [c_void_p, c_void_p]),
    "ffvc_clip_coef": (c_int, [c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "ffvc_dropout": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_float, ctypes.c_uint32, c_void_p]),
    "ffvc_mean_sq": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ffvc_mean_sq_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ffvc_tv_loss_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_tv_loss_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_rowsum": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "ffvc_copy_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "ffvc_im2col3x3": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_mul_dev_scalar": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "ffvc_slab_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "ffvc_sumsq": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "ffvc_axpby": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    "ffvc_tokmix_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "ffvc_tokmix_fwd": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_tokmix_bwd_hidden": (c_int, [c_void_p] * 8 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_tokmix_fwd_save": (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ffvc_set_option": (c_int, [c_char_p, c_int]),
    "ffvc_last_error": (c_char_p, []),
    "ffvc_version": (c_int, []),
    "ffvc_device_info": (c_int, [POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "ffvc_probe_tr16": (c_int, [c_void_p, c_void_p]),
}


class FFVCError(RuntimeError):
    pass


def load():
    """Load the shared object (once). Raises FFVCError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FFVCError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C feed_forward_vqgan_clip_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback."
        )
    # torch ships its own libamdhip64.so.7; it has to be resident BEFORE this library is mapped so both share ONE HIP
    # runtime (same SONAME).  Loaded the other way round the process holds two runtimes and launches on torch's
    # device pointers fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ffvc_last_error()
        raise FFVCError(f"{what} failed (status {status}): {msg.decode() if msg else '?'}")


def declared_symbols():
    return sorted(_SIGNATURES)
