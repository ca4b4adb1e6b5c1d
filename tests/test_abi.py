"""The C-ABI library loads and exports every symbol include/ffvc.h declares (no compute calls: no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ffvc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffvc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from feed_forward_vqgan_clip_amd import _lib
    assert _declared() == _lib.declared_symbols()


def test_library_exports_every_declared_symbol():
    from feed_forward_vqgan_clip_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libffvc_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    assert _lib.load().ffvc_version() >= 100


def test_gemm_desc_layout_matches_header():
    """ctypes mirror of ffvc_gemm_desc has the header's field order."""
    from feed_forward_vqgan_clip_amd._lib import GemmDesc
    text = open(os.path.join(ROOT, "include", "ffvc.h")).read()
    body = text[text.index("typedef struct ffvc_gemm_desc {"):text.index("} ffvc_gemm_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for line in body.splitlines()[1:]:
        line = line.strip().rstrip(";")
        if not line:
            continue
        decl = re.sub(r"^(const\s+)?(void|float|double|int32_t|int64_t|uint32_t|uint64_t)\s*\*?\s*", "", line)
        names += [re.sub(r"\[\d+\]$", "", n.strip().lstrip("*")) for n in decl.split(",")]      # (fixed-size arrays: name[8])
    assert names == [f[0] for f in GemmDesc._fields_]


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "feed_forward_vqgan_clip_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from feed_forward_vqgan_clip_amd import kernels as K
    from feed_forward_vqgan_clip_amd._lib import FFVCError
    x = torch.zeros(8, 8)
    with pytest.raises((FFVCError, RuntimeError)):
        K.gemm(x, x, x, 8, 8, 8, ldx=8, ldw=8)
    from feed_forward_vqgan_clip_amd.mappers import Mixer
    with pytest.raises(RuntimeError):
        Mixer(input_dim=8, image_size=2, channels=8, patch_size=1, dim=8, depth=1).prepare()
