"""Reading the checkpoints the reference reads, without the packages that wrote them.

* VQGAN checkpoints (reference main.py:84-103 -> taming's `init_from_ckpt`) are pytorch-lightning files: next to
  `state_dict` they pickle callbacks, OmegaConf containers, optimizer / scheduler objects of packages that are not installed
  here.  A plain `torch.load` dies on the first unknown class.  `tolerant_load` unpickles with a `find_class` that hands out
  inert stand-in classes for anything it cannot import, so the tensors come through untouched.
* Legacy mapper checkpoints (`model.th`, reference main.py:568-575, 1281-1289) are whole pickled `nn.Module` instances of the
  reference's own classes (`Mixer`, `Generator`, ...).  With the same stand-ins the object graph loads as a tree of stubs that
  still carries `_parameters / _buffers / _modules` and the `config` attribute; `module_state_dict` flattens it into the
  state_dict the module would have produced, which `main.load_model` then loads into a freshly built mapper (the reference's
  `_fix_*_gelu_issue` patches are unnecessary that way: no pickled activation objects survive).

Only state is recovered from stand-ins, never behaviour.  `find_class` is an ALLOWLIST: tensor / storage rebuilders of
torch, numpy's array rebuilders, `collections.OrderedDict`, the builtin containers and scalars resolve to the real objects; every
other global a pickle names — importable or not: `os.system`, `builtins.eval`, `subprocess.*`, but also `torch.nn.Linear` or an
optimizer class — becomes an inert stand-in whose "call" returns another stand-in, so a REDUCE opcode cannot run foreign code.
Rebuilders that would themselves unpickle a byte string with the STANDARD pickle module are not handed out as they are:
`torch.storage._load_from_bytes` (= `torch.load(BytesIO(b), weights_only=False)`) resolves to a shim that recurses through this
same unpickler, numpy's `scalar` / `_frombuffer` refuse object dtypes (`scalar(dtype('O'), bytes)` is `pickle.loads`), and
`_rebuild_wrapper_subclass` (takes a class argument) is a stand-in.
(That is a much smaller surface than plain `torch.load(weights_only=False)`; files should still come from a trusted source.)
"""
import io
import collections
import pickle

import torch


class _Stub:
    """Stand-in for an instance of a class that cannot be imported: keeps whatever state the pickle sets."""

    _ffvc_stub_of = "?"

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2:        # (dict state, slots state)
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        else:
            self.__dict__["_ffvc_state"] = state

    def __call__(self, *a, **k):                                 # objects rebuilt through a factory function: swallow it
        return _Stub()

    # dict / list subclasses are pickled as an empty instance + SETITEMS / APPENDS: keep those too
    def __setitem__(self, k, v):
        self.__dict__.setdefault("_ffvc_items", {})[k] = v

    def __getitem__(self, k):
        return self.__dict__.get("_ffvc_items", {})[k]

    def append(self, v):
        self.__dict__.setdefault("_ffvc_list", []).append(v)

    def extend(self, vs):
        self.__dict__.setdefault("_ffvc_list", []).extend(vs)

    def __getattr__(self, k):                                    # attribute-style access of dict-like containers (cfg.model_type)
        items = self.__dict__.get("_ffvc_items")
        if items is not None and k in items:
            return items[k]
        raise AttributeError(k)

    def __repr__(self):
        return f"<stub of {self._ffvc_stub_of}>"


_STUBS = {}


def _stub_class(module, name):
    key = f"{module}.{name}"
    if key not in _STUBS:
        _STUBS[key] = type(name, (_Stub,), {"_ffvc_stub_of": key, "__module__": "ffvc_stub." + module})
    return _STUBS[key]


_ALLOWED = {
    "collections": {"OrderedDict"},
    "builtins": {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice",
                 "range", "object"},
    "__builtin__": {"set", "frozenset", "list", "dict", "tuple", "int", "long", "float", "bool", "str", "unicode", "bytes", "complex",
                    "slice", "object"},
    "copyreg": {"_reconstructor"}, "copy_reg": {"_reconstructor"},
    "_codecs": {"encode"},
    "numpy": {"ndarray", "dtype"},
    "numpy.core.multiarray": {"_reconstruct"}, "numpy._core.multiarray": {"_reconstruct"},
    "torch": {"Size", "device", "dtype", "Tensor", "FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage",
              "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage", "TypedStorage"},
    "torch._utils": {"_rebuild_tensor", "_rebuild_tensor_v2", "_rebuild_tensor_v3", "_rebuild_parameter", "_rebuild_parameter_with_state",
                     "_rebuild_qtensor", "_rebuild_device_tensor_from_numpy"},
    "torch._tensor": {"_rebuild_from_type_v2", "Tensor"},
    "torch.nn.parameter": {"Parameter", "Buffer"},
    "torch.storage": {"UntypedStorage", "TypedStorage"},
    "torch.serialization": {"_get_layout"},
}


def _load_from_bytes(b):
    """Stand-in for torch.storage._load_from_bytes (legacy byte-serialised storages): the nested file goes through the SAME
    allowlisting unpickler instead of torch.load's default pickle module."""
    return torch.load(io.BytesIO(b), map_location="cpu", weights_only=False, pickle_module=_PickleModule)


def _no_object_dtype(dtype):
    import numpy as np
    dt = np.dtype(dtype)
    if dt.hasobject:
        raise pickle.UnpicklingError("checkpoint_io: numpy object dtypes are not rebuilt (their payload is a nested pickle)")
    return dt


def _np_scalar(dtype, obj=None):
    import numpy as np
    dt = _no_object_dtype(dtype)
    return np.frombuffer(obj, dtype=dt, count=1)[0] if obj is not None else dt.type()


def _np_frombuffer(buf, dtype, shape, order="C"):
    import numpy as np
    return np.frombuffer(buf, dtype=_no_object_dtype(dtype)).reshape(shape, order=order)


# names that resolve to a local shim instead of the object the pickle asked for
_SHIMS = {
    ("torch.storage", "_load_from_bytes"): _load_from_bytes,
    ("numpy.core.multiarray", "scalar"): _np_scalar, ("numpy._core.multiarray", "scalar"): _np_scalar,
    ("numpy.core.numeric", "_frombuffer"): _np_frombuffer, ("numpy._core.numeric", "_frombuffer"): _np_frombuffer,
}


class TolerantUnpickler(pickle.Unpickler):
    """pickle.Unpickler whose find_class never fails and never hands out foreign code: the allowlisted rebuilders / containers
    resolve to the real objects, every other global becomes an inert stand-in (names collected in `self.stubbed_names` and, for
    the duration of a tolerant_load, in the caller's set)."""

    _collect = None       # set by tolerant_load (thread-local would be needed for concurrent loads; tolerant_load takes a lock)

    def find_class(self, module, name):
        shim = _SHIMS.get((module, name))
        if shim is not None:
            return shim
        if name in _ALLOWED.get(module, ()):
            try:
                return super().find_class(module, name)           # (also applies pickle's Python-2 name mapping)
            except Exception:                                     # an allowlisted name this torch / numpy does not have
                pass
        if TolerantUnpickler._collect is not None:
            TolerantUnpickler._collect.add(f"{module}.{name}")
        return _stub_class(module, name)


class _PickleModule:
    """The duck-typed `pickle_module` torch.load accepts."""

    __name__ = "ffvc_tolerant_pickle"
    Unpickler = TolerantUnpickler
    load = staticmethod(lambda f, **kw: TolerantUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    PickleError = pickle.PickleError
    UnpicklingError = pickle.UnpicklingError


_LOAD_LOCK = __import__("threading").Lock()


def tolerant_load(path, return_stubbed=False):
    """torch.load(path, map_location='cpu') that survives classes of packages that are not installed (and does not execute what
    a pickle names outside the allowlist above)."""
    with _LOAD_LOCK:
        TolerantUnpickler._collect = set()
        try:
            obj = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleModule)
            names = sorted(TolerantUnpickler._collect)
        finally:
            TolerantUnpickler._collect = None
    return (obj, names) if return_stubbed else obj


def is_module_like(obj):
    return hasattr(obj, "_parameters") and hasattr(obj, "_modules")


def module_state_dict(mod, prefix=""):
    """state_dict of a pickled nn.Module (a real one or a tree of stand-ins): parameters, then persistent buffers, then the
    children in registration order — the order and names `nn.Module.state_dict` produces."""
    out = collections.OrderedDict()
    for k, v in (getattr(mod, "_parameters", None) or {}).items():
        if v is not None:
            out[prefix + k] = v.detach() if isinstance(v, torch.Tensor) else v
    skip = getattr(mod, "_non_persistent_buffers_set", None) or set()
    for k, v in (getattr(mod, "_buffers", None) or {}).items():
        if v is not None and k not in skip:
            out[prefix + k] = v
    for k, child in (getattr(mod, "_modules", None) or {}).items():
        if child is not None:
            out.update(module_state_dict(child, prefix + k + "."))
    return out


def plain_config(cfg):
    """Config objects come back as dicts, stub dicts, argparse-like stubs or OmegaConf stand-ins: reduce to a plain dict."""
    if isinstance(cfg, dict):
        return {k: plain_config(v) if isinstance(v, (dict, _Stub)) else v for k, v in cfg.items()}
    d = getattr(cfg, "__dict__", {})
    if "_ffvc_items" in d:                                       # a dict subclass read as a stand-in
        return {k: plain_config(v) if isinstance(v, (dict, _Stub)) else v for k, v in d["_ffvc_items"].items()}
    if "_content" in d:                                          # an OmegaConf DictConfig's storage
        content = d["_content"]
        if isinstance(content, dict):
            return {k: _omega_value(v) for k, v in content.items()}
    return {k: v for k, v in d.items() if not k.startswith("_")}


def _omega_value(node):
    d = getattr(node, "__dict__", {})
    if "_val" in d:
        return d["_val"]
    if "_content" in d:
        c = d["_content"]
        if isinstance(c, dict):
            return {k: _omega_value(v) for k, v in c.items()}
        if isinstance(c, (list, tuple)):
            return [_omega_value(v) for v in c]
    return node
