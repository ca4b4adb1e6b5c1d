"""Helpers to read tests/golden/*.npz (weights are stored as bf16 bit patterns in uint16)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def unpack_sd(npz, prefix):
    sd = {}
    for k in npz.files:
        if not k.startswith(prefix + "/"):
            continue
        a = npz[k]
        name = k[len(prefix) + 1:]
        if a.dtype == np.uint16:
            sd[name] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()
        else:
            sd[name] = torch.from_numpy(a.copy())
    return sd


def grads(npz, prefix):
    return {k[len(prefix) + 1:]: torch.from_numpy(npz[k].copy()) for k in npz.files if k.startswith(prefix + "/")}


def assert_close(a, b, rtol=1e-4, atol=1e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max abs err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"
