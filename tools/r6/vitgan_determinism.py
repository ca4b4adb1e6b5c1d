"""Round 6: run-to-run spread of the VitGAN generator's gradients (tests/test_models_gpu.py::_vitgan_grads) with and without grouped weight
gradients.  usage (GPU box): python tools/r6/vitgan_determinism.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_models_gpu import _vitgan_grads, _relrms  # noqa: E402

runs = {"g0a": _vitgan_grads(0)[0], "g0b": _vitgan_grads(0)[0], "g4a": _vitgan_grads(4)[0], "g4b": _vitgan_grads(4)[0]}
for a, b in [("g0a", "g0b"), ("g4a", "g4b"), ("g0a", "g4a")]:
    diffs = {k: _relrms(runs[a][k], runs[b][k]) for k in runs[a]}
    nz = {k: v for k, v in diffs.items() if v > 0}
    worst = sorted(nz.items(), key=lambda kv: -kv[1])[:6]
    print(f"{a} vs {b}: {len(nz)} of {len(diffs)} tensors differ; worst:", [(k, f"{v:.2e}") for k, v in worst], flush=True)
