"""checkpoint_io: reading pytorch-lightning VQGAN checkpoints and legacy pickled-module mapper checkpoints without the packages /
classes that wrote them (reference main.py:84-103, 568-575, 1273-1290)."""
import collections
import subprocess
import sys
import textwrap

import pytest
import torch

from feed_forward_vqgan_clip_amd import checkpoint_io as C

WRITER = textwrap.dedent('''
    import sys, types, collections, torch
    from torch import nn
    out = sys.argv[1]
    # --- a "pytorch_lightning"-like package and an "omegaconf"-like container that exist ONLY in this writer process
    pl = types.ModuleType("pytorch_lightning_fake"); cb = types.ModuleType("pytorch_lightning_fake.callbacks")
    sys.modules["pytorch_lightning_fake"] = pl; sys.modules["pytorch_lightning_fake.callbacks"] = cb
    class ModelCheckpoint:
        def __init__(self): self.best = 0.25; self.monitor = "val/loss"
    ModelCheckpoint.__module__ = "pytorch_lightning_fake.callbacks"; cb.ModelCheckpoint = ModelCheckpoint
    class AttrDict(dict): pass
    AttrDict.__module__ = "pytorch_lightning_fake"; pl.AttrDict = AttrDict
    g = torch.Generator().manual_seed(0)
    sd = collections.OrderedDict(("decoder.w%d" % i, torch.randn(3, 4, generator=g)) for i in range(3))
    torch.save({"state_dict": sd, "callbacks": {ModelCheckpoint: ModelCheckpoint()}, "hyper_parameters": AttrDict(lr=1e-3),
                "epoch": 7}, out + "/lightning.ckpt")
    # --- a legacy whole-module pickle of classes defined in a module that will not exist for the reader
    ref = types.ModuleType("reference_fake_models"); sys.modules["reference_fake_models"] = ref
    class Block(nn.Module):
        def __init__(self):
            super().__init__(); self.fc = nn.Linear(4, 4); self.act = nn.GELU(); self.register_buffer("scale", torch.ones(1))
            self.register_buffer("tmp", torch.zeros(2), persistent=False)
    class Net(nn.Module):
        def __init__(self):
            super().__init__(); self.inp = nn.Linear(3, 4); self.blocks = nn.ModuleList([Block(), Block()]); self.config = AttrDict(model_type="mlp_mixer", dim=4)
    for c in (Block, Net): c.__module__ = "reference_fake_models"; setattr(ref, c.__name__, c)
    torch.manual_seed(1); net = Net()
    torch.save(net, out + "/model.th"); torch.save(net.state_dict(), out + "/expected_sd.th")
''')


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("ckpt")
    subprocess.run([sys.executable, "-c", WRITER, str(d)], check=True)
    return d


def test_plain_torch_load_fails_and_tolerant_load_survives(files):
    with pytest.raises(Exception):
        torch.load(files / "lightning.ckpt", map_location="cpu", weights_only=False)
    ckpt, stubbed = C.tolerant_load(files / "lightning.ckpt", return_stubbed=True)
    assert "pytorch_lightning_fake.callbacks.ModelCheckpoint" in stubbed
    assert ckpt["epoch"] == 7 and list(ckpt["state_dict"]) == ["decoder.w0", "decoder.w1", "decoder.w2"]
    g = torch.Generator().manual_seed(0)
    for i in range(3):
        assert torch.equal(ckpt["state_dict"]["decoder.w%d" % i], torch.randn(3, 4, generator=g))
    # state of the stand-ins is kept, behaviour is not
    (cb,) = ckpt["callbacks"].values()
    assert cb.best == 0.25 and cb.monitor == "val/loss"


def test_legacy_pickled_module_flattens_to_its_state_dict(files):
    with pytest.raises(Exception):
        torch.load(files / "model.th", map_location="cpu", weights_only=False)
    net = C.tolerant_load(files / "model.th")
    assert C.is_module_like(net)
    sd = C.module_state_dict(net)
    want = torch.load(files / "expected_sd.th", map_location="cpu")
    assert list(sd) == list(want)                      # same names, same order, non-persistent buffer left out
    for k in want:
        assert torch.equal(sd[k], want[k])
    assert C.plain_config(net.config) == {"model_type": "mlp_mixer", "dim": 4}


def test_module_state_dict_matches_torch_on_a_real_module():
    m = torch.nn.Sequential(collections.OrderedDict(a=torch.nn.Linear(2, 3), b=torch.nn.BatchNorm1d(3)))
    sd = C.module_state_dict(m)
    assert list(sd) == list(m.state_dict())
    for k, v in m.state_dict().items():
        assert torch.equal(sd[k], v)


def test_tolerant_load_does_not_execute_importable_globals(tmp_path):
    """a pickle that names an importable callable (os.system, builtins.eval, subprocess.*) must not run it — every
    global outside the allowlist of tensor / container rebuilders becomes an inert stand-in."""
    import os
    import pickle

    from feed_forward_vqgan_clip_amd import checkpoint_io as cio
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    class Evil2:
        def __reduce__(self):
            return (eval, (f"open({str(marker)!r}, 'w').close()",))

    p = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {"w": torch.arange(6.0).view(2, 3)}, "a": Evil(), "b": Evil2()}, p, pickle_module=pickle)
    obj, stubbed = cio.tolerant_load(p, return_stubbed=True)
    assert not marker.exists()
    assert torch.equal(obj["state_dict"]["w"], torch.arange(6.0).view(2, 3))
    assert any(n.endswith(".system") for n in stubbed) and any(n.endswith(".eval") for n in stubbed)


def test_tolerant_load_does_not_execute_nested_pickles(tmp_path):
    """rebuilders that unpickle a byte string with the standard pickle module are a way around the allowlist —
    `torch.storage._load_from_bytes(<inner pickle>)` and numpy's `scalar(dtype('O'), <inner pickle>)`.  The first goes through the
    same unpickler (the inner os.system becomes a stand-in), the second is refused; a legitimate byte-serialised storage and a
    numpy scalar still load."""
    import io
    import os
    import pickle

    import numpy as np

    from feed_forward_vqgan_clip_amd import checkpoint_io as cio
    marker = tmp_path / "pwned"

    class Inner:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    buf = io.BytesIO()
    torch.save({"x": Inner()}, buf, pickle_module=pickle)

    class Nested:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (buf.getvalue(),))

    good = io.BytesIO()
    torch.save(torch.arange(4.0), good)

    class GoodNested:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (good.getvalue(),))

    p = tmp_path / "nested.ckpt"
    torch.save({"evil": Nested(), "good": GoodNested(), "s": np.float32(2.5), "w": torch.ones(2)}, p, pickle_module=pickle)
    obj, stubbed = cio.tolerant_load(p, return_stubbed=True)
    assert not marker.exists()
    assert any(n.endswith(".system") for n in stubbed)
    assert torch.equal(obj["good"], torch.arange(4.0)) and float(obj["s"]) == 2.5 and torch.equal(obj["w"], torch.ones(2))

    class ObjScalar:
        def __reduce__(self):
            return (np._core.multiarray.scalar, (np.dtype("O"), pickle.dumps(Inner())))

    q = tmp_path / "objscalar.ckpt"
    torch.save({"a": ObjScalar()}, q, pickle_module=pickle)
    with pytest.raises(Exception):
        cio.tolerant_load(q)
    assert not marker.exists()

