"""Host-side behaviour of the mapper classes that needs no GPU."""
import os
import sys

import pytest
import torch


def test_mixer_patch_size_other_than_one_is_refused_like_the_reference():
    """mlp_mixer_pytorch.py:88-89 views the [bs, (S/p)^2, C] output as [bs, S, S, C]: the reference's own forward raises for
    patch_size > 1, so the only runnable value is 1 (what main.py:479-488 passes).  Ours refuses at construction."""
    from feed_forward_vqgan_clip_amd.mappers import Mixer
    with pytest.raises(NotImplementedError, match="patch_size"):
        Mixer(input_dim=16, image_size=4, channels=8, patch_size=2, dim=32, depth=1)
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout not present (GPU box)")
    sys.path.insert(0, ref)
    try:
        from mlp_mixer_pytorch import Mixer as RefMixer
        m = RefMixer(input_dim=16, image_size=4, channels=8, patch_size=2, dim=32, depth=1)
        with pytest.raises(RuntimeError, match="invalid for input of size"):
            m(torch.randn(2, 16))
    finally:
        sys.path.remove(ref)
