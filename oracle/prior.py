"""Oracle (CPU fp32) restatement of the Net2Net prior's sampling direction.  TEST INFRASTRUCTURE ONLY.

main.py:1453-1462 builds net2net.modules.flow.flatflow.ConditionalFlatCouplingFlow and main.py:1037-1040 calls
`prior.sample(H)`.  net2net is not in /root/reference and not installed here -> PARITY UNPINNED: this file restates the
published module (ActNorm -> InvLeakyRelu(0.9) -> two conditional affine couplings with swapped halves -> Shuffle, per
flow; BasicFullyConnectedNet = Linear/LeakyReLU(0.01) stack, tanh on the scale heads) with plain torch.nn.functional
calls, independently of feed_forward_vqgan_clip_amd/prior.py, and is used only to check the HIP path against.
"""
import torch
import torch.nn.functional as F


def _fc(sd, prefix, x, tanh):
    idx = sorted(int(k.split(".")[-2]) for k in sd if k.startswith(prefix + ".main.") and k.endswith(".weight"))
    for n, i in enumerate(idx):
        x = F.linear(x, sd[f"{prefix}.main.{i}.weight"], sd[f"{prefix}.main.{i}.bias"])
        if n < len(idx) - 1:
            x = F.leaky_relu(x, 0.01)
    return torch.tanh(x) if tanh else x


def reverse(sd, z, cond, n_flows):
    """latent z (B, C) + conditioning (B, D) -> sample (B, C): the inverse of every block, last block first."""
    x = z.reshape(z.shape[0], -1).float()
    emb = _fc(sd, "embedder", cond.reshape(cond.shape[0], -1).float(), False)
    for i in reversed(range(n_flows)):
        p = f"sub_layers.{i}"
        x = x[:, sd[p + ".shuffle.backward_shuffle_idx"]]
        for j in reversed(range(2)):
            if j % 2 == 0:
                a, b = torch.chunk(x, 2, dim=1)
                x = torch.cat((b, a), dim=1)
            a, b = torch.chunk(x, 2, dim=1)
            ci = torch.cat((a, emb), dim=1)
            b = (b - _fc(sd, f"{p}.coupling.t.{j}", ci, False)) * _fc(sd, f"{p}.coupling.s.{j}", ci, True).neg().exp()
            x = torch.cat((a, b), dim=1)
        scaling = (x >= 0).float() + (x < 0).float() * 0.9
        x = x / scaling
        x = x / sd[p + ".norm_layer.scale"].reshape(1, -1) - sd[p + ".norm_layer.loc"].reshape(1, -1)
    return x
