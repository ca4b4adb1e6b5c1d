"""Net2Net prior (text embedding -> image embedding) at inference: `load_prior_model` / `prior.sample`, the piece of
main.py:1447-1462 that `test(..., prior_path=...)` uses (main.py:1022-1023,1037-1040).

The reference builds `net2net.modules.flow.flatflow.ConditionalFlatCouplingFlow` [upstream: net2net is not in
/root/reference nor in this image -> PARITY UNPINNED; structure and state_dict names restated from the published module]:

    embedder   : BasicFullyConnectedNet(conditioning_dim -> embedding_dim, hidden 256, depth 2)
    sub_layers : n_flows x [ActNorm -> InvLeakyRelu(0.9) -> ConditionalDoubleVectorCouplingBlock -> Shuffle]
    coupling   : two (s, t) pairs of BasicFullyConnectedNet(in/2 + embedding_dim -> in/2, hidden_dim, hidden_depth),
                 s ends in tanh;  x_keep' = x_keep * exp(s) + t, halves swapped before the second pair
    BasicFullyConnectedNet = Linear, LeakyReLU(0.01), depth x [Linear, LeakyReLU], Linear [, Tanh]

Sampling runs the flow in reverse from z ~ N(0, I).  Every Linear is an exact-fp32 MFMA `ffvc_gemm` with the bias and the
LeakyReLU / tanh fused in the epilogue (FFVC_ACT_LRELU / FFVC_ACT_TANH); the per-element glue between them (halves,
exp, ActNorm affine, channel shuffle) is tiny ([B, in_channels]) and stays in torch on the device.  `train_prior`
(main.py:1335-1445) is not built.
"""
import torch

from . import kernels as K
from . import ops


def _cfg_get(cfg, *path):
    for k in path:
        cfg = cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)
    return cfg


class _FCNet:
    """BasicFullyConnectedNet: keys `<prefix>.main.<2i>.{weight,bias}`."""

    def __init__(self, sd, prefix, tanh):
        self.layers = []
        i = 0
        while f"{prefix}.main.{i}.weight" in sd:
            self.layers.append(ops.Weights.frozen(sd[f"{prefix}.main.{i}.weight"], sd[f"{prefix}.main.{i}.bias"],
                                                  torch.float32, need_dgrad=False))
            i += 2
        if len(self.layers) < 2:
            raise KeyError(f"prior: no Linear layers under '{prefix}.main'")
        self.tanh = tanh

    def __call__(self, x):
        x = x.contiguous()
        n = len(self.layers)
        for j, W in enumerate(self.layers):
            last = j == n - 1
            act = (K.ACT_TANH if self.tanh else K.ACT_NONE) if last else K.ACT_LRELU
            y = torch.empty(x.shape[0], W.N, dtype=torch.float32, device=x.device)
            K.gemm(x, W.sh, y, x.shape[0], W.N, W.K, ldx=W.K, ldw=W.K, bias=W.bias, act=act)
            x = y
        return x


class ConditionalFlatCouplingFlow:
    def __init__(self, state_dict, in_channels, conditioning_dim, n_flows):
        if not torch.cuda.is_available():
            raise RuntimeError("the prior needs a HIP device; there is no CPU fallback")
        sd = state_dict
        self.in_channels, self.conditioning_dim, self.n_flows = in_channels, conditioning_dim, n_flows
        self.embedder = _FCNet(sd, "embedder", False)
        self.blocks = []
        for i in range(n_flows):
            p = f"sub_layers.{i}"
            self.blocks.append(dict(
                loc=sd[p + ".norm_layer.loc"].reshape(1, -1).float().cuda(),
                scale=sd[p + ".norm_layer.scale"].reshape(1, -1).float().cuda(),
                s=[_FCNet(sd, f"{p}.coupling.s.{j}", True) for j in range(2)],
                t=[_FCNet(sd, f"{p}.coupling.t.{j}", False) for j in range(2)],
                back=sd[p + ".shuffle.backward_shuffle_idx"].long().cuda(),
                fwd=sd[p + ".shuffle.forward_shuffle_idx"].long().cuda()))

    @torch.no_grad()
    def reverse(self, z, cond):
        """z: (B, in_channels[,1,1]) latent, cond: (B, conditioning_dim[,1,1]) -> (B, in_channels, 1, 1)."""
        x = z.reshape(z.shape[0], -1).float().cuda()
        emb = self.embedder(cond.reshape(cond.shape[0], -1).float().cuda())
        half = self.in_channels // 2
        for blk in reversed(self.blocks):
            x = x[:, blk["back"]]                                              # Shuffle^-1
            for j in (1, 0):                                                   # coupling^-1, second pair first
                if j % 2 == 0:
                    x = torch.cat((x[:, half:], x[:, :half]), dim=1)
                xa, xk = x[:, :half], x[:, half:]
                ci = torch.cat((xa, emb), dim=1)
                xk = (xk - blk["t"][j](ci)) * torch.exp(-blk["s"][j](ci))
                x = torch.cat((xa, xk), dim=1)
            x = x / torch.where(x >= 0, torch.ones_like(x), torch.full_like(x, 0.9))   # InvLeakyRelu^-1
            x = x / blk["scale"] - blk["loc"]                                  # ActNorm^-1
        return x.view(x.shape[0], -1, 1, 1)

    @torch.no_grad()
    def forward(self, x, cond):
        """The normalising direction (x -> z, logdet), used to check reverse(forward(x)) == x."""
        x = x.reshape(x.shape[0], -1).float().cuda()
        emb = self.embedder(cond.reshape(cond.shape[0], -1).float().cuda())
        half = self.in_channels // 2
        logdet = torch.zeros(x.shape[0], device=x.device)
        for blk in self.blocks:
            x = blk["scale"] * (x + blk["loc"])
            logdet = logdet + torch.log(blk["scale"].abs()).sum()
            x = x * torch.where(x >= 0, torch.ones_like(x), torch.full_like(x, 0.9))
            for j in (0, 1):
                if j % 2 != 0:
                    x = torch.cat((x[:, half:], x[:, :half]), dim=1)
                xa, xk = x[:, :half], x[:, half:]
                ci = torch.cat((xa, emb), dim=1)
                s = blk["s"][j](ci)
                xk = xk * torch.exp(s) + blk["t"][j](ci)
                x = torch.cat((xa, xk), dim=1)
                logdet = logdet + s.sum(dim=1)
            x = x[:, blk["fwd"]]
        return x.view(x.shape[0], -1, 1, 1), logdet

    def sample(self, xc, generator=None):
        """main.py:1039: image-embedding samples for text embeddings xc (B, conditioning_dim, 1, 1)."""
        zz = torch.randn(xc.shape[0], self.in_channels, generator=generator)
        return self.reverse(zz, xc)

    def to(self, *_a, **_k):
        return self


def random_state_dict(in_channels, conditioning_dim, embedding_dim, hidden_dim, hidden_depth, n_flows, seed=0,
                      conditioning_hidden_dim=256, conditioning_depth=2):
    """Seeded weights in the module's key layout (no trained prior is shipped)."""
    g = torch.Generator().manual_seed(seed)

    def fc(prefix, dim, depth, hidden, out, sd):
        dims = [dim] + [hidden] * (depth + 1) + [out]
        for i in range(len(dims) - 1):
            bound = dims[i] ** -0.5
            sd[f"{prefix}.main.{2 * i}.weight"] = (torch.rand(dims[i + 1], dims[i], generator=g) * 2 - 1) * bound
            sd[f"{prefix}.main.{2 * i}.bias"] = (torch.rand(dims[i + 1], generator=g) * 2 - 1) * bound

    sd = {}
    fc("embedder", conditioning_dim, conditioning_depth, conditioning_hidden_dim, embedding_dim, sd)
    half = in_channels // 2
    for i in range(n_flows):
        p = f"sub_layers.{i}"
        sd[p + ".norm_layer.loc"] = torch.randn(1, in_channels, 1, 1, generator=g) * 0.1
        sd[p + ".norm_layer.scale"] = 1.0 + torch.randn(1, in_channels, 1, 1, generator=g) * 0.1
        sd[p + ".norm_layer.initialized"] = torch.tensor(1, dtype=torch.uint8)
        for j in range(2):
            fc(f"{p}.coupling.s.{j}", half + embedding_dim, hidden_depth, hidden_dim, half, sd)
            fc(f"{p}.coupling.t.{j}", half + embedding_dim, hidden_depth, hidden_dim, half, sd)
        perm = torch.randperm(in_channels, generator=g)
        sd[p + ".shuffle.forward_shuffle_idx"] = perm
        sd[p + ".shuffle.backward_shuffle_idx"] = torch.argsort(perm)
    return sd


def build_prior_model(config, input_size, output_size, state_dict):
    """main.py:1453-1462: in_channels = output_size (image embedding), conditioning_dim = input_size (text embedding)."""
    return ConditionalFlatCouplingFlow(state_dict, in_channels=output_size, conditioning_dim=input_size,
                                       n_flows=int(_cfg_get(config, "model", "n_flows")))


def load_prior_model(prior_path):
    """main.py:1447-1451: checkpoint {"model", "config", "input_size", "output_size", "step"} written by train_prior."""
    ckpt = torch.load(prior_path, map_location="cpu", weights_only=False)
    return build_prior_model(ckpt["config"], ckpt["input_size"], ckpt["output_size"], ckpt["model"])
