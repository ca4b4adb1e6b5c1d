"""Per-stage precision budget of the train-step forward at the benchmark's own model sizes (GPU tool).

The fp32-MFMA mode of the HIP path is the reference here (it matches the CPU oracle to ~2e-7 on the loss,
bench.py `parity_full_size`).  Each stage is then run in the low-precision mode while every other stage stays fp32,
so the loss deviation of the timed mode can be attributed:

    mapper   -> z rel-rms, VQ index agreement, loss deviation with only the mapper in low precision
    decoder  -> xr rel-rms (same codes), loss deviation with only the decoder in low precision
    clip     -> embed rel-rms, loss deviation with only the image tower in low precision
    all      -> the timed mode

    python tools/error_budget.py --batch 4 [--depth 32 --dim 1024] [--seeds 3] [--lp bf16,f16]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from feed_forward_vqgan_clip_amd import augment as faug  # noqa: E402
from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import ops  # noqa: E402
from feed_forward_vqgan_clip_amd import vqgan as fvq  # noqa: E402

DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def relrms(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


class Pipe:
    """The forward half of TrainStep cut into stages (main.py:729-811)."""

    def __init__(self, cfg, mixer_sd, vq_sd, clip_sd, cdt, vq_cfg, clip_dt=None, vq_dt=None):
        self.net = fmain.build_model(cfg, vq_cfg["z_channels"])
        self.net.load_state_dict(mixer_sd)
        self.net = self.net.cuda().prepare(cdt)
        self.vq = fvq.VQGAN(vq_sd, vq_cfg, vq_dt or cdt)
        self.clip = fclip.CLIP(clip_sd, clip_dt or cdt)
        self.mc = fmain.MakeCutouts(cfg.clip_size, cfg.cutn, augs=cfg.get("augs"))

    @torch.no_grad()
    def mapper(self, feats):
        return self.net(feats).permute(0, 2, 3, 1).contiguous()                 # NHWC fp32

    @torch.no_grad()
    def decode(self, z_nhwc):
        z = ops.clamp_with_grad(z_nhwc, self.vq.z_min, self.vq.z_max)
        return fvq.synth_nhwc(self.vq, z)                                       # (xr NHWC fp32, idx)

    @torch.no_grad()
    def embed(self, xr, facs, noise, prm):
        p = self.mc.patches(xr, self.clip.patch, tuple(fmain.CLIP_MEAN), tuple(fmain.CLIP_STD), self.clip.cdt, facs, noise, prm)
        return self.clip.encode_patches(p)


def mixer_emulated(sd, feats, depth, S, C, dt):
    """torch restatement of mappers.Mixer.forward's ROUNDING POINTS for a storage dtype `dt` (GEMM operands and stored
    activations rounded to dt, fp32 accumulate, fp32 residual stream / LayerNorm statistics) — lets a candidate storage
    format (f16) be judged before any kernel exists.  Tool only; never on the product path."""
    F = torch.nn.functional
    r = lambda t: t.to(dt).float()  # noqa: E731
    lin = lambda x, w, b: r(x) @ r(w).t() + b  # noqa: E731
    B = feats.shape[0]
    h = r(lin(feats, sd["proj.weight"], sd["proj.bias"])).view(B, C, S * S).transpose(1, 2)
    h = lin(h, sd["mixer.1.weight"], sd["mixer.1.bias"])
    for i in range(2, depth + 2):
        p = f"mixer.{i}."
        hn = r(F.layer_norm(h, h.shape[-1:], sd[p + "0.norm.weight"], sd[p + "0.norm.bias"]))
        w1, w2 = sd[p + "0.fn.0.weight"][:, :, 0], sd[p + "0.fn.3.weight"][:, :, 0]
        t = r(w1) @ hn + sd[p + "0.fn.0.bias"][:, None]
        h = h + (r(w2) @ r(F.gelu(t)) + sd[p + "0.fn.3.bias"][:, None])
        hn = r(F.layer_norm(h, h.shape[-1:], sd[p + "1.norm.weight"], sd[p + "1.norm.bias"]))
        t = lin(hn, sd[p + "1.fn.0.weight"], sd[p + "1.fn.0.bias"])
        h = h + lin(F.gelu(t), sd[p + "1.fn.3.weight"], sd[p + "1.fn.3.bias"])
    p = f"mixer.{depth + 2}."
    hn = F.layer_norm(h, h.shape[-1:], sd[p + "weight"], sd[p + "bias"])
    return lin(hn, sd["final_proj.weight"], sd["final_proj.bias"]).view(B, S, S, C)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emulate", default="", help="comma list of torch dtypes (f16,bf16,fp32) for the mapper emulation")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--cutn", type=int, default=8)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--lp", default="bf16")
    ap.add_argument("--augs", default="default")
    args = ap.parse_args()
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=args.dim, depth=args.depth, dropout=0, cutn=args.cutn,
                       batch_size=args.batch, repeat=1, nb_noise=None, diversity_coef=0, clip_model="ViT-B/32",
                       clip_size=224, model_type="mlp_mixer", vq_image_size=16,
                       augs=None if args.augs == "default" else args.augs.split(","))
    torch.manual_seed(1234)
    mixer_sd = {k: v.detach().clone() for k, v in fmain.build_model(cfg, 256).state_dict().items()}
    vq_sd = fvq.random_state_dict(fvq.F16_16384, seed=1234)
    clip_sd = fclip.random_state_dict(fclip.VIT_B32, seed=1234)
    ref = Pipe(cfg, mixer_sd, vq_sd, clip_sd, torch.float32, fvq.F16_16384)
    out = {"batch": args.batch, "cutn": args.cutn, "model": f"mixer {args.depth}x{args.dim}", "modes": {}}
    for lp in args.lp.split(","):
        low = Pipe(cfg, mixer_sd, vq_sd, clip_sd, DT[lp], fvq.F16_16384)
        rows = []
        for s in range(args.seeds):
            B, n = args.batch, args.cutn * args.batch
            tok = fmain.synthetic_tokens(B, seed=99 + s).cuda()
            g = torch.Generator().manual_seed(5 + s)
            facs = (torch.rand(n, generator=g) * 0.1).cuda()
            noise = torch.randn(n, 3, 224, 224, generator=g).cuda()
            prm = None
            if ref.mc.augs:
                prm = {k: v.cuda() for k, v in faug.draw_params(n, 224, ref.mc.augs, generator=g).items()}
            feats = ref.clip.encode_text(tok).float()
            z0 = ref.mapper(feats)
            xr0, idx0 = ref.decode(z0)
            e0 = ref.embed(xr0, facs, noise, prm)
            l0 = float(ops.spherical_loss(e0, feats, 1.0))

            def loss_from_z(z, dec, emb):
                xr, idx = dec.decode(z)
                return float(ops.spherical_loss(emb.embed(xr, facs, noise, prm), feats, 1.0)), xr, idx

            z1 = low.mapper(feats)
            la, _, idx1 = loss_from_z(z1, ref, ref)                              # mapper low, rest fp32
            lb, xr1, _ = loss_from_z(z0, low, ref)                               # decoder low
            e1 = low.embed(xr0, facs, noise, prm)                                # clip low
            lc = float(ops.spherical_loss(e1, feats, 1.0))
            ld, _, _ = loss_from_z(z1, low, low)                                 # everything low (the timed mode)
            lbc, _, _ = loss_from_z(z0, low, low)                                # decoder + clip low, mapper fp32
            emu = {}
            for name in [e for e in args.emulate.split(",") if e]:
                torch.backends.cuda.matmul.allow_tf32 = False
                sdc = {k: v.cuda() for k, v in mixer_sd.items()}
                ze = mixer_emulated(sdc, feats, args.depth, 16, 256, DT[name])
                le, _, idxe = loss_from_z(ze.contiguous(), ref, ref)
                emu[name] = {"z_relrms": relrms(ze, z0), "vq_flips": int((idxe != idx0).sum()), "rel_mapper_only": abs(le - l0) / l0}
            rows.append({
                "emulated_mapper": emu,
                "loss_fp32": l0,
                "z_relrms": relrms(z1, z0), "vq_agree": float((idx1 == idx0).float().mean()),
                "vq_flips": int((idx1 != idx0).sum()), "vq_positions": idx0.numel(),
                "xr_relrms_dec": relrms(xr1, xr0), "embed_relrms_clip": relrms(e1, e0),
                "rel_mapper_only": abs(la - l0) / l0, "rel_decoder_only": abs(lb - l0) / l0,
                "rel_clip_only": abs(lc - l0) / l0, "rel_dec_clip": abs(lbc - l0) / l0, "rel_all": abs(ld - l0) / l0,
                "signed": {"mapper": (la - l0) / l0, "decoder": (lb - l0) / l0, "clip": (lc - l0) / l0, "all": (ld - l0) / l0},
            })
            print(f"# [{lp}] seed {s}: " + json.dumps(rows[-1]), file=sys.stderr)
        out["modes"][lp] = rows
        del low
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
