"""bench.py — train-step images/sec of the hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 4
    python bench.py --gpus N --steps K --warmup W          (starts the N ranks itself, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] = SURVEY.md cfg2): MLP-Mixer 32x1024 mapper + VQGAN f16-16384 decoder
(256x256) + CLIP ViT-B/32, per-GPU batch 64 prompts, cutn 8, 16-bit MFMA with fp32 accumulation.  Precision contract: the
timed dtype is the package default, IEEE f16 (`--dtype f16`; main.DEFAULT_COMPUTE_DTYPE) — BASELINE.json names bf16, which has
the same width and MFMA rate; f16's finer mantissa is what holds the loss inside north_star's 1e-4 of the CPU reference.  The
line therefore also carries the SAME step timed in bf16 (`alt_dtype`) and a `deviation_from_baseline` note.  Synthetic
seeded token batches and random-init weights of that architecture (no network).  One "step" = text tower ->
mapper -> clamp -> VQ -> decoder -> cutouts(+noise) -> image tower -> spherical loss -> backward -> gradient
all-reduce (N>1) -> fused Adam.  Weak scaling: the per-GPU batch is fixed (the reference's semantics, main.py:647,678).

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" + "cpu_baseline").  Round 6 added:
`overflow_steps` (timed steps whose f16 backward tripped the non-finite guard: must be 0), `allocator_in_timed_region` (new segments of
torch's caching allocator between the first and the last timed step: the line reserves one large block up front so that no hipMalloc
stalls a timed step), `roofline.hw_bound_ms` + `top_gaps_ms` (the hardware bound beside the self-referential `attainable_ms`),
`parity_full_size.grad_parity` (the oracle's gradients vs the HIP backward at full model size in the timed dtype) and, at N > 1,
`dp_exposure.busbw_GBps`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic forward GMAC per unit (SURVEY.md §8d / BASELINE.md §2)
GMAC = dict(mixer=86.07, vq=2.147, dec=126.37, img=4.409, txt=2.980)
PEAK_BF16_TFLOPS = 2516.6      # 256 CU x 4096 FLOP/clk x 2.4 GHz dense bf16 / f16 MFMA (MI355X_MICROARCH.md)
DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": torch.float32}


# SURVEY.md §8d table, per (mapper, depth, dim, vq_image_size) / decode size / perceptor
GMAC_MAPPER = {("mlp_mixer", 32, 1024, 16): 86.07, ("mlp_mixer", 8, 128, 16): 0.856, ("mlp_mixer", 1, 1024, 32): 17.85,
               ("vitgan", 32, 1024, 16): 6.53, ("xtransformer", 16, 256, 32): 28.12}
GMAC_DEC = {16: (126.37, 2.147), 32: (508.72, 8.590)}                     # vq_image_size -> (decoder, VQ as written)
GMAC_CLIP = {"ViT-B/32": (4.409, 2.980), "ViT-L/14": (81.01, 6.65)}        # image per cutout, text per prompt


def step_tflop(B, cutn, mapper=("mlp_mixer", 32, 1024, 16), clip_model="ViT-B/32"):
    """FLOP/step = 2 * [B*txt + B*(3*mapper + vq + 2*dec + 2*cutn*img)]  (SURVEY.md §8d / BASELINE.md §2), in TFLOP;
    None for a configuration the survey's table does not cover."""
    arch = {"openclip/ViT-L-14/laion2b_s32b_b82k": "ViT-L/14"}.get(clip_model, clip_model)
    if mapper not in GMAC_MAPPER or mapper[3] not in GMAC_DEC or arch not in GMAC_CLIP:
        return None
    dec, vq = GMAC_DEC[mapper[3]]
    img, txt = GMAC_CLIP[arch]
    return 2.0 * (B * txt + B * (3 * GMAC_MAPPER[mapper] + vq + 2 * dec + 2 * cutn * img)) / 1e3


def build(args, device):
    from feed_forward_vqgan_clip_amd import clip as fclip
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd import main as fmain
    from feed_forward_vqgan_clip_amd import vqgan as fvq
    from feed_forward_vqgan_clip_amd.optim import FusedAdam

    cdt = DTYPES[args.dtype]
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=args.dim, depth=args.depth, dropout=0, cutn=args.cutn,
                       batch_size=args.batch, repeat=1, nb_noise=None, diversity_coef=0, clip_model=args.clip_model,
                       model_type=args.model_type, vq_image_size=args.vq_image_size,
                       augs=None if args.augs == "default" else args.augs.split(","),
                       augment_sequential=not getattr(args, "augment_fused", False))
    torch.manual_seed(1234)
    net = fmain.build_model(cfg, 256)
    mixer_sd = {k: v.detach().clone() for k, v in net.state_dict().items()} if args.keep_cpu_weights else None
    net = net.to(device).prepare(cdt)
    vq_sd = fvq.random_state_dict(fvq.F16_16384, seed=1234)
    arch, quick = fmain.clip_arch(args.clip_model)
    clip_sd = fclip.random_state_dict(arch, seed=1234)
    vq = fvq.VQGAN(vq_sd, fvq.F16_16384, cdt, fp8=getattr(args, "dec_fp8", False))
    perceptor = fclip.CLIP(clip_sd, cdt, quick_gelu=quick, fp8=args.clip_fp8)
    opt = FusedAdam(net.parameters(), lr=cfg.lr)
    opt.loss_scale = args.loss_scale if cdt == torch.float16 else 1.0
    if hvd.is_distributed():
        opt = hvd.DistributedOptimizer(opt, wire_dtype=torch.bfloat16 if args.grad_wire == "bf16" else None,
                                       tail_wire_dtype=torch.bfloat16 if getattr(args, "grad_wire_tail", "fp32") == "bf16" else None)
        hvd.broadcast_parameters(net, root_rank=0)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    if getattr(args, "reserve", True) and not os.environ.get("FFVC_SHARE_DEVICE"):      # (ranks sharing one GPU in the tests: no)
        fmain.reserve_device_memory(device=device)      # one large block for the caching allocator (see its docstring); FFVC_RESERVE_GIB=0 off
    return cfg, stepper, (mixer_sd, vq_sd, clip_sd)


def effective_cores():
    """Host cores this process may actually use: min(affinity, cgroup v2/v1 CPU quota).  (The GPU boxes expose
    256 logical CPUs behind a 16-CPU quota; oversubscribing them makes the oracle ~100x slower.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def cpu_baseline(sds, cutn, seconds_budget=30.0, augs="default"):
    """The oracle (CPU restatement, fp32) timed on this box's host cores on a bounded sample of the same
    workload: full train steps (fwd + loss + bwd + Adam) at batch 4 of the cfg2 models."""
    from feed_forward_vqgan_clip_amd import main as fmain
    from feed_forward_vqgan_clip_amd import vqgan as fvq
    from oracle import mappers as omap
    from oracle import step as ostep

    mixer_sd, vq_sd, clip_sd = sds
    cores = effective_cores()
    torch.set_num_threads(cores)
    B = 4
    tok = fmain.synthetic_tokens(B, seed=99)
    params = {k: v.clone().requires_grad_(True) for k, v in mixer_sd.items()}
    plist = list(params.values())
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in plist]
    cb = vq_sd["quantize.embedding.weight"]
    g = torch.Generator().manual_seed(5)
    facs = (torch.rand(cutn * B, generator=g) * 0.1).view(-1, 1, 1, 1)
    noise = torch.randn(cutn * B, 3, 224, 224, generator=g)
    from feed_forward_vqgan_clip_amd import augment as faug
    # raw kornia draws, applied by the oracle as the reference applies them (nn.Sequential, operator after operator:
    # oracle/kornia_aug.apply_chain)
    chain = None if augs == "R" else faug.draw_chain(cutn * B, 224, faug.DEFAULT if augs == "default" else
                                                     tuple(a for a in augs.split(",") if a != "R"), generator=g)

    first_step = {}

    def one(step):
        loss, mid = ostep.train_step_loss(
            lambda sd, f: omap.mixer_forward(sd, f, image_size=16, channels=256, depth=len([k for k in sd if k.endswith(".0.norm.weight")])),
            params, vq_sd, clip_sd, tok, cutn=cutn, cut_size=224, z_min=cb.min().item(), z_max=cb.max().item(),
            facs=facs, noise=noise, aug_chain=chain)
        grads = torch.autograd.grad(loss, plist)
        if step == 1:
            # the reference's step is zero_grad -> backward -> step (main.py:825-837): keep what the FIRST oracle step produced — the
            # gradient of every mapper parameter and the codes it decoded — for the full-size gradient parity leg (full_size_parity)
            first_step["grads"] = {k: g_.detach().clone() for k, g_ in zip(params, grads)}
            first_step["oidx"] = ostep.vq_indices(mid["z"].detach().movedim(1, 3), cb).view(-1)
        with torch.no_grad():
            ostep.adam_step(plist, grads, state, 1e-3, step)
        return float(loss.detach())

    t0 = time.time()
    loss0 = one(1)               # first step doubles as warm-up (allocator, MKL threads); its loss is the parity reference
    first = time.time() - t0
    n, t0 = 0, time.time()
    while first + (time.time() - t0) < seconds_budget and n < 3:
        one(n + 2)
        n += 1
    if n == 0:                   # one step already used the budget: report it (includes warm-up cost)
        n, dt = 1, first
    else:
        dt = (time.time() - t0) / n
    out = {"value": B / dt, "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": f"{n} full oracle train step(s) (fwd+loss+bwd+Adam, fp32) at batch {B}, cutn {cutn}, same cfg2 "
                     f"models/shapes; {dt:.2f} s/step"}
    return out, {"loss": loss0, "tok": tok, "facs": facs.view(-1), "noise": noise, "aug_chain": chain,
                 "grads": first_step.get("grads"), "oidx": first_step.get("oidx"), "lr": 1e-3}


def _relrms(a, b):
    a, b = a.double(), b.double()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt().clamp_min(1e-30))


def full_size_parity(args, sds, ref):
    """Parity of the timed mode at the benchmark's own model sizes, as a per-stage error budget.

    (1) the oracle (CPU fp32 restatement, pinned to the reference by tests/golden) vs the HIP fp32-MFMA mode at batch
        `pb` (default 4) -> `rel_fp32`: pins the HIP path's algorithm at full size;
    (2) the timed 16-bit mode vs that fp32 run, stage by stage at batch `pb`: mapper (z rel-rms, VQ code agreement),
        decoder (xr with the SAME codes), image tower (embed), loss with the reference's codes handed to the decoder
        (`rel_same_codes`: the VQ argmin is a discontinuity of the reference itself) and free-running (`rel_free`);
    (3) free-running at the benchmark's batch (HIP fp32 vs timed mode on the same weights / inputs) -> `rel_free_bench_batch`.
    """
    from feed_forward_vqgan_clip_amd import augment as faug
    from feed_forward_vqgan_clip_amd import clip as fclip
    from feed_forward_vqgan_clip_amd import main as fmain
    from feed_forward_vqgan_clip_amd import vqgan as fvq
    from feed_forward_vqgan_clip_amd.optim import FusedAdam
    from oracle import mappers as omap
    from oracle import step as ostep
    mixer_sd, vq_sd, clip_sd = sds
    pb, cutn = args.parity_batch, args.cutn
    augs = None if args.augs == "default" else args.augs.split(",")

    def inputs(B, seed):
        g = torch.Generator().manual_seed(seed)
        tok = fmain.synthetic_tokens(B, seed=seed)
        facs = torch.rand(cutn * B, generator=g) * 0.1
        noise = torch.randn(cutn * B, 3, 224, 224, generator=g)
        names = faug.DEFAULT if augs is None else tuple(a for a in augs if a != "R")
        chain = faug.draw_chain(cutn * B, 224, names, generator=g) if names else None
        return tok, facs, noise, chain

    def make(cdt, B):
        cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=args.dim, depth=args.depth, dropout=0, cutn=cutn,
                           batch_size=B, repeat=1, nb_noise=None, diversity_coef=0, clip_model="ViT-B/32",
                           model_type="mlp_mixer", vq_image_size=16, augs=augs, augment_sequential=not args.augment_fused)
        net = fmain.build_model(cfg, 256)
        net.load_state_dict(mixer_sd)
        net = net.cuda().prepare(cdt)
        return fmain.TrainStep(cfg, net, fvq.VQGAN(vq_sd, fvq.F16_16384, cdt), fclip.CLIP(clip_sd, cdt),
                               FusedAdam(net.parameters(), lr=1e-3))

    def run(st, tok, facs, noise, chain, force_idx=None):
        # the HIP step consumes the SAME raw draws through the plan the timed configuration uses (default: one launch per
        # warp = kornia's order of resamples; --augment-fused: the composed single launch)
        segs = None if chain is None else faug.to_device(faug.plan(chain, len(facs), 224, sequential=not args.augment_fused), "cuda")
        with torch.no_grad():
            loss, mid = st.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(), aug_params=segs, force_idx=force_idx)
        return float(loss), mid

    res = {"batch": pb, "timed_dtype": args.dtype,
           "oracle_augmentation": "kornia 0.5.10 nn.Sequential restatement (oracle/kornia_aug.apply_chain) on the same raw draws",
           "hip_augmentation": "fused single resample (opt-in)" if args.augment_fused else "sequential plan (default): kornia's two resamples evaluated in ONE launch (ffvc_augment_seq_*)"}
    tok, facs, noise, prm = inputs(pb, 99)
    cb = vq_sd["quantize.embedding.weight"]
    torch.set_num_threads(effective_cores())
    with torch.no_grad():                       # (1) the oracle, forward only
        oloss, omid = ostep.train_step_loss(
            lambda sd, f: omap.mixer_forward(sd, f, image_size=16, channels=256, depth=args.depth), mixer_sd, vq_sd, clip_sd,
            tok, cutn=cutn, cut_size=224, z_min=cb.min().item(), z_max=cb.max().item(), facs=facs.view(-1, 1, 1, 1),
            noise=noise, aug_chain=prm)
    res["loss_oracle_fp32"] = float(oloss)
    st32 = make(torch.float32, pb)
    l32, m32 = run(st32, tok, facs, noise, prm)
    res["loss_hip_fp32"] = l32
    res["rel_fp32"] = abs(l32 - float(oloss)) / abs(float(oloss))
    oidx = ostep.vq_indices(omid["z"].movedim(1, 3), cb).view(-1)
    res["vq_agree_fp32"] = float((m32["indices"].cpu().view(-1) == oidx).float().mean())
    lo = DTYPES[args.dtype]
    if lo != torch.float32:                     # (2) stage budget of the timed mode against the fp32 run
        stl = make(lo, pb)
        ll, ml = run(stl, tok, facs, noise, prm)
        lsc, msc = run(stl, tok, facs, noise, prm, force_idx=m32["indices"])
        res.update({
            "loss_hip_" + args.dtype: ll,
            "z_relrms": _relrms(ml["z"], m32["z"]),
            "vq_agree": float((ml["indices"] == m32["indices"]).float().mean()),
            "vq_flips": int((ml["indices"] != m32["indices"]).sum()), "vq_positions": int(m32["indices"].numel()),
            "xr_relrms_same_codes": _relrms(msc["xr"], m32["xr"]),
            "embed_relrms_same_codes": _relrms(msc["embed"], m32["embed"]),
            "rel_same_codes": abs(lsc - l32) / abs(l32),
            "rel_free": abs(ll - l32) / abs(l32),
            "rel_" + args.dtype: abs(ll - float(oloss)) / abs(float(oloss)),
            "rel_free_vs_oracle": abs(ll - float(oloss)) / abs(float(oloss)),      # timed dtype, free-running, vs the kornia-chain CPU oracle
            "rel_same_codes_vs_oracle": abs(lsc - float(oloss)) / abs(float(oloss)),
        })
        del stl
    del st32
    torch.cuda.empty_cache()
    if ref is not None and ref.get("grads") is not None and pb == len(ref["tok"]):
        # (2b) GRADIENT parity at full model size in the timed dtype (reference: zero_grad -> backward -> step, main.py:825-837).
        # The CPU baseline's first oracle step (batch 4, same weights) left its mapper gradients behind; the HIP step runs on the SAME
        # prompts / draws / noise with the oracle's codes handed to the decoder (the argmin is a discontinuity of the reference
        # itself), loss-scaled backward as in the timed step, then ONE Adam update.
        res["grad_parity"] = grad_parity(make, lo, args, ref)
        torch.cuda.empty_cache()
    if lo != torch.float32 and args.batch > pb:  # (3) the benchmark's own batch, free-running
        tok, facs, noise, prm = inputs(args.batch, 123)
        st32 = make(torch.float32, args.batch)
        l32, m32 = run(st32, tok, facs, noise, prm)
        del st32
        torch.cuda.empty_cache()
        stl = make(lo, args.batch)
        ll, ml = run(stl, tok, facs, noise, prm)
        res.update({"bench_batch": args.batch, "loss_hip_fp32_bench_batch": l32, "loss_timed_bench_batch": ll,
                    "rel_free_bench_batch": abs(ll - l32) / abs(l32),
                    "vq_agree_bench_batch": float((ml["indices"] == m32["indices"]).float().mean())})
        if not args.no_oracle_bench_batch:
            # the CPU oracle (kornia-chain augmentations) at the benchmark's own batch, forward only (~1 min of host time)
            t0 = time.time()
            with torch.no_grad():
                ol, _ = ostep.train_step_loss(
                    lambda sd, f: omap.mixer_forward(sd, f, image_size=16, channels=256, depth=args.depth), mixer_sd, vq_sd, clip_sd,
                    tok, cutn=cutn, cut_size=224, z_min=cb.min().item(), z_max=cb.max().item(), facs=facs.view(-1, 1, 1, 1),
                    noise=noise, aug_chain=prm)
            res.update({"loss_oracle_fp32_bench_batch": float(ol), "oracle_bench_batch_s": round(time.time() - t0, 1),
                        "rel_fp32_bench_batch_vs_oracle": abs(l32 - float(ol)) / abs(float(ol)),
                        "rel_free_bench_batch_vs_oracle": abs(ll - float(ol)) / abs(float(ol))})
        del stl
        torch.cuda.empty_cache()
    return res


def grad_parity(make, cdt, args, ref):
    """HIP backward of the timed dtype vs the oracle's gradients (see full_size_parity (2b)) -> dict for the bench line."""
    from feed_forward_vqgan_clip_amd import augment as faug
    from feed_forward_vqgan_clip_amd import ops
    tok, facs, noise, chain = ref["tok"], ref["facs"], ref["noise"], ref["aug_chain"]
    st = make(cdt, len(tok))
    ls = float(args.loss_scale) if cdt == torch.float16 else 1.0
    st.opt.loss_scale = ls
    segs = None if chain is None else faug.to_device(faug.plan(chain, len(facs), 224, sequential=not args.augment_fused), "cuda")
    loss, _ = st.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(), aug_params=segs, force_idx=ref["oidx"].cuda())
    st.opt.zero_grad()
    (loss if ls == 1.0 else loss * ls).backward()
    ops.join_side_stream()
    torch.cuda.synchronize()
    named = dict(st.net.named_parameters())
    before = {k: p.detach().float().cpu().clone() for k, p in named.items()}
    dot = nh = no = 0.0
    per, zero_like = {}, {}
    total = sum(g_.numel() for g_ in ref["grads"].values())
    grms = (sum(float(g_.double().pow(2).sum()) for g_ in ref["grads"].values()) / total) ** 0.5
    for k, go in ref["grads"].items():
        gh = named[k].grad.detach().float().cpu().double() / ls
        go = go.double()
        dot += float((gh * go).sum())
        nh += float(gh.pow(2).sum())
        no += float(go.pow(2).sum())
        # structurally zero gradients (pre-norm mixer: the residual stream's gradient sums to zero over the channel axis, so the
        # second token-mixing bias has none): the oracle holds rounding noise there -> HIP's values against the global scale instead
        if float(go.pow(2).mean().sqrt()) < 1e-3 * grms:
            zero_like[k] = float(gh.pow(2).mean().sqrt()) / grms
        else:
            per[k] = float((gh - go).pow(2).sum().sqrt() / go.pow(2).sum().sqrt().clamp_min(1e-30))
    worst = max(per.items(), key=lambda kv: kv[1])
    st.opt.step()                                   # the update itself (fused Adam, step 1)
    torch.cuda.synchronize()
    lr, eps = ref.get("lr", 1e-3), 1e-8
    dd = do = agree = cnt = 0.0
    for k, go in ref["grads"].items():
        d_h = named[k].detach().float().cpu().double() - before[k].double()
        go = go.double()
        d_o = -lr * go / (go.abs() + eps)           # torch.optim.Adam, step 1: m_hat = g, v_hat = g^2 (oracle/step.py::adam_step)
        dd += float((d_h - d_o).pow(2).sum())
        do += float(d_o.pow(2).sum())
        if k not in zero_like:
            agree += float((torch.sign(d_h) == torch.sign(d_o)).sum())
            cnt += d_o.numel()
    rs = sorted(per.values())
    out = {"batch": len(tok), "dtype": {torch.float16: "f16", torch.bfloat16: "bf16"}.get(cdt, "fp32"), "loss_scale": ls,
           "codes": "the oracle's (forced)", "loss_hip": float(loss), "loss_oracle": ref["loss"],
           "grad_cosine": dot / max((nh * no) ** 0.5, 1e-300), "grad_flat_relrms": (max(nh + no - 2 * dot, 0.0) / max(no, 1e-300)) ** 0.5,
           "grad_norm_ratio": (nh / max(no, 1e-300)) ** 0.5, "worst_tensor": worst[0], "worst_tensor_relrms": worst[1],
           "median_tensor_relrms": rs[len(rs) // 2], "tensors": len(rs),
           "structurally_zero_tensors": len(zero_like), "structurally_zero_max_rms_over_global_rms": max(zero_like.values()) if zero_like else 0.0,
           "adam_delta_relrms": (dd / max(do, 1e-300)) ** 0.5, "adam_delta_sign_agreement": agree / max(cnt, 1.0),
           "note": "oracle gradients = first CPU-baseline step (fp32, batch 4); HIP = timed dtype, loss-scaled backward, same weights / "
                   "prompts / draws; Adam step 1 moves every element by ~lr*sign(g): adam_delta_* compare the update elementwise; "
                   "worst / median over the tensors with a non-zero reference gradient (oracle rms >= 1e-3 x global rms)"}
    del st
    return out


HBM_ATTAINABLE_BPS = 6.3e12      # what a streaming kernel reaches of the 8 TB/s HBM3E peak on this part (MI355X_MICROARCH.md)


def attainable_leg(stepper, tok, prof_instep, hbm_instep, ms_per_step, table_path=None):
    """Two bounds beside the measured step.

    `attainable_ms` — what the step's OWN kernels allow: every distinct GEMM launch of the step re-issued ALONE, back to back (same
    pointers / epilogue / flags: kernels.REPLAY; row-split + skinny remainders and the fp8 entry points replay as the launch group
    they are), HBM-bound kernels at their algorithmic bytes / 6.3 TB/s, everything else at its measured time in a SERIALISED step
    (weight gradients on the main stream, no text prefetch).  It is a scheduling statement: "the step is the sum of its kernels".

    `hw_bound_ms` — what the HARDWARE allows for the same work: every GEMM's FLOPs at the best sustained rate any launch of this
    replay reached (stated: `hw_bound_gemm_rate_tflops`, the power-capped library ceiling on this box), HBM-bound kernels at their
    MINIMAL bytes (each operand once) / 6.3 TB/s, the fused token-mixing / attention / augmentation / reduction kernels at
    max(FLOPs / that rate, minimal bytes / 6.3 TB/s), and what has no model at its measured time (`unmodelled`).  The table's
    gap column = (isolated or serialised us - bound us) x launches: milliseconds recoverable per kernel, sorted.
    """
    from feed_forward_vqgan_clip_amd import kernels as K
    from feed_forward_vqgan_clip_amd import ops
    ops.set_wgrad_side_stream(False)
    stepper(tok)                                  # (untimed: the serialised order allocates differently, let the allocator settle)
    # the serialised step WITHOUT per-launch events first: with ~3000 event records per step the host, not the GPU, paces the
    # instrumented pass, and what it leaves idle must not be booked as "unmodelled" kernel time
    evp0, evp1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    evp0.record()
    stepper(tok)
    stepper(tok)
    evp1.record()
    torch.cuda.synchronize()
    wall_plain = evp0.elapsed_time(evp1) / 2.0
    K.PROFILE, K.HBM_PROFILE, K.AUX_PROFILE = [], [], []
    del K.HBM_MIN_BYTES[:]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    stepper(tok)
    ev1.record()
    torch.cuda.synchronize()
    prof_b, hbm_b, aux_b, hbm_min = K.PROFILE, K.HBM_PROFILE, K.AUX_PROFILE, list(K.HBM_MIN_BYTES)
    K.PROFILE = K.HBM_PROFILE = K.AUX_PROFILE = None
    # a third pass only CAPTURES the launches (kernels.REPLAY keeps every operand of the step alive, which sends the allocator to
    # hipMalloc: that pass is not timed)
    K.REPLAY = []
    stepper(tok)
    torch.cuda.synchronize()
    rep, K.REPLAY = K.REPLAY, None
    ops.set_wgrad_side_stream(True)
    wall_instr = ev0.elapsed_time(ev1)                              # the instrumented pass (host-paced: diagnostic only)
    wall_b = min(wall_plain, wall_instr)
    gemm_b = [e0.elapsed_time(e1) for _, _, e0, e1, _ in prof_b]
    hbm_b_ms = sum(e0.elapsed_time(e1) for _, _, e0, e1 in hbm_b)
    aux_b_ms = sum(e0.elapsed_time(e1) for _, _, _, e0, e1 in aux_b)
    rest_ms = max(0.0, wall_b - sum(gemm_b) - hbm_b_ms)            # everything that is neither a GEMM launch nor an HBM-profiled kernel
    unmodelled_ms = max(0.0, rest_ms - aux_b_ms)                   # ... of which no model exists (glue, torch ops, launch gaps)
    aligned = len(prof_instep) == len(prof_b) == len(rep)          # same program -> same launch sequence in both steps
    rows = {}
    for i, (cls, key, desc, keep) in enumerate(rep):
        r = rows.setdefault((cls,) + key, {"n": 0, "instep": 0.0, "serial": 0.0, "desc": desc, "flop": 2.0 * key[0] * key[1] * key[2] * key[3]})
        r["n"] += 1
        if aligned:
            r["instep"] += prof_instep[i][2].elapsed_time(prof_instep[i][3])
            r["serial"] += gemm_b[i]
    for r in rows.values():                                       # the isolated duration: 2 warm-up + 8 timed launches, back to back
        K.replay_gemm(r["desc"], 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.replay_gemm(r["desc"], 8)
        e1.record()
        torch.cuda.synchronize()
        r["iso_us"] = e0.elapsed_time(e1) / 8 * 1e3
    del rep
    gemm_iso_ms = sum(r["n"] * r["iso_us"] for r in rows.values()) / 1e3
    hb = {}
    for (name, nbytes, e0, e1), mb in zip(hbm_b, hbm_min):
        h = hb.setdefault(name, {"n": 0, "bytes": 0.0, "min_bytes": 0.0, "serial": 0.0})
        h["n"] += 1
        h["bytes"] += nbytes
        h["min_bytes"] += mb
        h["serial"] += e0.elapsed_time(e1)
    hbm_model_ms = sum(h["bytes"] for h in hb.values()) / HBM_ATTAINABLE_BPS * 1e3
    att = gemm_iso_ms + hbm_model_ms + rest_ms
    # ---- the hardware bound -------------------------------------------------------------------------------------------------
    # best sustained isolated rate of this replay: launches of at least 50 GFLOP (shorter ones are dominated by launch / ramp)
    big = [r["flop"] / (r["iso_us"] * 1e-6) / 1e12 for k, r in rows.items() if r["flop"] >= 5e10 and not k[0].endswith("f32")]
    best_rate = max(big) if big else max([r["flop"] / (r["iso_us"] * 1e-6) / 1e12 for r in rows.values()] + [1.0])
    f32_rates = [r["flop"] / (r["iso_us"] * 1e-6) / 1e12 for k, r in rows.items() if k[0].endswith("f32")]

    def gemm_bound_us(k, r):
        rate = 157.3 if k[0].endswith("f32") else (2.0 * best_rate if k[0].endswith("fp8") else best_rate)
        return r["flop"] / (rate * 1e12) * 1e6
    gemm_bound_ms = sum(r["n"] * gemm_bound_us(k, r) for k, r in rows.items()) / 1e3
    hbm_min_ms = sum(h["min_bytes"] for h in hb.values()) / HBM_ATTAINABLE_BPS * 1e3
    ax = {}
    for name, flops, nbytes, e0, e1 in aux_b:
        a = ax.setdefault(name, {"n": 0, "flops": 0.0, "bytes": 0.0, "serial": 0.0})
        a["n"] += 1
        a["flops"] += flops
        a["bytes"] += nbytes
        a["serial"] += e0.elapsed_time(e1)
    for a in ax.values():
        a["bound_ms"] = max(a["flops"] / (best_rate * 1e12), a["bytes"] / HBM_ATTAINABLE_BPS) * 1e3
    aux_bound_ms = sum(a["bound_ms"] for a in ax.values())
    hw_bound = gemm_bound_ms + hbm_min_ms + aux_bound_ms + unmodelled_ms
    gaps = []                                     # (gap ms, label)
    lines = [f"# attainable = sum(GEMM launches x isolated us) {gemm_iso_ms:.2f} ms + HBM kernels' algorithmic bytes / {HBM_ATTAINABLE_BPS / 1e12:.1f} TB/s "
             f"{hbm_model_ms:.2f} ms + the rest as measured in a serialised step {rest_ms:.2f} ms = {att:.2f} ms; step {ms_per_step:.2f} ms "
             f"-> frac_of_attainable {att / ms_per_step:.3f}; serialised step (no side streams) {wall_b:.2f} ms",
             f"# hw_bound = GEMM FLOPs at the best sustained isolated rate of this replay ({best_rate:.0f} TFLOP/s; fp32 MFMA launches at 157.3, fp8 at 2x) "
             f"{gemm_bound_ms:.2f} ms + HBM kernels' MINIMAL bytes / {HBM_ATTAINABLE_BPS / 1e12:.1f} TB/s {hbm_min_ms:.2f} ms + token-mix / attention / augmentation / "
             f"reduction kernels at max(FLOPs / that rate, bytes / {HBM_ATTAINABLE_BPS / 1e12:.1f} TB/s) {aux_bound_ms:.2f} ms + unmodelled (glue, torch ops, gaps; measured) "
             f"{unmodelled_ms:.2f} ms = {hw_bound:.2f} ms; step {ms_per_step:.2f} ms -> step / hw_bound {ms_per_step / hw_bound:.2f}",
             "# kernel class | M N K batch split_k flags act | launches/step | isolated us | TFLOP/s isolated | in-step us | serialised-step us | lost ms/step (in-step - isolated) | bound us | gap ms/step ((isolated - bound) x launches)"]
    for k, r in sorted(rows.items(), key=lambda kv: -(kv[1]["n"] * (kv[1]["iso_us"] - gemm_bound_us(kv[0], kv[1])))):
        ins, ser = r["instep"] / r["n"] * 1e3, r["serial"] / r["n"] * 1e3
        bnd = gemm_bound_us(k, r)
        gap = r["n"] * (r["iso_us"] - bnd) / 1e3
        gaps.append((gap, f"{k[0]} {k[1]}x{k[2]}x{k[3]} b{k[4]} f{k[6]} a{k[7]}"))
        lines.append(f"{k[0]:12s} {k[1]:6d} {k[2]:6d} {k[3]:6d} b{k[4]:<3d} sk{k[5]:<2d} f{k[6]:<5d} a{k[7]} | {r['n']:4d} | {r['iso_us']:8.1f} | "
                     f"{r['flop'] / (r['iso_us'] * 1e-6) / 1e12:7.1f} | {ins:8.1f} | {ser:8.1f} | {r['instep'] - r['n'] * r['iso_us'] / 1e3:7.3f} | {bnd:8.1f} | {gap:7.3f}")
    lines.append("# HBM-bound kernel | launches/step | MB/launch | model us (bytes / 6.3 TB/s) | in-step us | serialised-step us | lost ms/step | minimal MB/launch | bound us | gap ms/step ((serialised - bound) x launches)")
    hi = {}
    for name, nbytes, e0, e1 in hbm_instep or []:
        a = hi.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
    for name, h in sorted(hb.items(), key=lambda kv: -(kv[1]["serial"] - kv[1]["min_bytes"] / HBM_ATTAINABLE_BPS * 1e3)):
        model = h["bytes"] / h["n"] / HBM_ATTAINABLE_BPS * 1e6
        bnd = h["min_bytes"] / h["n"] / HBM_ATTAINABLE_BPS * 1e6
        ins = hi.get(name, [1, 0.0])
        gap = h["serial"] - h["n"] * bnd / 1e3
        gaps.append((gap, name))
        lines.append(f"{name:28s} | {h['n']:4d} | {h['bytes'] / h['n'] / 1e6:8.1f} | {model:8.1f} | {ins[1] / max(ins[0], 1) * 1e3:8.1f} | "
                     f"{h['serial'] / h['n'] * 1e3:8.1f} | {ins[1] - h['n'] * model / 1e3:7.3f} | {h['min_bytes'] / h['n'] / 1e6:8.1f} | {bnd:8.1f} | {gap:7.3f}")
    lines.append("# other kernel | launches/step | GFLOP/launch | MB/launch | serialised-step us | bound us (max(FLOPs / best rate, bytes / 6.3 TB/s)) | gap ms/step")
    for name, a in sorted(ax.items(), key=lambda kv: -(kv[1]["serial"] - kv[1]["bound_ms"])):
        gap = a["serial"] - a["bound_ms"]
        gaps.append((gap, name))
        lines.append(f"{name:28s} | {a['n']:4d} | {a['flops'] / a['n'] / 1e9:8.2f} | {a['bytes'] / a['n'] / 1e6:8.1f} | {a['serial'] / a['n'] * 1e3:8.1f} | "
                     f"{a['bound_ms'] / a['n'] * 1e3:8.1f} | {gap:7.3f}")
    lines.append(f"# unmodelled (measured, serialised step): {unmodelled_ms:.2f} ms")
    if table_path:
        os.makedirs(os.path.dirname(os.path.abspath(table_path)), exist_ok=True)
        with open(table_path, "w") as f:
            f.write("\n".join(lines) + "\n")
    gaps.sort(key=lambda g: -g[0])
    return {"attainable_ms": att, "frac_of_attainable": att / ms_per_step,
            "attainable_parts_ms": {"gemm_isolated": gemm_iso_ms, "hbm_at_6.3TBps": hbm_model_ms, "rest_serialised": rest_ms},
            "serialised_step_ms": wall_b, "serialised_step_instrumented_ms": wall_instr, "attainable_aligned": aligned,
            "attainable_note": "every distinct GEMM launch of the step replayed alone (same descriptor), HBM kernels at bytes / 6.3 TB/s, "
                               "everything else at its time in a serialised step (rest_serialised: measured, i.e. self-referential by "
                               "construction); frac_of_attainable = attainable_ms / ms_per_step",
            "hw_bound_ms": hw_bound, "step_over_hw_bound": ms_per_step / hw_bound,
            "hw_bound_parts_ms": {"gemm_flops_at_best_rate": gemm_bound_ms, "hbm_min_bytes_at_6.3TBps": hbm_min_ms,
                                  "tokmix_attention_augment_reductions_modelled": aux_bound_ms, "unmodelled_measured": unmodelled_ms},
            "hw_bound_gemm_rate_tflops": best_rate, "hw_bound_fp32_mfma_rate_seen_tflops": max(f32_rates) if f32_rates else None,
            "hw_bound_note": "GEMM FLOPs / the best sustained isolated rate of this replay (the library's power-capped ceiling on this box) + HBM "
                             "kernels at minimal bytes / 6.3 TB/s + token-mix / attention / augmentation / reductions at max(FLOP, byte) model + "
                             "unmodelled glue at its measured time",
            "top_gaps_ms": [{"kernel": lbl, "gap_ms": round(g, 3)} for g, lbl in gaps[:10]]}, lines


def launch_ranks(n):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE): start the N ranks as CHILD processes — one
    `python -m torch.distributed.run --nproc-per-node N bench.py <same argv>` — before this process has touched the GPU
    (nothing above this call initialises HIP; the parent never does), relay the children's output and print rank 0's JSON line as
    the parent's LAST stdout line.  Returns the exit code (non-zero child -> non-zero parent, also when no JSON line came back)."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL across processes); read by ROCr at hsa_init in the children
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"') and '"metric"' in ln:
            line = ln.rstrip("\n")
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print(f"bench.py: the {n}-rank launch produced no JSON line", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4, help="untimed steps first: torch's caching allocator settles on the step's allocation pattern "
                    "within the first four steps (a default of 2 put a one-off 140 ms hipMalloc stall into the timed region)")
    ap.add_argument("--batch", type=int, default=64, help="per-GPU prompts per step")
    ap.add_argument("--cutn", type=int, default=8)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=32)
    ap.add_argument("--model-type", default="mlp_mixer", choices=["mlp_mixer", "vitgan", "simple_vitgan", "xtransformer"],
                    help="mapper family (the headline workload cfg2 is mlp_mixer; others are dev / parity configs)")
    ap.add_argument("--vq-image-size", type=int, default=16, help="latent grid S (image = 16*S)")
    ap.add_argument("--clip-model", default="ViT-B/32", help="perceptor (main.py:1308-1333 names): ViT-B/32 (headline), "
                    "ViT-B/16, ViT-L/14, openclip/<arch>/<pretrained> (cfg5: openclip/ViT-L-14/laion2b_s32b_b82k)")
    ap.add_argument("--clip-fp8", action="store_true", help="image-tower linears (fwd + dgrad) on the fp8 MFMA path: e4m3 "
                    "weights / activations, e5m2 gradients, per-tensor delayed scaling (cfg5); NOT the headline configuration")
    ap.add_argument("--dec-fp8", action="store_true", help="the frozen decoder's large 3x3 convolutions (forward + dgrad) on the fp8 MFMA path "
                    "(e4m3 activations / filters, e5m2 gradients, per-tensor delayed scaling; cfg5); NOT the headline configuration")
    ap.add_argument("--augs", default="default", help="'default' = the reference's Af,Pe,Ji,Er (main.py:164-165), or a "
                    "comma list, e.g. 'R'")
    ap.add_argument("--augment-fused", action="store_true", help="opt-in: compose consecutive warps (Af -> Pe) into ONE interpolation / "
                    "launch instead of kornia's sequential resamples (config augment_sequential: false); NOT what the reference computes")
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "fp32"],
                    help="compute dtype of the timed step: 16-bit storage (bf16 | f16: same MFMA rate, f16 = 8x finer "
                         "mantissa + loss-scaled backward) with fp32 accumulation, or exact fp32 MFMA")
    ap.add_argument("--loss-scale", type=float, default=4096.0, help="static loss scale of the f16 backward pass")
    ap.add_argument("--parity-batch", type=int, default=4, help="batch of the oracle-vs-HIP parity leg")
    ap.add_argument("--grad-wire", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch prompts PER GPU (the reference's batch_size semantics, main.py:647,678); strong: "
                         "--batch is the GLOBAL batch, split evenly over the ranks")
    ap.add_argument("--no-prefetch-text", dest="prefetch_text", action="store_false",
                    help="encode each step's prompts inside the step instead of one step ahead on a side stream")
    ap.add_argument("--no-side-stream", action="store_true", help="A/B: weight gradients on the main stream (no second HIP stream)")
    ap.add_argument("--graph", action="store_true", help="replay the step behind the text tower as ONE captured hipGraph (TrainStep.enable_graph): "
                    "for the launch-bound mappers (VitGAN / x-transformer: ~2400 launches per step); single GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-oracle-bench-batch", action="store_true", help="skip the CPU oracle forward at the benchmark's batch in the parity leg")
    ap.add_argument("--no-alt-dtype", action="store_true", help="skip the second timing in the other 16-bit format")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch-invariant", action="store_true", help="skip the FFVC_SK_FIXUP=0 timing (a child process)")
    ap.add_argument("--no-attainable", action="store_true", help="skip the isolated-replay leg (roofline.attainable_ms)")
    ap.add_argument("--isolated-table", default=None, help="write the kernel | launches | isolated us | in-step us | lost ms table here")
    ap.add_argument("--gemm-shapes", type=int, default=0, help="print the N most expensive GEMM shapes (stderr)")
    ap.add_argument("--grad-wire-tail", default="fp32", choices=["fp32", "bf16"],
                    help="wire format of the exposed tail slices only (distributed.DistributedOptimizer tail policy)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))

    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd import kernels as K
    from feed_forward_vqgan_clip_amd import main as fmain

    hvd.init()
    rank, world = hvd.rank(), hvd.size()
    if world != args.gpus:
        if args.gpus != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.gpus > 1 and hvd.describe().get("ranks") != args.gpus:       # the process group must span exactly the GPUs the line claims
        raise SystemExit(f"--gpus {args.gpus}: the process group has {hvd.describe().get('ranks')} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(hvd.local_rank())
    device = torch.device("cuda", hvd.local_rank())
    if args.model_type != "mlp_mixer" or args.clip_model != "ViT-B/32" or args.vq_image_size != 16:
        args.no_cpu_baseline = True
    args.keep_cpu_weights = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    cfg, stepper, sds = build(args, device)
    if args.no_side_stream:
        from feed_forward_vqgan_clip_amd import ops as _ops
        _ops.set_wgrad_side_stream(False)

    if args.scaling == "strong":
        if args.batch % world:
            raise SystemExit(f"--scaling strong: global batch {args.batch} is not divisible by {world} ranks")
        args.batch //= world
    B = args.batch
    toks = fmain.synthetic_tokens(B * (args.steps + args.warmup + 1), seed=1234 + rank).to(device)

    def sync():
        if hvd.is_distributed():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    it = 0
    # every step also encodes the NEXT step's prompts (prefetched on a side stream under its backward pass): the timed
    # region contains exactly one text-tower pass per step, like the reference's loop
    batches = [toks[i * B:(i + 1) * B] for i in range(args.steps + args.warmup + 1)]
    for _ in range(args.warmup):
        stepper(batches[it], next_inp=batches[it + 1] if args.prefetch_text else None)
        it += 1
    if args.graph:
        if args.warmup < 1:
            raise SystemExit("--graph needs at least one eager warm-up step (scratch allocations happen there)")
        stepper.enable_graph(B, batches[it])          # one more (eager) warm-up step on the capture stream, then the capture
    sync()
    mem0 = torch.cuda.memory_stats(device)
    clk0 = K.clock_sample()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step GPU time (diagnostic only)
    # did any timed f16 step trip the non-finite guard?  The Adam kernel counts wavefront-level overflow events on the device
    # (FusedAdam._bad); a 4-byte device-to-device snapshot per step (no host synchronisation) tells afterwards WHICH steps did.
    inner_opt = getattr(stepper.opt, "opt", stepper.opt)
    bad_dev = getattr(inner_opt, "_bad", None)
    bad_hist = torch.zeros(args.steps + 1, dtype=bad_dev.dtype, device=device) if bad_dev is not None else None
    if bad_hist is not None:
        bad_hist[0:1].copy_(bad_dev.view(-1)[0:1])
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss, _ = stepper(batches[it], next_inp=batches[it + 1] if args.prefetch_text else None)
        marks[i + 1].record()
        if bad_hist is not None:
            bad_hist[i + 1:i + 2].copy_(bad_dev.view(-1)[0:1])
        it += 1
    clk1 = K.clock_sample()
    sync()
    dt = time.perf_counter() - t0
    mem1 = torch.cuda.memory_stats(device)
    step_ms = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(args.steps)]
    sclk_mhz = K.effective_clock_mhz(clk0, clk1)     # engine clock averaged over the timed region (and over the XCDs)
    if hvd.is_distributed():
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    import hashlib
    from feed_forward_vqgan_clip_amd import _lib as _flib
    out = {
        "lib_sha256": hashlib.sha256(open(_flib.LIB_PATH, "rb").read()).hexdigest(),   # which libffvc_hip.so produced the line
        "metric": "train-step images/sec (whole node), ViT-B/32 + VQGAN-f16 256x256, bs=64, 1/2/4/8 GPU",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "step_ms_main_stream": step_ms, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.dtype + ("+fp8(clip image tower linears)" if args.clip_fp8 else "") + ("+fp8(decoder 3x3 convs)" if args.dec_fp8 else ""), "data": "synthetic seeded token batches, random-init weights (no network)",
        "precision_recipe": {"bf16": "bf16 storage / MFMA inputs, fp32 accumulate, fp32 residual streams + norm statistics, "
                                     "fp32 text tower, VQ distances and loss",
                             "f16": "IEEE f16 storage / MFMA inputs, fp32 accumulate, fp32 residual streams + norm statistics, "
                                    f"fp32 text tower, VQ distances and loss, backward loss-scaled x{args.loss_scale:g}",
                             "fp32": "exact fp32 MFMA everywhere"}[args.dtype],
        "config": {"workload": (("cfg2: " if (args.depth, args.dim, args.vq_image_size, args.clip_model) == (32, 1024, 16, "ViT-B/32")
                                 else "") + f"MLP-Mixer {args.depth}x{args.dim}" if args.model_type == "mlp_mixer" else
                                f"{args.model_type} {args.depth}x{args.dim}") +
                               f" mapper + VQGAN f16-16384 decoder {16 * args.vq_image_size}x{16 * args.vq_image_size} + CLIP "
                               f"{args.clip_model}, per-GPU batch {B}, cutn {args.cutn}, augs {args.augs} + noise, full step "
                               "(fwd+loss+bwd+all-reduce+Adam)",
                   "global_batch": B * world, "parallelism": f"dp{world}", "grad_wire": args.grad_wire, "hip_graph": bool(args.graph),
                   "augmentation": "fused single resample (opt-in)" if args.augment_fused else "kornia order: one resample per warp (Af, then Pe+Ji+Er), both inside one launch",
                   "dp": hvd.describe()},
        "final_loss": float(loss.item()),
        # did torch's caching allocator go back to hipMalloc inside the timed region (a one-off stall of tens of ms)?  segments it created
        # and allocation retries between the first and the last timed step: both must be 0 for a clean line
        "allocator_in_timed_region": {"new_segments": int(mem1.get("segment.all.allocated", 0) - mem0.get("segment.all.allocated", 0)),
                                      "new_large_segments": int(mem1.get("segment.large_pool.allocated", 0) - mem0.get("segment.large_pool.allocated", 0)),
                                      "new_segment_MiB": round((mem1.get("reserved_bytes.all.allocated", 0) - mem0.get("reserved_bytes.all.allocated", 0)) / 2 ** 20, 1),
                                      "freed_segments": int(mem1.get("segment.all.freed", 0) - mem0.get("segment.all.freed", 0)),
                                      "alloc_retries": int(mem1.get("num_alloc_retries", 0) - mem0.get("num_alloc_retries", 0)),
                                      "reserved_GiB": round(mem1.get("reserved_bytes.all.current", 0) / 2 ** 30, 2)},
        # timed steps whose backward produced a non-finite scaled gradient (the Adam kernel's device-side guard): must be 0 for the
        # line to be 20 real updates
        "overflow_steps": (int((bad_hist[1:] != bad_hist[:-1]).sum().item()) if bad_hist is not None else None),
        "overflow_events": (int((bad_hist[-1] - bad_hist[0]).item()) if bad_hist is not None else None),
        # average engine clock over the timed steps (s_memtime / s_memrealtime): the chip clocks to its power budget, so the
        # MFMA peak actually available is PEAK x sclk / 2400 (profiles/r03_power_ceiling.txt)
        "sclk_mhz_effective": sclk_mhz,
        "deviation_from_baseline": (None if args.dtype == "bf16" else
                                    "BASELINE.json configs[1] says bf16; timed in IEEE f16 (same 16-bit width and MFMA rate, fp32 "
                                    "accumulate, 8x finer mantissa) because bf16 misses north_star's 1e-4 loss tolerance "
                                    "(6e-5..2e-3 measured); the bf16 timing of the same step is in alt_dtype"),
    }
    tf_step = step_tflop(B, args.cutn, (args.model_type, args.depth, args.dim, args.vq_image_size), args.clip_model)
    if tf_step:
        out["step_tflop"] = tf_step
        out["step_mfma_frac"] = tf_step / (ms_per_step * 1e-3) / PEAK_BF16_TFLOPS

    if not args.no_roofline:
        # one extra, event-bracketed step: per-launch HIP events on the launch stream (torch's current stream).  EVERY
        # rank runs it (its gradient all-reduces are collectives: a step on rank 0 alone would hang the others); only
        # rank 0 keeps the per-launch events
        K.PROFILE = [] if rank == 0 else None
        K.HBM_PROFILE = [] if rank == 0 else None
        dp_opt = stepper.opt if isinstance(stepper.opt, hvd.DistributedOptimizer) else None
        if dp_opt is not None:
            dp_opt.measure_exposure(True)        # how long after the backward pass each slice's all-reduce finished
        stepper(toks[:B])
        torch.cuda.synchronize()
        if dp_opt is not None and rank == 0:
            rep = dp_opt.exposure_report() or []
            bw = [r["busbw_GBps"] for r in rep if r.get("busbw_GBps")]
            out["dp_exposure"] = {"slices": len(rep), "exposed_ms": max([r["ms_after_backward"] or 0.0 for r in rep] + [0.0]),
                                  "late_slices": [r for r in rep if (r["ms_after_backward"] or 0.0) > 0.0][-8:],
                                  # achieved all-reduce bus bandwidth per slice (payload x 2(N-1)/N / time on the wire): against 7 xGMI
                                  # links x ~153 GB/s per GPU; the slowest and the median slice, and every slice's figure
                                  "busbw_GBps": {"min": min(bw), "median": sorted(bw)[len(bw) // 2], "max": max(bw)} if bw else None,
                                  "busbw_GBps_per_slice": [(r["slice"], r["MiB"], r.get("busbw_GBps")) for r in rep],
                                  "rccl_preset_enabled": os.environ.get("FFVC_RCCL_PRESET", "1") != "0"}
        if dp_opt is not None:
            dp_opt.measure_exposure(False)
    if rank == 0 and not args.no_roofline:
        prof, K.PROFILE = K.PROFILE, None
        agg = {}
        shapes = {}
        for name, flops, e0, e1, shp in prof:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            ms = e0.elapsed_time(e1)
            a[0] += 1
            a[1] += flops
            a[2] += ms * 1e-3
            sa = shapes.setdefault((name,) + shp, [0, 0.0, flops])
            sa[0] += 1
            sa[1] += ms
        if args.gemm_shapes:
            print("# top GEMM shapes by time: class M N K batch splitk | calls total_ms avg_ms TFLOP/s", file=sys.stderr)
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:args.gemm_shapes]:
                print(f"# {k[0]:14s} {k[1]:8d} {k[2]:6d} {k[3]:6d} b{k[4]:<5d} sk{k[5]:<3d} | {v[0]:4d} {v[1]:8.3f} {v[1]/v[0]:7.3f} "
                      f"{v[2]/(v[1]/v[0]*1e-3)/1e12:7.1f}", file=sys.stderr)
        dom = max(agg.items(), key=lambda kv: kv[1][2])
        name, (n, flops, secs) = dom
        peak = 157.3 if name.endswith("f32") else PEAK_BF16_TFLOPS
        # HBM bytes per launch come from rocprofv3 PMC passes (bench.py cannot run the profiler itself): the committed JSON
        # is stamped with the sha256 of the library it was measured on; a different library -> the number is stale -> null
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            import hashlib
            from feed_forward_vqgan_clip_amd import _lib as flib
            pmc_name = next((n for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), "r02_pmc_traffic.json")
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_name)))
            cls = name.rsplit("_", 1)[0]
            if cls in pmc:
                traffic_src = "profiles/" + pmc_name
                traffic_stale = pmc.get("_lib_sha256") != hashlib.sha256(open(flib.LIB_PATH, "rb").read()).hexdigest()
                if not traffic_stale:
                    traffic = pmc[cls]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {"bound": "mfma", "kernel": name, "launches_per_step": n,
                           "avg_launch_ms": secs / n * 1e3, "achieved": flops / secs / 1e12, "peak": peak,
                           "unit": "TFLOP/s", "frac": flops / secs / 1e12 / peak, "traffic": traffic,
                           "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                           "traffic_source": traffic_src, "traffic_stale": traffic_stale, "flop_per_launch": flops / n,
                           # the same fraction against the peak at the clock the chip actually held during the timed steps
                           "frac_at_effective_clock": flops / secs / 1e12 / (peak * sclk_mhz / 2400.0) if sclk_mhz > 0 else None}
        # the five most expensive GEMM shapes of the step (what actually directs kernel work; the class above averages ~20 shapes)
        out["roofline"]["shapes"] = [
            {"class": k[0], "M": k[1], "N": k[2], "K": k[3], "batch": k[4], "split_k": k[5], "launches": v[0], "ms": round(v[1], 3),
             "avg_launch_us": round(v[1] / v[0] * 1e3, 1), "tflops": round(v[2] / (v[1] / v[0] * 1e-3) / 1e12, 1),
             "frac": round(v[2] / (v[1] / v[0] * 1e-3) / 1e12 / (157.3 if k[0].endswith("f32") else PEAK_BF16_TFLOPS), 4)}
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:5]]
        out["kernel_classes"] = {k: {"launches": v[0], "ms": v[2] * 1e3, "tflops": v[1] / max(v[2], 1e-12) / 1e12}
                                 for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}
        out["gemm_ms_per_step"] = sum(v[2] for v in agg.values()) * 1e3
        hb, K.HBM_PROFILE = K.HBM_PROFILE, None
        hagg = {}
        for hname, nbytes, e0, e1 in hb or []:
            a = hagg.setdefault(hname, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += nbytes
            a[2] += e0.elapsed_time(e1) * 1e-3
        # HBM-bound kernels of the step: algorithmic bytes / HIP-event time, against the 8 TB/s HBM3E peak
        out["hbm_kernels"] = {k_: {"launches": v[0], "ms": v[2] * 1e3, "GB/s": v[1] / max(v[2], 1e-12) / 1e9,
                                   "frac_of_8TBps": v[1] / max(v[2], 1e-12) / 8e12} for k_, v in hagg.items()}
        if world == 1 and not args.no_attainable:
            att, table = attainable_leg(stepper, toks[:B], prof, hb, ms_per_step, args.isolated_table)
            out["roofline"].update(att)
            if args.gemm_shapes:
                print("\n".join(table), file=sys.stderr)
    if world == 1 and not args.no_alt_dtype and args.dtype in ("f16", "bf16") and not args.clip_fp8 and not args.dec_fp8:
        # the same step in the OTHER 16-bit storage format (BASELINE's bf16 when the headline is f16), same box, same inputs
        alt = "bf16" if args.dtype == "f16" else "f16"
        del stepper
        torch.cuda.empty_cache()
        a2 = argparse.Namespace(**vars(args))
        a2.dtype, a2.keep_cpu_weights = alt, False
        _, st2, _ = build(a2, device)
        n2 = max(3, min(args.steps, 8))
        for i in range(2):
            st2(batches[i % len(batches)], next_inp=batches[(i + 1) % len(batches)] if args.prefetch_text else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n2):
            l2_, _ = st2(batches[(2 + i) % len(batches)], next_inp=batches[(3 + i) % len(batches)] if args.prefetch_text else None)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / n2 * 1e3
        out["alt_dtype"] = {"dtype": alt, "ms_per_step": ms2, "value": B / (ms2 * 1e-3), "steps": n2, "final_loss": float(l2_.item())}
        del st2
        torch.cuda.empty_cache()
        stepper = None
    if world > 1:
        torch.distributed.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        stepper = None
        torch.cuda.empty_cache()
        out["cpu_baseline"], ref = cpu_baseline(sds, args.cutn, augs=args.augs)
        out["parity_full_size"] = full_size_parity(args, sds, ref)
        out["parity_full_size"]["loss_oracle_fp32_cpu_baseline_step"] = ref["loss"]      # first (batch-4) step of the CPU baseline
    if rank == 0 and world == 1 and not args.no_batch_invariant and not args.no_cpu_baseline:
        # what the `batch_invariant` mode costs: the same step with FFVC_SK_FIXUP=0 (no shape-dependent in-kernel
        # split-K: every mapper output element is summed in an order fixed by (N, K) alone).  The switch is read once per process
        # -> a child process; this one is idle meanwhile.
        import subprocess
        stepper = None
        torch.cuda.empty_cache()
        env = dict(os.environ, FFVC_SK_FIXUP="0")
        # the child times THIS run's configuration: its argv minus the step counts and the diagnostic legs
        keep, skip = [], False
        for a_ in sys.argv[1:]:
            if skip:
                skip = False
                continue
            if a_ in ("--steps", "--warmup", "--isolated-table", "--gemm-shapes", "--gpus"):
                skip = True
                continue
            if a_.split("=")[0] in ("--steps", "--warmup", "--isolated-table", "--gemm-shapes", "--gpus", "--no-cpu-baseline", "--no-alt-dtype", "--no-roofline"):
                continue
            keep.append(a_)
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-alt-dtype", "--no-roofline"] + keep
        sds = ref = None                      # the parent keeps nothing on the GPU while the child is timed
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            child = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            out["batch_invariant"] = {"ms_per_step": child["ms_per_step"], "value": child["value"], "steps": 10,
                                      "note": "same step with FFVC_SK_FIXUP=0 (config batch_invariant): bit-identical latents whatever the batch size"}
        except Exception as e:        # noqa: BLE001 — a diagnostic leg must not cost the line
            out["batch_invariant"] = {"error": repr(e)[:200]}
    if rank == 0:
        # libraries that write to C stdio (RCCL's NCCL_DEBUG=VERSION banner) flush at exit, i.e. AFTER Python's own buffer:
        # push their text out first so that the JSON line is the last line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if hvd.is_distributed():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
