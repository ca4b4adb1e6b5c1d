"""How much of the gradient exchange is enqueued before backward ends?  Single rank, real RCCL communicator (FFVC_DP_FORCE=1):

    FFVC_DP_FORCE=1 python tools/dp_overlap.py [mlp_mixer|vitgan] [batch]

For every bucket of DistributedOptimizer: its size, the parameters in it, and how long before the end of the backward pass (event on the main stream) its
all-reduce had FINISHED (event on the exchange stream, right behind the all-reduce).  Prints a table and a JSON summary line;
exits non-zero unless every slice was enqueued inside backward() and the slices that were not finished before the last 2 % of
the pass are the tail of the launch order (the last gradients backward produces: `proj.weight` of the Mixer cut into slices,
the first Linear of the VitGAN mapper, plus the one or two slices in front of them).
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FFVC_DP_FORCE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

from feed_forward_vqgan_clip_amd import clip as fclip  # noqa: E402
from feed_forward_vqgan_clip_amd import distributed as hvd  # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain  # noqa: E402
from feed_forward_vqgan_clip_amd import vqgan as fvq  # noqa: E402
from feed_forward_vqgan_clip_amd.optim import FusedAdam  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "mlp_mixer"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    hvd.init()
    assert hvd.is_distributed() and hvd.describe()["backend"] == "nccl", hvd.describe()
    torch.cuda.set_device(0)
    extra = dict(num_heads=6) if kind == "vitgan" else {}
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=1024, depth=32, dropout=0, cutn=8, batch_size=B, repeat=1, nb_noise=None,
                       diversity_coef=0, clip_model="ViT-B/32", model_type=kind, vq_image_size=16, **extra)
    torch.manual_seed(1)
    cdt = torch.float16
    net = fmain.build_model(cfg, 256).cuda().prepare(cdt)
    vq = fvq.VQGAN(fvq.random_state_dict(fvq.F16_16384, seed=1), fvq.F16_16384, cdt)
    perceptor = fclip.CLIP(fclip.random_state_dict(fclip.VIT_B32, seed=1), cdt)
    inner = FusedAdam(net.parameters(), lr=cfg.lr)
    inner.loss_scale = 4096.0
    opt = hvd.DistributedOptimizer(inner)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    tok = fmain.synthetic_tokens(B, seed=2).cuda()
    names = {id(p): n for n, p in net.named_parameters()}
    a = opt.arena
    for _ in range(2):                                  # warm up (allocator, communicator)
        stepper(tok)
    torch.cuda.synchronize()
    events = {}
    orig = opt._enqueue

    def enqueue(b, g):
        orig(b, g)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())          # the exchange stream: right behind the all-reduce
        events[b] = ev

    opt._enqueue = enqueue
    loss, _ = stepper.forward_loss(tok)
    opt.zero_grad()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    (loss * inner.loss_scale).backward()
    t1.record()
    launched_in_backward = set(events)
    opt.step()
    torch.cuda.synchronize()
    bwd_ms = t0.elapsed_time(t1)
    total = sum(e - s for s, e, _ in opt.buckets) * 4
    rows, late = [], []
    for b, (s, e, idxs) in enumerate(opt.buckets):
        ms_before_end = events[b].elapsed_time(t1) if b in events else float("nan")
        pn = [names[id(a.plist[i])] for i in idxs]
        rows.append((b, (e - s) * 4 / 2**20, ms_before_end, b in launched_in_backward, pn[0] if len(pn) == 1 else f"{pn[-1]} .. {pn[0]}"))
    print(f"# {kind} 32x1024, batch {B}, f16, 1 rank on RCCL: backward {bwd_ms:.1f} ms, gradient bucket {total / 2**20:.0f} MiB in {len(opt.buckets)} slices")
    print("# slice   MiB   all-reduce done before backward's end [ms]   enqueued inside backward   wire   parameters")
    early = 0
    for b, mib, ms, inb, pn in rows:
        print(f"  {b:4d} {mib:6.1f} {ms:10.2f} {'yes' if inb else 'NO':>6s}   {str(opt._wire_of[b] or 'fp32').replace('torch.', ''):>8s}   {pn}")
        if ms == ms and ms > 0.02 * bwd_ms:
            early += mib
        else:
            late.append(pn)
    first = names[id(a.plist[0])]
    summary = {"mapper": kind, "batch": B, "backward_ms": bwd_ms, "slices": len(rows), "bucket_MiB": total / 2**20,
               "MiB_done_before_last_2pct_of_backward": early, "frac_bytes_early": early / (total / 2**20), "late": late,
               "first_parameter": first, "dp": hvd.describe()}
    print(json.dumps(summary))
    # the tail of the launch order = what DistributedOptimizer itself re-cut into 16 MiB pieces after the first step (the last
    # ~192 MiB of gradients the pass produces: proj.weight's slices, the first block's bucket) plus the tiny slice in front of them
    order = sorted(range(len(opt.buckets)), key=lambda b: -rows[b][2] if rows[b][2] == rows[b][2] else 1e9)   # earliest finished first
    tail = set(opt.tail_slices) | set(order[-2:])
    bad = [r for r in rows if not r[3]] + [r for r in rows if not (r[2] == r[2] and r[2] > 0.02 * bwd_ms) and r[0] not in tail]
    if bad:
        print("FAIL: slices outside the tail of the launch order that were late / not enqueued inside backward:", bad)
        sys.exit(1)


if __name__ == "__main__":
    main()
