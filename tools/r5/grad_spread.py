"""Which gradient of the tiny bf16 step differs run to run?  The same forward + backward N times on identical
inputs / weights / codes: per parameter, the rel-rms spread between runs and the error against the CPU oracle; then the same with
intermediate tensors compared, to find the first kernel whose OUTPUT is not bit-reproducible."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_models_gpu as T                                        # noqa: E402
from feed_forward_vqgan_clip_amd import main as fmain              # noqa: E402
from oracle import mappers as omap, step as ostep                  # noqa: E402

cdt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
LS = float(sys.argv[3]) if len(sys.argv) > 3 else (8192.0 if cdt == torch.float16 else 1.0)
cfg, net, vq, perceptor, opt, clip_sd, vq_sd, tok, facs, noise = T._tiny_step(cdt)
stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
msd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
osd = {k: v.clone().requires_grad_(True) for k, v in msd.items()}
oloss, omid = ostep.train_step_loss(lambda sd, f: omap.mixer_forward(sd, f, image_size=12, channels=64, depth=2), osd, vq_sd, clip_sd, tok,
                                    cutn=4, cut_size=32, z_min=vq.z_min, z_max=vq.z_max, facs=facs.view(-1, 1, 1, 1), noise=noise,
                                    vq_cfg=T.TINY_VQ)
oloss.backward()
oidx = ostep.vq_indices(omid["z"].detach().movedim(1, 3), vq_sd["quantize.embedding.weight"])
runs, mids = [], []
for r in range(N):
    loss, mid = stepper.forward_loss(tok.cuda(), facs=facs.cuda(), noise=noise.cuda(), force_idx=oidx.cuda())
    for k in ("z", "xr", "embed"):
        mid[k].retain_grad()
    opt.zero_grad()
    (loss * LS).backward()
    torch.cuda.synchronize()
    runs.append({k: p.grad.detach().double().cpu().clone() / LS for k, p in net.named_parameters()})
    m = {k: v.detach().double().cpu().clone() for k, v in mid.items() if torch.is_tensor(v)}
    for k in ("z", "xr", "embed"):
        m["d" + k] = mid[k].grad.detach().double().cpu().clone() / LS
    mids.append(m)
rr = lambda a, b: ((a - b).pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)).item()   # noqa: E731
gmax = max(v.grad.abs().max().item() for v in osd.values())
print(f"# {cdt}: {N} runs, same inputs.  param | rel-rms vs oracle (min..max over runs) | max rel-rms between a run and the run mean | bit-identical runs")
rows = []
for k in runs[0]:
    if osd[k].grad.abs().max().item() <= 1e-4 * gmax:
        continue
    mean = sum(r[k] for r in runs) / N
    errs = [rr(r[k], osd[k].grad.double()) for r in runs]
    spread = max(rr(r[k], mean) for r in runs)
    same = all(torch.equal(r[k], runs[0][k]) for r in runs)
    rows.append((max(errs), k, min(errs), spread, same))
for e, k, lo, sp, same in sorted(rows, reverse=True):
    print(f"{k:42s} | {lo:.4f} .. {e:.4f} | {sp:.2e} | {same}")
print("# intermediates (d* = gradient wrt it): bit-identical across runs? | max rel-rms between a run and run 0 | max |diff| / max |value|")
for k in mids[0]:
    same = all(torch.equal(m[k], mids[0][k]) for m in mids)
    sp = max(rr(m[k], mids[0][k]) for m in mids[1:])
    mx = max(((m[k] - mids[0][k]).abs().max() / (mids[0][k].abs().max() + 1e-30)).item() for m in mids[1:])
    print(f"{k:12s} {same!s:5s} {sp:.2e} {mx:.2e}")
# against the oracle's intermediates' gradients
omid["z"].retain_grad() if False else None
