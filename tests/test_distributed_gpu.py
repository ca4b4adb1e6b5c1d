"""Data-parallel path with the real HIP kernels: two processes share cuda:0 and exchange gradients over gloo (RCCL
refuses two ranks on one device; the collective itself is torch.distributed's, everything around it — flat bucket,
side-stream joins, bucket readiness from fused wgrad / LayerNorm kernels, 1/N folded into Adam — is ours).
DP-equivalence: two ranks x 2 prompts == one process x 4 prompts on identical noise / augmentation draws."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CLIP_CFG = dict(embed_dim=32, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8,
                context_length=16, vocab_size=96, transformer_width=64, transformer_heads=1, transformer_layers=2)
VQ_CFG = dict(ch=64, ch_mult=(1, 1, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32, z_channels=64,
              out_ch=3, embed_dim=64, n_embed=128)
CUTN, B = 4, 4


def _collect(q, procs, limit=120.0):
    """Results of all ranks, failing fast when a rank died and after `limit` seconds (a hung run must not eat the GPU
    budget of the caller).  Ranks announce themselves once they are past the process-group rendezvous: if that never
    happens the box could not host two ranks (environment, seen sporadically on shared boxes) and the test is skipped
    rather than failed; a hang AFTER the rendezvous is a failure."""
    import queue as _queue
    import time as _time
    res, ready, deadline = [], 0, _time.time() + limit
    while len(res) < len(procs):
        try:
            item = q.get(timeout=2)
            if item[0] == "ready":
                ready += 1
            else:
                res.append(item)
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"a worker died with exit code {dead}"
            if _time.time() >= deadline:
                if ready < len(procs):
                    pytest.skip(f"only {ready}/{len(procs)} ranks got through the gloo rendezvous in {limit:.0f} s")
                raise AssertionError(f"ranks hung after the rendezvous (no result within {limit:.0f} s)")
    res.sort(key=lambda t: t[0])
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    g = torch.Generator().manual_seed(3)
    tok = torch.zeros(B, 16, dtype=torch.long)
    for i, L in enumerate([3, 6, 9, 12]):
        tok[i, 0] = 94
        tok[i, 1:L] = torch.randint(1, 94, (L - 1,), generator=g)
        tok[i, L] = 95
    facs = torch.rand(CUTN, B, generator=g) * 0.1
    noise = torch.randn(CUTN, B, 3, 32, 32, generator=g)
    return tok, facs, noise


def _run(rank, world, steps=2, big=False, shard_of=None):
    """-> flat parameter vector after `steps` optimizer steps on this rank's shard.  big: a Mixer wide and deep enough for the
    grouped weight gradients (dim 1024, 10 blocks, 16 x 16 tokens: blocks 4-7 and 8-9 go out as grouped launches)."""
    from feed_forward_vqgan_clip_amd import clip as fclip
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd import main as fmain
    from feed_forward_vqgan_clip_amd import vqgan as fvq
    from feed_forward_vqgan_clip_amd.optim import FusedAdam
    cfg = fmain.Config(lr=1e-3, epochs=1, noise_dim=0, dim=1024 if big else 64, depth=10 if big else 2, dropout=0, cutn=CUTN,
                       batch_size=B // world, repeat=1, nb_noise=None, diversity_coef=0, clip_model="ViT-B/32", clip_dim=32, clip_size=32,
                       model_type="mlp_mixer", vq_image_size=16 if big else 12, augs=["R"])
    if shard_of is not None:                          # single process standing in for rank `shard_of[0]` of `shard_of[1]`: no exchange
        cfg.batch_size = B // shard_of[1]
    torch.manual_seed(5 + rank)                       # replicas start DIFFERENT: the broadcast has to repair them
    cdt = torch.float16 if big else torch.float32             # (the grouped launch is a 16-bit kernel)
    net = fmain.build_model(cfg, 64).cuda().prepare(cdt)
    if big:
        assert sum(1 for b in net._blocks for W in (b[4], b[5]) if W.group is not None) == 12
    vq = fvq.VQGAN(fvq.random_state_dict(VQ_CFG, 12), VQ_CFG, cdt)
    perceptor = fclip.CLIP(fclip.random_state_dict(CLIP_CFG, 11), cdt)
    opt = FusedAdam(net.parameters(), lr=cfg.lr)
    if big:
        opt.loss_scale = 1024.0
    if world > 1:
        opt = hvd.DistributedOptimizer(opt, bucket_bytes=(32 << 20) if big else (64 << 10), tail_bytes=0)
        assert len(opt.buckets) >= 2
        hvd.broadcast_parameters(net, root_rank=0)
        hvd.broadcast_optimizer_state(opt, root_rank=0)
    stepper = fmain.TrainStep(cfg, net, vq, perceptor, opt)
    tok, facs, noise = _inputs()
    shard = slice(rank * (B // world), (rank + 1) * (B // world))
    if shard_of is not None:
        shard = slice(shard_of[0] * (B // shard_of[1]), (shard_of[0] + 1) * (B // shard_of[1]))
    args = dict(facs=facs[:, shard].reshape(-1).cuda(), noise=noise[:, shard].reshape(-1, 3, 32, 32).cuda())
    # step 1 by hand, to look at the exchanged gradient (the bucket holds the SUM over ranks; 1/N lives in Adam)
    loss, _ = stepper.forward_loss(tok[shard].cuda(), **args)
    opt.zero_grad()
    ls = getattr(opt, "loss_scale", 1.0)
    (loss * ls).backward()
    if world > 1:
        opt.synchronize()
        opt._synced = True
    torch.cuda.synchronize()
    grads = (net._ffvc_arena.grads.detach() / (world * ls)).cpu().numpy().copy()
    opt.step()
    for _ in range(steps - 1):
        loss, _ = stepper(tok[shard].cuda(), **args)
    torch.cuda.synchronize()
    return net._ffvc_arena.params.detach().cpu().numpy().copy(), float(loss), grads


def _worker(rank, world, port, q, big=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=False)         # a hung rank prints where it is stuck
    torch.cuda.set_device(0)
    from feed_forward_vqgan_clip_amd import distributed as hvd
    hvd.init(backend="gloo")
    assert hvd.size() == world and hvd.rank() == rank
    q.put(("ready", rank))                                     # past the rendezvous
    params, loss, grads = _run(rank, world, big=big)
    (l,) = hvd.allreduce_scalars(torch.tensor(loss, device="cuda"))
    q.put((rank, params, float(l), grads))


def _two_ranks_once(limit, big=False):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, big)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, procs, limit)
    finally:
        for p in procs:                      # never leave a rank behind (it would keep the device / the port)
            if p.is_alive():
                p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_dp_world2_equals_single_process(cuda):
    # Two processes sharing ONE GPU through gloo is an arrangement only this test uses; on a loaded box a rank has been seen to
    # stall past the limit (cold imports, the other rank holding the device).  One retry on a fresh port tells an environment
    # hiccup from a defect: a real deadlock fails twice.
    try:
        res = _two_ranks_once(180.0)
    except AssertionError as e:
        if "hung after the rendezvous" not in str(e):
            raise
        res = _two_ranks_once(300.0)
    assert (res[0][1] == res[1][1]).all(), "replicas diverged"   # replicas stay bit-identical
    torch.manual_seed(0)
    ref, ref_loss, ref_grads = _run(0, 1)                      # rank-0 seed (5), all four prompts, no exchange
    g, rg = torch.from_numpy(res[0][3]).double(), torch.from_numpy(ref_grads).double()
    assert (res[0][3] == res[1][3]).all(), "exchanged gradients differ between ranks"
    relrms = ((g - rg).pow(2).mean().sqrt() / rg.pow(2).mean().sqrt()).item()
    assert relrms < 1e-4, relrms                               # averaged shard gradients == full-batch gradient
    # Adam turns round-off on zero-gradient parameters into +-lr steps, so parameters are compared where the gradient
    # is significant; two steps of lr = 1e-3
    sig = rg.abs() > 1e-3 * rg.abs().max()
    err = (torch.from_numpy(res[0][1]) - torch.from_numpy(ref)).abs()[sig].max().item()
    assert err < 1e-4, f"parameter deviation {err}"        # (gradient check above is the strict one)
    assert abs(res[0][2] - ref_loss) < 1e-3 * abs(ref_loss) + 1e-6, (res[0][2], ref_loss)


def test_dp_world2_with_grouped_weight_gradients(cuda):
    """Round 5: the channel-MLP weight gradients of consecutive blocks are deferred into grouped launches (ops.WgradGroup) and their
    parameters reported when the group goes out — not when autograd's own hook fires for the layer.  Two ranks (shared GPU, gloo
    exchange) on a Mixer that groups (1024 wide, 10 blocks, f16): replicas bit-identical, exchanged gradient == the single-process
    full-batch gradient up to f16 rounding.  (The first version of the deferral launched a bucket's all-reduce before the deferred
    gradient was written; the DistributedOptimizer's second-contribution check caught it.)"""
    try:
        res = _two_ranks_once(240.0, big=True)
    except AssertionError as e:
        if "hung after the rendezvous" not in str(e):
            raise
        res = _two_ranks_once(400.0, big=True)
    assert (res[0][1] == res[1][1]).all(), "replicas diverged"
    assert (res[0][3] == res[1][3]).all(), "exchanged gradients differ between ranks"
    # Reference without an exchange: ONE process runs the two shards one after the other from rank 0's weights (the same shapes,
    # hence the same kernels and roundings as the ranks) and averages.  (Against the full batch of 4 in one pass the f16 step differs
    # by ~20 % rms: other split-K shapes -> other roundings -> other VQ codes; that comparison is made in fp32 by the test above.)
    shards = [_run(0, 1, steps=1, big=True, shard_of=(r, 2)) for r in range(2)]
    rg = (torch.from_numpy(shards[0][2]).double() + torch.from_numpy(shards[1][2]).double()) / 2
    g = torch.from_numpy(res[0][3]).double()
    relrms = ((g - rg).pow(2).mean().sqrt() / rg.pow(2).mean().sqrt()).item()
    assert relrms < 1e-5, relrms                               # (the LayerNorm / bias column sums meet in fp32 atomics: 1e-8)


@pytest.mark.parametrize("kind", ["mlp_mixer", "vitgan"])
def test_exchange_is_enqueued_under_the_backward_pass(cuda, kind):
    """Single rank on a real RCCL communicator (FFVC_DP_FORCE=1): every slice of the gradient bucket except the slices of the LAST
    gradient of the backward pass has its all-reduce finished before the last 2 % of backward (tools/dp_overlap.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVC_DP_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dp_overlap.py"), kind, "4"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(r.stdout)
    assert summary["dp"]["backend"] == "nccl" and summary["frac_bytes_early"] > 0.85


def test_own_rccl_communicator_single_rank(cuda):
    """csrc/comm.hip: the library's own communicator (ffvc_rccl_unique_id -> ffvc_rccl_comm_create -> ffvc_allreduce_bucket ->
    ffvc_rccl_comm_destroy), resolved from the RCCL already in the process.  One rank: the sum over the ranks is the identity;
    what is checked is the plumbing — creation, the three wire dtypes, enqueueing on a foreign stream, error reporting."""
    from feed_forward_vqgan_clip_amd import kernels as K
    assert K.RcclComm.available() > 0
    comm = K.RcclComm(K.RcclComm.unique_id(), 0, 1)
    xs = torch.cuda.Stream()
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        x = torch.randn(3 << 20, device="cuda").to(dt)
        ref = x.clone()
        xs.wait_stream(torch.cuda.current_stream())
        comm.allreduce(x, xs)
        ev = torch.cuda.Event()
        ev.record(xs)
        torch.cuda.current_stream().wait_event(ev)
        y = x * 1                                     # consumer on the current stream, ordered by the event only
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(y, ref)
    with pytest.raises(TypeError):                    # only the gradient / wire dtypes travel
        comm.allreduce(torch.zeros(4, dtype=torch.int64, device="cuda"))
    comm.destroy()
    with pytest.raises(ValueError):
        K.RcclComm(b"short", 0, 1)


def test_bucket_exchange_through_the_own_communicator(cuda):
    """FFVC_DP_NATIVE=1 + FFVC_DP_FORCE=1: the DistributedOptimizer's slices go through ffvc_allreduce_bucket on the exchange
    stream (one rank; same overlap criterion as with torch.distributed's process group)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFVC_DP_FORCE="1", FFVC_DP_NATIVE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dp_overlap.py"), "mlp_mixer", "4"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    summary = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert summary["dp"]["bucket_exchange"].startswith("ffvc_allreduce_bucket") and summary["frac_bytes_early"] > 0.85



def test_bare_bench_command_launches_its_own_ranks(cuda):
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE starts its two ranks itself (bench.launch_ranks:
    a child `torch.distributed.run`, the parent never touches the GPU) and relays rank 0's JSON line as its last stdout line.
    Two ranks share this box's one GPU (gloo exchange, FFVC_SHARE_DEVICE=1) — the launch contract is what is under test."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FFVC_DP_BACKEND="gloo", FFVC_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--depth", "2", "--dim", "128", "--cutn", "2", "--no-cpu-baseline", "--no-alt-dtype"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4
    assert line["config"]["dp"]["ranks"] == 2 and line["value"] > 0
