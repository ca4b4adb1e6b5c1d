"""World-size-2 tests of the data-parallel layer on CPU (gloo): bucketed asynchronous gradient all-reduce,
parameter broadcast, sampler partitioning.  DP-equivalence: averaged per-rank gradients == single-process
gradients on the concatenated batch (the loss is a mean and Horovod averages: main.py:627,811)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp
from torch import nn


def _collect(q, procs, limit=150.0):
    """Results of all ranks, failing fast when a rank died and after `limit` seconds (a hung rendezvous must not eat
    the GPU budget of the caller)."""
    import queue as _queue
    import time as _time
    res, deadline = [], _time.time() + limit
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f"a worker died with exit code {dead}"
            assert _time.time() < deadline, f"ranks did not report within {limit:.0f} s"
    res.sort(key=lambda t: t[0])
    return res


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(16, 64), nn.Tanh(), nn.Linear(64, 48), nn.Tanh(), nn.Linear(48, 8))


def _worker(rank, world, port, wire, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    hvd.init(backend="gloo")
    assert hvd.size() == world and hvd.rank() == rank
    net = _model()
    if rank != 0:                              # perturb non-root replicas: broadcast must repair them
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    hvd.broadcast_parameters(net, root_rank=0) if False else hvd.broadcast(arena.params, 0)  # flat-bucket broadcast (no shadows on CPU)
    opt = hvd.DistributedOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), arena=arena, bucket_bytes=1024,
                                   wire_dtype=wire, tail_bytes=0)
    assert len(opt.buckets) >= 3               # several buckets -> several overlapped all-reduces
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    sampler = hvd.DistributedSampler(8, shuffle=False)
    idx = list(iter(sampler))
    for _ in range(2):
        opt.zero_grad()
        loss = ((net(X[idx]) - Y[idx]) ** 2).mean()
        loss.backward()
        opt.step()
    (l,) = hvd.allreduce_scalars(loss.detach())
    q.put((rank, idx, arena.params.detach().numpy().copy(), float(l)))   # numpy: pickled by value (no shm fd hand-off)


@pytest.mark.parametrize("wire", [None, torch.bfloat16])
def test_dp_equivalence_world2(wire):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, wire, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, procs)
    finally:
        for p in procs:                      # never leave a rank behind (it would keep the device / the port)
            if p.is_alive():
                p.join(timeout=20)
            if p.is_alive():
                p.terminate()

    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # partition: disjoint, covers the dataset
    assert sorted(res[0][1] + res[1][1]) == list(range(8))
    assert (res[0][2] == res[1][2]).all()                    # replicas stay identical
    # single-process reference on the concatenated batch
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    ref = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    for _ in range(2):
        arena.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        ref.step()
    tol = 1e-6 if wire is None else 2e-3
    assert (torch.from_numpy(res[0][2]) - arena.params.detach()).abs().max().item() < tol


def _tail_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    hvd.init(backend="gloo")
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    opt = hvd.DistributedOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), arena=arena, bucket_bytes=1024, tail_bytes=3000,
                                   tail_bucket_bytes=512, tail_wire_dtype=torch.bfloat16)
    n0, wire0 = len(opt.buckets), list(opt._wire_of)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    idx = list(iter(hvd.DistributedSampler(8, shuffle=False)))
    counts = []
    for _ in range(4):
        opt.zero_grad()
        ((net(X[idx]) - Y[idx]) ** 2).mean().backward()
        opt.step()
        counts.append(len(opt.buckets))
    covered = sorted((s, e) for s, e, _ in opt.buckets)
    q.put((rank, arena.params.detach().numpy().copy(), n0, counts, [str(w) for w in opt._wire_of], covered, arena.total,
           all(w is None for w in wire0)))


def test_dp_tail_slices_are_recut_and_travel_in_bf16_after_the_first_step():
    """The slices that go on the wire last (observed launch order of step 1) are re-cut into small pieces on the bf16 wire from
    step 2 on; the others keep fp32.  The bucket list still tiles the gradient arena, replicas stay identical, and the result is
    the single-process one up to the bf16 rounding of the tail gradients."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    (_, p0, n0, counts, wires, covered, total, all_fp32_first), (_, p1, *_rest) = res
    assert (p0 == p1).all()
    assert all_fp32_first and counts[0] > n0 and counts[1:] == [counts[0]] * 3            # re-cut once, after the first step
    assert "torch.bfloat16" in wires and "None" in wires                                   # tail on bf16, the rest on fp32
    assert covered[0][0] == 0 and covered[-1][1] == total and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    ref = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    for _ in range(4):
        arena.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        ref.step()
    assert (torch.from_numpy(p0) - arena.params.detach()).abs().max().item() < 3e-3


def _tail_default_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    hvd.init(backend="gloo")
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    opt = hvd.DistributedOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), arena=arena, bucket_bytes=1024, tail_bytes=3000,
                                   tail_bucket_bytes=512)                   # tail_wire_dtype left at its default
    n0 = len(opt.buckets)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    idx = list(iter(hvd.DistributedSampler(8, shuffle=False)))
    for _ in range(3):
        opt.zero_grad()
        ((net(X[idx]) - Y[idx]) ** 2).mean().backward()
        opt.step()
    # a rank whose observed launch order differs from rank 0's must not silently re-cut differently
    order = list(range(len(opt.buckets)))
    if rank == 1:
        order[0], order[1] = order[1], order[0]
    try:
        opt._retune_tail(order)
        raised = ""
    except RuntimeError as e:
        raised = str(e)
    q.put((rank, arena.params.detach().numpy().copy(), n0, len(opt.buckets), [str(w) for w in opt._wire_of], raised,
           hvd.describe()["wire"]))


def test_dp_tail_keeps_the_fp32_wire_by_default_and_ranks_must_agree_on_the_order():
    """Default tail policy = re-cut only: nothing narrower than `wire_dtype` goes on the wire unless asked (the reference
    exchanges fp32, main.py:627), so the result equals the single-process one to fp32 round-off.  The re-cut layout is rank 0's
    broadcast order; a rank that observed another order raises (on every rank) instead of enqueueing mismatched all-reduces."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tail_default_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    (_, p0, n0, n1, wires, raised0, wire_desc), (_, p1, _, _, _, raised1, _) = res
    assert (p0 == p1).all() and n1 > n0 and set(wires) == {"None"}
    assert wire_desc["slices"] == "float32" and wire_desc["tail"] == "float32"
    assert "different orders" in raised0 and "different orders" in raised1
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    ref = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    for _ in range(3):
        arena.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        ref.step()
    assert (torch.from_numpy(p0) - arena.params.detach()).abs().max().item() < 1e-6


def test_bare_bench_command_spawns_the_ranks_and_propagates_their_exit_code():
    """`python bench.py --gpus 2` without a launcher starts two ranks as children (bench.launch_ranks).  No GPU here: both ranks
    get through the rendezvous and stop at "needs an MI355X" — the parent relays that and exits non-zero without a JSON line
    (the GPU leg of the same contract: tests/test_distributed_gpu.py::test_bare_bench_command_launches_its_own_ranks)."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU leg of the launch contract")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") >= 1 and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_sampler_matches_torch():
    from torch.utils.data import DistributedSampler as TorchSampler

    from feed_forward_vqgan_clip_amd.distributed import DistributedSampler
    data = list(range(37))
    for epoch in (0, 3):
        for r in range(4):
            a = DistributedSampler(37, num_replicas=4, rank_=r, shuffle=True, seed=0)
            b = TorchSampler(data, num_replicas=4, rank=r, shuffle=True, seed=0)
            a.set_epoch(epoch)
            b.set_epoch(epoch)
            assert list(iter(a)) == list(iter(b))


# ----------------------------------------------------------------------------- sliced buckets, overlapped bucket-wise update
class _FlatSGD:
    """CPU stand-in with FusedAdam's DP surface (`arena`, `grad_scale`, `loss_scale`, `step(ranges=...)`): plain SGD over slices
    of the flat bucket, so the bucket-wise `overlap_update` path of DistributedOptimizer runs under gloo."""

    def __init__(self, arena, lr):
        self.arena, self.lr, self.grad_scale, self.loss_scale = arena, lr, 1.0, 1.0
        self.param_groups = [{"lr": lr}]
        self.ranges_seen = []

    def zero_grad(self):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None, ranges=None):
        a = self.arena
        for s, e in (ranges if ranges is not None else ((0, a.total),)):
            self.ranges_seen.append((s, e))
            a.params[s:e].add_(a.grads[s:e], alpha=-self.lr * self.grad_scale / self.loss_scale)


def _big_model():
    torch.manual_seed(0)
    # the FIRST layer's weight (the last gradient of the backward pass) is larger than two buckets -> cut into slices
    return nn.Sequential(nn.Linear(64, 96), nn.Tanh(), nn.Linear(96, 8))


def _worker_fused(rank, world, port, wire, overlap, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), FFVC_DP_OVERLAP_UPDATE="1" if overlap else "0")
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    hvd.init(backend="gloo")
    net = _big_model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    inner = _FlatSGD(arena, 0.05)
    opt = hvd.DistributedOptimizer(inner, bucket_bytes=8192, wire_dtype=wire, tail_bytes=0)
    w0 = net[0].weight
    slices = [b for b, (_, _, idxs) in enumerate(opt.buckets) if idxs == [arena.plist.index(w0)]]
    assert len(slices) >= 3, opt.buckets                         # 24 KiB tensor, 8 KiB buckets
    assert sum(e - s for s, e, _ in opt.buckets) == arena.total  # the slices tile the bucket exactly
    # attribute forwarding: reads and the loss_scale write go to the wrapped optimizer
    opt.loss_scale = 8.0
    assert inner.loss_scale == 8.0 and opt.loss_scale == 8.0 and opt.lr == 0.05
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 64, generator=g), torch.randn(8, 8, generator=g)
    idx = list(iter(hvd.DistributedSampler(8, shuffle=False)))
    for _ in range(2):
        opt.zero_grad()
        (((net(X[idx]) - Y[idx]) ** 2).mean() * 8.0).backward()          # loss-scaled backward
        opt.step()
    n_ranges = len(inner.ranges_seen)
    # a second backward before step() must be refused explicitly (gradient accumulation / weight sharing)
    opt.zero_grad()
    ((net(X[idx]) - Y[idx]) ** 2).mean().backward()
    err = ""
    try:
        ((net(X[idx]) - Y[idx]) ** 2).mean().backward()
    except RuntimeError as e:
        err = str(e)
    opt.synchronize()
    q.put((rank, arena.params.detach().numpy().copy(), n_ranges, err))


@pytest.mark.parametrize("wire,overlap", [(None, True), (None, False), (torch.bfloat16, True)])
def test_sliced_buckets_and_overlapped_update_match_the_synchronous_path(wire, overlap):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fused, args=(r, 2, port, wire, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, procs)
    finally:
        for p in procs:
            if p.is_alive():
                p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (res[0][1] == res[1][1]).all()                                      # replicas identical
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    net = _big_model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 64, generator=g), torch.randn(8, 8, generator=g)
    for _ in range(2):                                                         # single process, full batch, plain SGD
        arena.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        with torch.no_grad():
            arena.params.add_(arena.grads, alpha=-0.05)
    tol = 1e-6 if wire is None else 2e-3
    assert (torch.from_numpy(res[0][1]) - arena.params.detach()).abs().max().item() < tol
    # overlap_update: one optimizer launch per bucket (in launch order); synchronous path: one launch per step
    assert res[0][2] == (2 * 5 if overlap else 2) or (overlap and res[0][2] > 2), res[0][2]
    assert "second backward()" in res[0][3] and "weight shared" in res[0][3]


# ---------------------------------------------------------------------------------------------------------------------------------
# world 4: deferred (grouped) weight gradients + the rank-agreement check with ONE permuted rank
def _deferred_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from feed_forward_vqgan_clip_amd import distributed as hvd
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    hvd.init(backend="gloo")
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    opt = hvd.DistributedOptimizer(torch.optim.SGD(net.parameters(), lr=0.1), arena=arena, bucket_bytes=1024, tail_bytes=3000,
                                   tail_bucket_bytes=512)
    n0 = len(opt.buckets)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    idx = list(iter(hvd.DistributedSampler(8, shuffle=False)))
    # what ops.WgradGroup does on the GPU: the weight gradients of a group of layers are written by ONE launch after the last member's
    # backward -> autograd's own hooks for them must be ignored (`_ffvc_deferred`), the report comes from the flush, newest layer first
    deferred = [net[0].weight, net[2].weight]
    launch_orders = []
    for _ in range(3):
        opt.zero_grad()
        for p in deferred:
            p._ffvc_deferred = True
        ((net(X[idx]) - Y[idx]) ** 2).mean().backward()
        early = [b for b in opt._handles if any(i in opt.buckets[b][2] for i in (arena._index[id(p)] for p in deferred))]
        for p in deferred:                         # the flush: marks cleared, then the reports in the order backward produces gradients
            p._ffvc_deferred = False
        for p in reversed(deferred):
            opt._param_ready(p)
        launch_orders.append(dict(opt._order))
        opt.step()
    order = list(range(len(opt.buckets)))
    if rank == 2:                                  # ONE rank observed another launch order
        order[0], order[1] = order[1], order[0]
    try:
        opt._retune_tail(order)
        raised = ""
    except RuntimeError as e:
        raised = str(e)
    q.put((rank, arena.params.detach().numpy().copy(), n0, len(opt.buckets), len(early), raised))


def test_world4_deferred_weight_gradients_and_one_permuted_rank():
    """Four ranks (gloo): the buckets that hold a deferred weight gradient go on the wire only after the flush reported it (never from
    autograd's own hook), the replicas stay identical and equal the single-process result; the re-cut layout is rank 0's, and when ONE
    of the four ranks observed a different launch order EVERY rank raises instead of pairing all-reduces wrongly."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_deferred_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    try:
        res = _collect(q, procs, limit=240.0)
    finally:
        for p in procs:
            if p.is_alive():
                p.join(timeout=20)
            if p.is_alive():
                p.terminate()
    p0 = res[0][1]
    for _, pr, n0, n1, early, raised in res:
        assert (pr == p0).all()
        assert early == 0                          # no bucket with a deferred gradient left before its flush
        assert n1 > n0                             # the tail was re-cut after the first step
        assert "different orders" in raised        # all four ranks, although only rank 2 was permuted
    from feed_forward_vqgan_clip_amd.arena import ParamArena
    net = _model()
    arena = ParamArena(net, torch.float32, allow_cpu=True)
    ref = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(1)
    X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)
    for _ in range(3):
        arena.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        ref.step()
    assert (torch.from_numpy(p0) - arena.params.detach()).abs().max().item() < 1e-6
