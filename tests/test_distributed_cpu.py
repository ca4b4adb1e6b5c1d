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
                                   wire_dtype=wire)
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
