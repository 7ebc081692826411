"""world_size=2 gloo (CPU) test of the data-parallel step plumbing: shard ranges, flat gradient bucket,
one all-reduce, identical parameters on both ranks afterwards and equality with the single-process global-batch gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from situation_recognition_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    frozen = torch.nn.Linear(3, 3)
    for p in frozen.parameters():
        p.requires_grad = False
    params = list(net.parameters()) + list(frozen.parameters())
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(10, 8, generator=g), torch.randn(10, 4, generator=g)
    lo, hi = parallel.shard_range(10, rank, world)
    bucket = parallel.GradBucket(params)
    assert bucket.flat.numel() == sum(p.numel() for p in net.parameters())
    loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
    loss.backward()
    bucket.reduce()
    gn = torch.nn.utils.clip_grad_norm_([p for p in params if p.requires_grad], 1.0)
    torch.optim.Adamax(net.parameters(), lr=0.01).step()
    parallel.barrier()
    out[rank] = (float(gn), [p.detach().clone() for p in net.parameters()])
    dist.destroy_process_group()


def test_two_rank_step_matches_global_batch():
    from situation_recognition_amd import parallel
    assert [parallel.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    assert parallel.shard_range(6144, 7, 8) == (5376, 6144)
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    (g0, p0), (g1, p1) = out[0], out[1]
    assert abs(g0 - g1) < 1e-7
    for a, b in zip(p0, p1):
        assert torch.equal(a, b)
    # single process, global batch (equal shard sizes -> mean of shard means == global mean)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(10, 8, generator=g), torch.randn(10, 4, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    torch.optim.Adamax(net.parameters(), lr=0.01).step()
    assert abs(float(gn) - g0) < 1e-5
    for a, b in zip(net.parameters(), p0):
        assert (a - b).abs().max() < 1e-6
