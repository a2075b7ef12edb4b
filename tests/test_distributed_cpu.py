"""CPU, world_size 2 over gloo: the host logic of the N>1 path -- jet sharding and the
single flat-bucket gradient all-reduce + global-norm clip (no kernels involved)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG  # noqa: F401  (puts the package on sys.path)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pipeline.experiment import FlatGrads, shard_range
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    frozen = torch.nn.Parameter(torch.ones(4))            # stands for the never-touched embedding (Q1)
    params = list(net.parameters()) + [frozen]
    flat = FlatGrads(params)
    flat.zero()
    x = torch.full((4, 5), float(rank + 1))
    net(x).sum().backward()                               # autograd accumulates INTO the flat views
    assert all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in params)
    local = flat.flat.clone()
    flat.all_reduce_mean(dist, world)
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok_mean = torch.allclose(flat.flat, want)
    ok_frozen = float(frozen.grad.abs().sum()) == 0.0
    norm = flat.clip_(1.0)
    ok_clip = abs(float(torch.linalg.vector_norm(flat.flat)) - min(1.0, float(norm))) < 1e-4
    lo, hi = shard_range(1001, rank, world)
    q.put((rank, ok_mean, ok_frozen, ok_clip, lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] and r[3] for r in res)
    assert res[0][4] == 0 and res[0][5] == res[1][4] and res[1][5] == 1001


def test_shard_range_covers_everything():
    from pipeline.experiment import shard_range
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
