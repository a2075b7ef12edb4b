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
    # the single tensor gather of generate(): shards that differ by one jet, rank 0 gets the concatenation in rank order
    from pipeline.experiment import gather_generated
    from tensorclass import TensorMultiModal
    J = 3 + rank
    g = torch.Generator().manual_seed(rank)
    gen = TensorMultiModal(None, torch.randn(J, 5, 3, generator=g), torch.randint(0, 8, (J, 5, 1), generator=g),
                           torch.randint(0, 2, (J, 5, 1), generator=g))
    got = gather_generated(gen, dist, rank, world)
    ok_gather = True
    if rank == 0:
        want_parts = []
        for r in range(world):
            gr = torch.Generator().manual_seed(r)
            want_parts.append((torch.randn(3 + r, 5, 3, generator=gr), torch.randint(0, 8, (3 + r, 5, 1), generator=gr),
                               torch.randint(0, 2, (3 + r, 5, 1), generator=gr)))
        ok_gather = (torch.equal(got.continuous, torch.cat([w[0] for w in want_parts]))
                     and torch.equal(got.discrete, torch.cat([w[1] for w in want_parts]))
                     and torch.equal(got.mask, torch.cat([w[2] for w in want_parts]))
                     and got.discrete.dtype == torch.int64 and got.mask.dtype == torch.int64)
    else:
        ok_gather = got is None
    q.put((rank, ok_mean, ok_frozen, ok_clip and ok_gather, lo, hi))
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


def test_bench_launcher_forms_the_ranks_it_reports():
    """`bench.py --gpus 2` without a torchrun environment starts its own ranks (a child torch.distributed.run, before
    any GPU call) and the printed n_gpus is the size of the process group that really formed (gloo here)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MMB_DIST_BACKEND="gloo", MMB_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rendezvous-check"], env=env, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"n_gpus": 2, "rendezvous": "ok"}
    # a launcher that formed a different world than --gpus asks for must fail loudly
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--rendezvous-check"], env=env2, capture_output=True,
                       text=True, timeout=240)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_layered_philox_offsets_differ_across_ranks():
    """the layered generation loop's per-step draws must not repeat across ranks."""
    from model.solvers import philox_offset
    seen = set()
    for rank in (0, 1, 7):
        for tag in (1, 2, 3):
            for stream in (1, 2, 0xFFFFF):
                for step in (0, 1, 199):
                    o = philox_offset(rank, tag, stream, step)
                    assert 0 <= o < 2 ** 64 and o not in seen
                    seen.add(o)
