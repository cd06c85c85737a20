"""CPU, world_size 2 over gloo: the only cross-rank logic on this path -- sharding of entries by index and the
max-time / sum-of-units aggregation bench.py uses (the data path itself has no collective)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    first = bench.shard_indices(rank, 1000)
    # rank r "measures" (r+1) seconds for 1000 MiB
    t, u = bench.aggregate(float(rank + 1), 1000.0 * (1 << 20), dist, torch.device("cpu"))
    q.put((rank, first, t, u))
    dist.destroy_process_group()


def test_shard_and_aggregate_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0, 1000]                       # disjoint corpus index ranges
    for _, _, t, u in res:
        assert t == 2.0 and u == 2000.0 * (1 << 20)               # max over ranks, sum over ranks


def test_shards_cover_without_overlap():
    import bench
    n, world = 10000, 8
    spans = [(bench.shard_indices(r, n), bench.shard_indices(r, n) + n) for r in range(world)]
    for a, b in zip(spans, spans[1:]):
        assert a[1] == b[0]
