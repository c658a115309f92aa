"""N>1 path on CPU: two gloo ranks shard the image stream like bench.py (image i -> rank i mod N, no
data-path collective) and agree on the max-over-ranks time."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = bench.shard_seeds(rank, world, batch=8)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, seeds, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    s0, s1 = got[0][1], got[1][1]
    assert set(s0).isdisjoint(s1) and sorted(s0 + s1) == list(range(16))   # a partition of the stream
    assert all(i % 2 == 0 for i in s0) and all(i % 2 == 1 for i in s1)
    assert got[0][2] == got[1][2] == 2.0                                     # max over ranks


def test_single_rank_is_the_whole_stream():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.shard_seeds(0, 1, batch=64) == list(range(64))
