"""world_size-2 gloo test (CPU) of the multi-GPU plumbing used by bench.py: frame sharding, barrier,
max-over-ranks timing and the host-side result gather.  The data path itself has no collective."""
import os
import socket

import torch
import torch.multiprocessing as mp

import sassd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sassd import dist as D
    r, lr, w = D.init("gloo")
    assert (r, w) == (rank, world)
    mine = D.frame_shard(11, r, w)
    D.barrier()
    tmax = D.allreduce_max(1.0 + rank)             # rank 1 is "slower"
    total = D.allreduce_sum(len(mine))
    allres = D.gather_results([(i, "frame%d" % i) for i in mine])
    q.put((rank, mine, tmax, total, allres))
    D.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_frame_sharding_and_timing_reduction():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [o[1] for o in outs]
    assert sorted(shards[0] + shards[1]) == list(range(11)) and not set(shards[0]) & set(shards[1])
    assert abs(len(shards[0]) - len(shards[1])) <= 1
    for o in outs:
        assert o[2] == 2.0            # max over ranks
        assert o[3] == 11.0           # all frames accounted for
        gathered = sorted(x for part in o[4] for x in part)
        assert [g[0] for g in gathered] == list(range(11))
