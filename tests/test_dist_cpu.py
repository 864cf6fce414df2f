"""world_size-2 gloo test (CPU) of the multi-GPU plumbing used by bench.py: frame sharding, barrier,
max-over-ranks timing and the host-side result gather.  The data path itself has no collective."""
import os
import socket

import numpy as np

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
    assert D.allgather_float(10.0 + rank) == [10.0, 11.0]
    allres = D.gather_results([(i, "frame%d" % i) for i in mine])
    import types
    from sassd import loader as L
    ds = types.SimpleNamespace(flag=np.ones(13, dtype=np.uint8), test_mode=False)
    ld = L.build_dataloader(ds, 2, 1, dist=True)                   # rank / world come from the process group
    assert (ld.sampler.rank, ld.sampler.num_replicas, ld.batch_size) == (rank, world, 2)
    ld.sampler.set_epoch(3)
    import test_runner_cpu as TR
    from sassd import runner
    annos = runner.single_test(TR._Model(), TR._DS(5))             # this rank's frames + the other rank's, dataset order
    order = [int(a['image_idx'][0]) if len(a['name']) else -1 for a in annos]
    q.put((rank, mine, tmax, total, allres, list(ld.sampler), order))
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
    a, b = outs[0][5], outs[1][5]                                  # the training loader's per-rank epoch shares
    assert len(a) == len(b) == 8 and set(a + b) == set(range(13))
    assert outs[0][6] == outs[1][6] == [0, 1, -1, 3, 4]            # single_test: every rank holds the merged result list


def _grad_sync_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import sassd  # noqa: F401
    from sassd import dist as D, train
    D.init("gloo")
    torch.manual_seed(100 + rank)                                  # different initial weights per rank
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3, bias=False))
    flat = train.FlatParams(m)
    from sassd import kernels as K
    ran = []
    flat.pack_plan = type("Plan", (), {"run": lambda self: ran.append(flat.data.clone())})()
    gen0 = K.weights_generation()
    sync = train.GradSync(flat, buckets=3)                         # broadcast from rank 0; bucketed, overlapped exchange
    # the broadcast wrote every parameter through the flat buffer: packed weight images are invalidated and the
    # optimizer's PackPlan re-runs on the BROADCAST weights (ranks != 0 kept pre-broadcast packs)
    assert K.weights_generation() > gen0 and len(ran) == 1 and torch.equal(ran[0], flat.data)
    assert len(sync.ranges) >= 2 and sync.overlap
    assert sync.ranges[0][2] == 0 and sync.ranges[-1][3] == flat.numel
    assert all(a[3] == b[2] for a, b in zip(sync.ranges, sync.ranges[1:]))       # the ranges tile the flat buffer
    w0 = flat.data.clone()
    x = torch.full((4, 6), float(rank + 1)) + torch.arange(24.).view(4, 6) * 0.01
    reds = []
    for it in range(2):                                            # second iteration: the hook bookkeeping was reset
        flat.zero_grad()
        # no hooks during the reference backward: local gradient first, on a copy of the graph
        m(x).pow(2).sum().backward()
        for w in sync.works:                                       # buckets went out DURING backward
            w.wait()
        launched = list(sync._launched)
        sync.all_reduce_grads()
        reds.append(flat.grad.clone())
    assert any(launched), "no bucket was launched from a gradient hook"
    assert torch.equal(reds[0], reds[1])
    # local gradient for the cross-rank check (hooks removed: nothing is exchanged)
    flat.zero_grad()
    for h in sync._hooks:
        h.remove()
    m(x).pow(2).sum().backward()
    local = flat.grad.clone()
    # a parameter without a gradient on ONE rank must not change the order in which buckets are exchanged
    la, lb, lc = (torch.nn.Linear(4, 4) for _ in range(3))
    flat2 = train.FlatParams(torch.nn.Sequential(la, lb, lc))
    sync2 = train.GradSync(flat2, buckets=3)
    assert len(sync2.ranges) == 3 and sync2.order == [2, 1, 0]
    x2 = torch.arange(8.).view(2, 4) * (rank + 1)
    sync2.reset()
    flat2.zero_grad()
    (lc(la(x2)) if rank == 0 else lc(lb(la(x2)))).sum().backward()      # rank 0 never touches lb
    sync2.all_reduce_grads()
    red2 = flat2.grad.clone()
    flat2.zero_grad()
    for h in sync2._hooks:
        h.remove()
    (lc(la(x2)) if rank == 0 else lc(lb(la(x2)))).sum().backward()
    q.put((rank, w0.numpy(), local.numpy(), reds[0].numpy(), flat2.grad.clone().numpy(), red2.numpy()))
    D.barrier()


def test_gradient_all_reduce_world2():
    """DDP exchange of the training path over gloo, world size 2: one broadcast of the flat parameter buffer, one
    all-reduce (sum) of the flat gradient buffer (the 1/world mean is applied inside the update kernel)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_grad_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = dict()
    for _ in range(2):
        r, w0, local, red, local2, red2 = q.get(timeout=180)
        got[r] = (w0, local, red, local2, red2)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][0], got[1][0])                    # identical parameters after the broadcast
    assert not np.array_equal(got[0][1], got[1][1])                # different shards -> different local gradients
    assert np.allclose(got[0][2], got[0][1] + got[1][1], rtol=1e-6, atol=1e-6)
    assert np.array_equal(got[0][2], got[1][2])
    # bucket order is fixed: rank 0's missing gradient (zeros) + rank 1's, bucket by bucket
    assert np.abs(got[0][3][20:40]).max() == 0 and np.abs(got[1][3][20:40]).max() > 0
    assert np.allclose(got[0][4], got[0][3] + got[1][3], rtol=1e-6, atol=1e-6) and np.array_equal(got[0][4], got[1][4])
