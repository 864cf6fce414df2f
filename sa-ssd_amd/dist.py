"""Multi-GPU plumbing for the hot path: one process per GPU, frames sharded across ranks, no data-path collective
(inference is embarrassingly parallel over frames; the reference runs `single_test` serially,
tools/test.py:29-32).  torch.distributed is used only for the start/stop barrier and the max-over-ranks timing
reduction; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests)."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, force_single=False):
    """Initialise the default process group from the torchrun environment (no-op for world size 1 unless
    `force_single`: a one-rank communicator, so that the collectives of the training loop -- RCCL all-reduces launched
    from gradient hooks on the HIP stream -- execute on a box with a single GPU; bench.py --force-ddp, tests)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force_single) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            port = 29500
            if world == 1:                                   # nobody to agree with: any free port
                import socket
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    port = sock.getsockname()[1]
            os.environ["MASTER_PORT"] = str(port)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def allreduce_max(value, device="cpu"):
    """max over ranks of a python float."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_sum(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def allgather_float(value, device="cpu"):
    """[value of rank 0, value of rank 1, ...] of a python float (per-rank step times in bench.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def frame_shard(num_frames, rank, world):
    """Round-robin frame indices of this rank: disjoint, complete, balanced to within one frame."""
    return list(range(rank, num_frames, world))


def gather_results(obj):
    """Host-side gather of per-rank result lists to rank 0 (like collecting result dicts after single_test)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
