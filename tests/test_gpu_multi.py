"""-m gpu multi-device checks: the real RCCL path with world_size = torch.cuda.device_count() (skipped on a 1-GPU box)
and the same exchange on a one-rank communicator (always runs)
-- gradient exchange of the training loop, and bench.py's own `--gpus N` launcher."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def test_rccl_exchange_on_one_rank(dev):
    """Not skipped on a 1-GPU box: the bucketed, hook-launched gradient all-reduce of the training step on a ONE-RANK
    nccl (RCCL) communicator -- every collective of the DDP path executes on the HIP stream (tests/dist_gpu_single.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_single.py")], capture_output=True, text=True,
                         env=env, timeout=600)
    assert out.returncode == 0 and "RCCL_SINGLE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_rccl_gradient_exchange(dev):
    n = _ngpu()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (the driver's multi-GPU node)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
                          "--master-addr", "127.0.0.1", "--master-port", "29611",
                          os.path.join(ROOT, "tests", "dist_gpu_worker.py")], capture_output=True, text=True, env=env,
                         timeout=600)
    assert out.returncode == 0 and "RCCL_OK world=%d" % n in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_bench_spawns_its_own_ranks(dev):
    """`python bench.py --gpus N` without a torchrun environment must become N ranks and report n_gpus = N."""
    n = _ngpu()
    if n < 2:
        pytest.skip("needs >= 2 GPUs (the driver's multi-GPU node)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "10", "--warmup",
                          "3", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == n and d["value"] > 0
