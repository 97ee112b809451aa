"""One-process-per-GPU data parallelism for the VxmDense training step.

Replaces `torch.nn.DataParallel` (scripts/torch/train.py:151-154: single process, per-step parameter
broadcast + output gather + gradient reduce to GPU 0) with: every rank owns one MI355X and an equal
shard of the global batch (train.py:128-129 requires batch % n_gpus == 0), computes its local mean
loss, and the only exchange is ONE RCCL all-reduce of the flat 1.31 MB gradient bucket over xGMI
(`FlatAdam.step`).  Equal shards make the mean of per-rank means equal the reference's
loss-on-the-gathered-batch.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def shard_range(global_batch, rank, world):
    """[start, stop) of this rank's samples; the reference's divisibility assert (train.py:128-129)."""
    assert global_batch % world == 0, \
        'Batch size (%d) should be a multiple of the nr of gpus (%d)' % (global_batch, world)
    per = global_batch // world
    return rank * per, (rank + 1) * per


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def native_comm():
    """`VXM_COMM=rccl`: a libvxm_comm.so communicator for the gradient all-reduce (None otherwise: torch.distributed's
    'nccl' backend, the same RCCL, does it).  Needs an initialised process group for the rendezvous."""
    if os.environ.get("VXM_COMM", "") != "rccl" or not (dist.is_available() and dist.is_initialized()):
        return None
    from .comm import NativeComm
    return NativeComm.from_torch_dist()
