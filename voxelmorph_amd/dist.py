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
    if os.environ.get("VXM_SHARE_DEVICE", "") == "1" and torch.cuda.is_available():
        # several ranks on the devices this box has (the multi-rank path exercised on a 1-GPU box: tests/test_gpu_graph.py); RCCL refuses two
        # ranks on one device, so such a job names its transport itself (VXM_DIST_BACKEND=gloo: the bucket is staged through the host)
        local = local % torch.cuda.device_count()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("VXM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, local, world


def self_launch(nproc, script, argv, visible_devices=None):
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node nproc ... script argv` (one rank per
    GPU of this node, rendezvous on 127.0.0.1 and a free port) when it was started WITHOUT a torchrun environment but asked
    for nproc > 1 ranks (`bench.py --gpus N`, `train.py --gpu 0,1,..`).  Returns normally when nothing has to be done."""
    if nproc <= 1 or "RANK" in os.environ or "LOCAL_RANK" in os.environ:
        return
    import socket
    import sys
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL across processes needs it on this driver
    if visible_devices:
        env["HIP_VISIBLE_DEVICES"] = visible_devices
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execvpe(cmd[0], cmd, env)


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
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def native_comm(required=False):
    """The gradient exchange of a multi-rank run: a `libvxm_comm.so` communicator (include/vxm_comm.h — RCCL behind the C ABI,
    SURVEY.md §8b) whenever the job has more than one rank on HIP devices; torch.distributed is then only the rendezvous
    for the 128-byte unique id.  `VXM_COMM=torch` keeps `torch.distributed.all_reduce` (the same RCCL through torch's 'nccl'
    backend; also what the gloo CPU tests use); `VXM_COMM=rccl` forces the native communicator for a one-rank job too.
    required: raise (on every rank) instead of returning None when the communicator cannot be built -- bench.py asks for that, so a
    scaling run never silently measures a different exchange than the one it reports."""
    mode = os.environ.get("VXM_COMM", "")
    if mode == "torch" or not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend() != "nccl" or (dist.get_world_size() == 1 and mode != "rccl"):
        return None
    from .comm import NativeComm
    comm = NativeComm.try_from_torch_dist()
    if comm is None and required:
        # every rank took the same decision (try_from_torch_dist agrees on it), so every rank raises: no half-dead job
        raise RuntimeError("voxelmorph_amd: the libvxm_comm.so RCCL communicator could not be built for this %d-rank job (reason on rank 0's "
                           "stderr); set VXM_COMM=torch to run the gradient all-reduce through torch.distributed instead"
                           % dist.get_world_size())
    return comm
