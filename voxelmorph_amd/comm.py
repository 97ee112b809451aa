"""ctypes binding of libvxm_comm.so (include/vxm_comm.h): the gradient all-reduce of the data-parallel step as a direct
RCCL call on the step's HIP stream, without torch.distributed on the data path.

This is the exchange of every multi-rank HIP job (`voxelmorph_amd.dist.native_comm()`; `VXM_COMM=torch` opts out to
`torch.distributed.all_reduce` on the 'nccl' backend, which is the same RCCL underneath).  torch.distributed (any
backend, gloo included) is only the rendezvous that carries rank 0's 128-byte unique id to the other ranks.
"""
import ctypes
import os

import torch  # noqa: F401  (before the CDLL: librccl / libamdhip64 resolve to the copies torch has loaded)

from ._lib import VxmHipError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvxm_comm.so")
UNIQUE_ID_BYTES = 128

_c = ctypes
SIGNATURES = {
    "vxm_comm_last_error_string": [],
    "vxm_comm_unique_id": [_c.c_void_p],
    "vxm_comm_init": [_c.c_int, _c.c_int, _c.c_void_p],
    "vxm_comm_world": [],
    "vxm_comm_rccl_version": [],
    "vxm_allreduce_sum_f32": [_c.c_void_p, _c.c_int64, _c.c_void_p],
    "vxm_broadcast_f32": [_c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p],
    "vxm_comm_destroy": [],
    "vxm_comm_abort": [],
}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VxmHipError("libvxm_comm.so not found at %s -- build it with voxelmorph_amd/csrc/build.sh" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = _c.c_char_p if name == "vxm_comm_last_error_string" else _c.c_int
        _lib = h
    return _lib


def rccl_version():
    return int(lib().vxm_comm_rccl_version())


def _call(name, *args):
    h = lib()
    status = getattr(h, name)(*args)
    if status != 0:
        raise VxmHipError("%s failed (status %d): %s" % (name, status, h.vxm_comm_last_error_string().decode()))


class NativeComm:
    """One RCCL communicator per process, bound to the current HIP device."""

    def __init__(self, rank, world, unique_id):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % UNIQUE_ID_BYTES)
        self.rank, self.world = rank, world
        buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        _call("vxm_comm_init", rank, world, buf)

    @staticmethod
    def new_unique_id():
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        _call("vxm_comm_unique_id", buf)
        return buf.raw

    @classmethod
    def from_torch_dist(cls, group=None):
        """Rendezvous over an initialised torch.distributed group (rank 0's id is broadcast as an object)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(rank, world, box[0])

    last_failure = ""          # why the last try_from_torch_dist() returned None (bench.py puts it into its `comm` object)

    @classmethod
    def try_from_torch_dist(cls, group=None):
        """`from_torch_dist`, agreed on by all ranks: if the communicator cannot be created on ANY rank (e.g. two ranks on
        one GPU, which RCCL refuses), every rank drops it and returns None, so that the job continues on
        torch.distributed's RCCL instead of hanging in a half-initialised collective."""
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)

        def agree(ok):
            flags = [None] * world
            dist.all_gather_object(flags, bool(ok), group=group)
            return flags

        def report(flags, err, what):
            cls.last_failure = "libvxm_comm.so %s on %d of %d ranks (%s)" % (what, flags.count(False), len(flags), err or "see other ranks")
            if rank == 0:
                import sys
                print("voxelmorph_amd: libvxm_comm.so %s on %d of %d ranks (%s); gradient all-reduce through "
                      "torch.distributed (same RCCL)" % (what, flags.count(False), len(flags), err or "see other ranks"), file=sys.stderr)

        # round 1: can every rank load the library at all?  Agreed on BEFORE any rank enters the unique-id broadcast: a rank
        # that failed here must not leave the others waiting in a collective it never joins.
        err = ""
        try:
            lib()
        except (VxmHipError, OSError, AttributeError) as exc:
            err = str(exc)
        flags = agree(not err)
        if not all(flags):
            report(flags, err, "cannot be loaded")
            return None
        # round 2: rank 0's unique id travels as an (id | error) object, so a failure to create it is seen by every rank
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.new_unique_id()
            except VxmHipError as exc:
                box[0] = exc
        dist.broadcast_object_list(box, src=0, group=group)
        if not isinstance(box[0], (bytes, bytearray)):
            report([False] * world, str(box[0]), "unique id unavailable")
            return None
        # round 3: the communicator itself (ncclCommInitRank is collective inside RCCL; it fails on all ranks or on none in
        # the cases seen -- two ranks on one device --, and the agreement below covers a split decision)
        comm = None
        try:
            comm = cls(rank, world, box[0])
        except (VxmHipError, OSError) as exc:
            err = str(exc)
        flags = agree(comm is not None)
        if not all(flags):
            if comm is not None:
                comm.destroy()
            report(flags, err, "communicator unavailable")
            return None
        # round 4: one small all-reduce with a known answer (rank r contributes r + 1), so that a communicator that was built but does
        # not exchange correctly is dropped here, by every rank, and not found out by a diverging training run
        ok = False
        try:
            probe = torch.full((1024,), float(rank + 1), device="cuda")
            comm.all_reduce_sum(probe)
            # BOUNDED wait: if a peer failed before entering this collective (its launch raised) it will never complete here, and an
            # unbounded synchronize would keep this rank out of the agreement below for ever -- poll an event, abort on expiry
            done = torch.cuda.Event()
            done.record()
            import time
            deadline = time.monotonic() + float(os.environ.get("VXM_COMM_CHECK_TIMEOUT_S", "60"))
            while not done.query() and time.monotonic() < deadline:
                time.sleep(0.002)
            if not done.query():
                err = "self-check all-reduce did not complete within the time limit (a peer never entered it)"
                comm.abort()
            else:
                ok = bool((probe == world * (world + 1) / 2.0).all())
                if not ok:
                    err = "self-check all-reduce returned %r, expected %r" % (float(probe[0]), world * (world + 1) / 2.0)
        except (VxmHipError, OSError, RuntimeError) as exc:
            err = str(exc)
        flags = agree(ok)
        if all(flags):
            return comm
        comm.destroy()                                  # (no-op after an abort)
        report(flags, err, "self-check failed")
        return None

    def _check(self, t):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise VxmHipError("NativeComm: contiguous fp32 HIP tensors only, got %s %s" % (t.device, t.dtype))

    def all_reduce_sum(self, t):
        self._check(t)
        _call("vxm_allreduce_sum_f32", t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)

    def broadcast(self, t, root=0):
        self._check(t)
        _call("vxm_broadcast_f32", t.data_ptr(), t.numel(), root, torch.cuda.current_stream().cuda_stream)

    def destroy(self):
        _call("vxm_comm_destroy")

    def abort(self):
        """ncclCommAbort: give up a communicator whose collective cannot complete (a peer failed before entering it)"""
        _call("vxm_comm_abort")
