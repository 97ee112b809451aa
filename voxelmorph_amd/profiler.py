"""Live per-kernel timing with HIP events on the launch stream (bench.py roofline leg).

`functional.py` brackets each C-ABI launch with `region(name, flops=, bytes=)` when a profiler is
installed; events are recorded on torch's current stream — the stream the kernels are launched on —
and resolved after the timed region, so the timed loop itself is not synchronised.
"""
import contextlib

import torch

ACTIVE = None


class KernelTimer:
    def __init__(self):
        self.pending = []        # (name, start_event, end_event, flops, bytes)
        self.stats = {}

    @contextlib.contextmanager
    def region(self, name, flops=0.0, nbytes=0.0, nominal=0.0):
        """flops = FLOPs the launch executes; nominal = FLOPs of the reference formulation it replaces (differs when the
        kernel evaluates an upsampled segment through collapsed weights: 8 instead of 27 taps)."""
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        try:
            yield
        finally:
            e.record()
            self.pending.append((name, s, e, flops, nbytes, nominal or flops))

    def resolve(self):
        torch.cuda.synchronize()
        for name, s, e, flops, nbytes, nominal in self.pending:
            st = self.stats.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, nominal=0.0))
            st["launches"] += 1
            st["ms"] += s.elapsed_time(e)
            st["flops"] += flops
            st["bytes"] += nbytes
            st["nominal"] += nominal
        self.pending = []
        return self.stats


@contextlib.contextmanager
def region(name, flops=0.0, nbytes=0.0, nominal=0.0):
    if ACTIVE is None:
        yield
    else:
        with ACTIVE.region(name, flops, nbytes, nominal):
            yield


def install(timer):
    global ACTIVE
    ACTIVE = timer


def uninstall():
    global ACTIVE
    ACTIVE = None
