"""Bounded run-ahead of a training loop: at most `depth` steps enqueued on the GPU.

The step functions of this package never synchronise, so a host that needs 2.5 ms to enqueue a 17 ms step runs arbitrarily far ahead.
That is harmless for the GPU and bad for the caching allocator: blocks a second stream still holds (weight-gradient launches,
`record_stream`) cannot be reused until their events have passed, so the deeper the run-ahead, the more blocks are in flight and the
more `hipMalloc`s (10-15 ms each) land in the middle of a loop that looked warmed up (measured in bench.py: DESIGN.md section 4.3).
Two steps in flight keep the GPU fed and the allocator in its steady state after a handful of steps.
"""
import time

import torch


class InFlight:
    def __init__(self, depth=2):
        self.events = [None] * depth
        self.k = 0
        self.wait_s = 0.0            # host time spent waiting (bench.py subtracts it from its host-side figure)

    def wait(self):
        """call before enqueuing a step: blocks until the step enqueued `depth` steps ago has finished"""
        ev = self.events[self.k % len(self.events)]
        if ev is not None:
            t0 = time.perf_counter()
            ev.synchronize()
            self.wait_s += time.perf_counter() - t0

    def mark(self):
        """call after enqueuing a step (on the step's stream)"""
        if torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record()
            self.events[self.k % len(self.events)] = ev
        self.k += 1
