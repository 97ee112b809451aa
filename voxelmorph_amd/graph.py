"""One training step as ONE hipGraph launch.

The eager step of this package is ~250 C-ABI launches driven from Python (autograd nodes, ctypes calls, caching-allocator traffic):
2.5 - 6 ms of host time per step on the bench hosts, and -- at 4 pairs per GPU -- hipMalloc / hipFree calls of the caching allocator
that stall the host for tens of milliseconds when its block reuse pattern slips.  Every launch of the step has
arguments that do not change from step to step (static shapes, the weights and the optimiser state updated in place, the Adam step
counter in device memory: `FlatAdam(capturable=True)`), so the whole step -- zero_grad, forward, losses, backward on both streams,
weight re-packing, Adam -- is captured once into a hipGraph and replayed with one `hipGraphLaunch` per step: no Python per launch, no
allocator call, memory of the step fixed at capture (a private pool of the caching allocator).

    step = GraphedStep(lambda: loss_fn(model(src, trg)), opt)        # fn: forward + loss, returns the loss (or a tuple starting with it)
    for batch in loader:
        src.copy_(batch.src); trg.copy_(batch.trg)                   # inputs are STATIC tensors: refill them in place
        loss = step()                                                # eager for the first `eager_steps` calls, then capture + replay

Every call performs exactly one optimiser step, whichever way it is submitted.  Multi-rank (`opt.world > 1`): the graph holds
zero_grad + forward + backward; the all-reduce of the flat bucket and the Adam launch follow it eagerly (two calls), so the collective
keeps its own stream semantics.  Reference: the loop body of scripts/torch/train.py:194-223.
"""
import torch


class GraphedStep:
    def __init__(self, fn, opt, eager_steps=2, enabled=True, range_guard=None):
        self.fn, self.opt = fn, opt
        self.eager_steps = max(1, int(eager_steps))      # at least one eager step: lazy one-off work (hipFuncSetAttribute, plans) must not be captured
        self.enabled = bool(enabled)
        self.calls = 0
        self.graph = None
        self.out = None
        self._one = None
        self.replays = 0
        self.recaptures = 0
        self.capture_error = None
        # range_guard (default ON since round 6; VXM_RANGE_GUARD=0 or range_guard=False switch it off): the FIRST eager step runs under the
        # dynamic-range probe of the fp16-piece conv engine (voxelmorph_amd/diagnostics.py: one small launch per activation / gradient tensor
        # of the fused U-Net, ~3 ms once) and moves this process to the three-piece engine BEFORE anything is captured when a tensor of the
        # batch has more than 0.1 % of its values in that engine's absolute-error regime.  Steady-state cost: none (the captured step carries no
        # probe).  On ordinary data the report stays far below the limit (noise pairs: 4.5e-4 on the worst tensor); it is the heavy-tailed
        # batch -- an intensity outlier per staged tile -- that trips it.  Multi-rank: every rank judges its own first batch.
        import os
        self.range_guard = (os.environ.get("VXM_RANGE_GUARD", "1") != "0") if range_guard is None else bool(range_guard)
        self.range_report = None

    # the step, eagerly (also what is captured)
    def _body(self, with_update):
        self.opt.zero_grad()
        out = self.fn()
        loss = out[0] if isinstance(out, (tuple, list)) else out
        if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype or self._one.shape != loss.shape:
            self._one = torch.ones_like(loss)            # (eager steps only: loss.backward() would launch this fill on every step)
        loss.backward(self._one)
        if with_update:
            self.opt.step()
        return out

    def eager(self):
        self.calls += 1
        if self.range_guard and self.calls == 1:
            from .diagnostics import guard_engine
            box = []
            self.range_report = guard_engine(lambda: box.append(self._body(True)))
            if box:
                return box[0]
        return self._body(True)

    def _capture(self):
        from .torch.functional_bf16 import invalidate_packs
        if not getattr(self.opt, "capturable", False):
            raise RuntimeError("GraphedStep: the optimiser's step counter must live on the device (FlatAdam(capturable=True))")
        # the packed operators must be rebuilt INSIDE the graph on every replay (the weights change under it): make every cached copy stale
        invalidate_packs(self.opt.params)
        self.single = self.opt.world == 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # Multi-rank: a communicator library's own threads (RCCL's proxy) may touch the HIP runtime while this thread captures; only calls of
        # the capturing thread may invalidate the capture then.  One rank: the strict default.
        mode = "global" if self.single else "thread_local"
        with torch.cuda.graph(g, capture_error_mode=mode):
            out = self._body(self.single)
        self.graph, self.out = g, out
        self._hyper = self._hyper_now()
        # Multi-rank: opt.step() stays outside the graph, so gradients that reach the optimiser through autograd's `p.grad` (the per-op 2-D
        # network, FlatAdam(direct_grads=False), a parameter used twice) are rewritten by every replay in the tensors the capture allocated --
        # but the first eager opt.step() folds p.grad into the bucket and drops the reference.  Keep the captured tensors and hand them back
        # to their parameters after each replay.
        self._captured_grads = []
        if self.single:
            self.opt._stale = True
        else:
            for p, view in zip(self.opt.params, self.opt._views):
                if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                    self._captured_grads.append((p, p.grad))

    def _hyper_now(self):
        """the optimiser's by-value kernel arguments (learning rate, betas, eps): frozen inside a captured Adam launch"""
        return (float(self.opt.lr), tuple(float(b) for b in self.opt.betas), float(self.opt.eps))

    def __call__(self):
        if not self.enabled or self.calls < self.eager_steps:
            return self.eager()
        if self.graph is not None and self.single and self._hyper != self._hyper_now():
            # lr / betas / eps are by-value arguments of the captured Adam launch: a schedule that moved them needs a new capture
            # (the eager and the multi-rank paths read them per step)
            self.graph = None
            self.recaptures += 1
        if self.graph is None:
            try:
                self._capture()
            except Exception as exc:       # a capture the runtime refuses must not take the training down: same kernels, launch by launch
                import warnings
                from .torch.functional_bf16 import invalidate_packs
                warnings.warn("voxelmorph_amd.GraphedStep: hipGraph capture of the training step failed (%s: %s); continuing with "
                              "launch-by-launch submission" % (type(exc).__name__, exc))
                self.graph, self.enabled, self.capture_error = None, False, "%s: %s" % (type(exc).__name__, exc)
                invalidate_packs(self.opt.params)      # operators "packed" inside the failed capture were never executed
                torch.cuda.synchronize()
                return self.eager()
        self.calls += 1
        self.replays += 1
        self.graph.replay()
        if self.single:
            # Adam ran inside the graph: the parameters changed behind autograd's back
            torch.autograd.graph.increment_version(tuple(self.opt.params))
        else:
            self.opt._stale = False
            for p in self.opt.params:          # the bucket was written by the graph: a second eager backward must not overwrite it
                p._vxm_sink_written = True
            for p, g in self._captured_grads:  # gradients the replay left in autograd's tensors: opt.step() folds them into the bucket
                p.grad = g
            self.opt.step()
        return self.out
