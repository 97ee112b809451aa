"""Flat-buffer Adam for the VxmDense path.

The reference trains with `torch.optim.Adam(model.parameters(), lr)` (scripts/torch/train.py:161)
over 24 small tensors and, multi-GPU, `torch.nn.DataParallel` (train.py:151-154).  Here all 327,331
parameters live in ONE contiguous fp32 buffer and so do their gradients: the gradient buffer is the
single RCCL all-reduce bucket (1.31 MB) and one `vxm_adam_step` launch updates everything, with the
1/world_size averaging folded in.  Update rule and defaults match torch.optim.Adam.
"""
import torch

from ._lib import call, ptr, require_device, stream


class FlatAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, process_group=None, direct_grads=True, comm=None, capturable=True):
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: model has no trainable parameters")
        dev = self.params[0].device
        self.n = sum(p.numel() for p in self.params)
        self.flat_param = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat_grad)
        self.exp_avg_sq = torch.zeros_like(self.flat_grad)
        self.lr, self.betas, self.eps = lr, betas, eps
        # capturable (torch.optim.Adam's flag of the same name): the step counter and the bias-correction factors live in device memory
        # (vxm_adam_step_dev), so the update can sit inside a replayed hipGraph (voxelmorph_amd/graph.py); False keeps them on the host
        self.capturable = bool(capturable) and self.flat_param.is_cuda
        self._host_steps = 0
        self.dev_state = torch.zeros(2, dtype=torch.int64, device=dev) if self.capturable else None     # VXM_ADAM_STATE_BYTES
        self.group = process_group
        self.world = 1
        self.comm = comm                       # voxelmorph_amd.comm.NativeComm: direct RCCL calls instead of torch.distributed
        if comm is not None:
            self.world = comm.world
        elif torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        off = 0
        self._views = []
        for p in self.params:
            n = p.numel()
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view(p.shape)
            gview = self.flat_grad[off:off + n].view(p.shape)
            self._views.append(gview)
            # the fused U-Net backward writes parameter gradients straight into the bucket
            p._vxm_grad_sink = gview if direct_grads else None
            p._vxm_sink_written = False
            off += n
        self.direct = direct_grads

    @property
    def step_count(self):
        """optimiser steps taken so far (capturable: read back from the device counter, a synchronising copy -- not for the hot loop)"""
        return int(self.dev_state[0].item()) if self.capturable else self._host_steps

    @step_count.setter
    def step_count(self, k):
        if self.capturable:
            self.dev_state.zero_()
            self.dev_state[0] = int(k)
        else:
            self._host_steps = int(k)

    def broadcast_params(self, src=0):
        """One-off: make every rank start from rank `src`'s weights (replaces DataParallel's per-step
        broadcast_coalesced)."""
        if self.comm is not None:
            self.comm.broadcast(self.flat_param, src)
        elif self.world > 1:
            if self.flat_param.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
                host = self.flat_param.cpu()
                torch.distributed.broadcast(host, src, group=self.group)
                self.flat_param.copy_(host)
            else:
                torch.distributed.broadcast(self.flat_param, src, group=self.group)
        self._params_rewritten()

    def _params_rewritten(self):
        """The flat buffer was written behind autograd's back (collective / kernel): bump the parameters' version counters (a
        graph recorded before must fail loudly) and invalidate the packed bf16 operators cached per parameter."""
        torch.autograd.graph.increment_version(tuple(self.params))
        from .torch.functional_bf16 import invalidate_packs
        invalidate_packs(self.params)

    def zero_grad(self):
        if self.flat_grad.is_cuda:          # a memset on the stream (a memset node in a captured step), not an ATen fill kernel
            call("vxm_fill_zero", ptr(self.flat_grad), self.flat_grad.numel() * 4, stream())
        else:
            self.flat_grad.zero_()
        self._stale = False
        for p in self.params:
            p.grad = None
            p._vxm_sink_written = False

    def load_grads_from_params(self):
        """Fold autograd-produced `p.grad` tensors into the flat bucket.  The fused 3-D engine (UnetFn.backward) writes its
        parameter gradients straight into the bucket; every other producer — the per-op 2-D network (planar.conv2d ->
        ConvFn), a standalone ConvBlock, any torch op on a parameter — hands them to autograd, which accumulates them in
        `p.grad`.  Both contributions are summed here; `p.grad` is then released so a second call cannot add it twice."""
        for p, g in zip(self.params, self._views):
            if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                g.add_(p.grad)
                p.grad = None

    def all_reduce_sum(self, t):
        """SUM all-reduce of `t` in place over the job's ranks through the exchange this optimiser was built on: the libvxm_comm.so RCCL
        communicator, torch.distributed's RCCL ('nccl'), or -- ranks that share one device in the 1-GPU tests, CPU tensors in the CPU tests --
        gloo (device tensors staged through the host: gloo is a rendezvous-only backend here)."""
        if self.comm is not None:
            self.comm.all_reduce_sum(t)
        elif self.world > 1:
            if t.is_cuda and torch.distributed.get_backend(self.group) == "gloo":
                host = t.cpu()
                torch.distributed.all_reduce(host, op=torch.distributed.ReduceOp.SUM, group=self.group)
                t.copy_(host)
            else:
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM, group=self.group)

    def reduce_grads(self):
        """The only data-path collective: SUM all-reduce of the flat gradient bucket (RCCL over xGMI when
        the backend is 'nccl'; gloo in the CPU tests).  The 1/world average is applied by the Adam kernel."""
        self.all_reduce_sum(self.flat_grad)

    def step(self):
        if getattr(self, "_stale", False):
            # the bucket still holds the previous step's (already all-reduced) gradients: a loop that calls model.zero_grad()
            # or nothing at all would silently add the new gradients on top of them
            raise RuntimeError("FlatAdam.step(): no FlatAdam.zero_grad() since the previous step -- the flat gradient bucket is "
                               "only cleared (and the fused backward only allowed to write it again) by the optimiser's own "
                               "zero_grad(); model.zero_grad() is not sufficient")
        self.load_grads_from_params()
        self.reduce_grads()
        require_device(self.flat_param)            # the update itself is a HIP kernel: no CPU fallback
        if self.capturable:
            call("vxm_adam_step_dev", ptr(self.flat_param), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n,
                 float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), ptr(self.dev_state), 1.0 / self.world, stream())
        else:
            self._host_steps += 1
            call("vxm_adam_step", ptr(self.flat_param), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), self.n,
                 float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), self._host_steps,
                 1.0 / self.world, stream())
        # the kernel wrote the parameters behind autograd's back: bump their version counters so that a backward pass over
        # a graph recorded BEFORE this step fails loudly (UnetFn checks them) instead of using the new weights
        torch.autograd.graph.increment_version(tuple(self.params))
        self._stale = True
