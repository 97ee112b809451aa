"""On-device data path for the VxmDense training step.

Replaces, for this path only, `voxelmorph/generators.py:9-107` (`volgen`, `scan_to_scan`) together with the
host→device step of `scripts/torch/train.py:199-201` (`torch.from_numpy(d).to(device).float().permute(0,4,1,2,3)`):
the reference builds float64 numpy batches `[B,*vol,1]` on the training thread, copies them pageable and
converts / permutes on the device every step — at tens of pairs/s per GPU that is the bottleneck.

Here each rank owns a `PairLoader`:
  * volumes are loaded once (npz `vol` / npy / preloaded arrays — `py/utils.py:69-129`; NIfTI needs nibabel, which
    this image does not have), converted to fp32 and kept in PINNED host memory; volumes that fit are uploaded
    once and stay resident in HBM (288 GB per GPU: ~10,000 volumes of 160x192x224), so a step's "load" is a
    device-side gather into the batch tensors;
  * otherwise a background thread fills pinned staging batches and a dedicated HIP copy stream uploads batch k+1
    while batch k trains (`non_blocking=True` from pinned memory; an event orders the consumer stream);
  * batches come out already in the layout the kernels take: `[B, 1, D, H, W]` fp32, contiguous.
The tuple contract of `scan_to_scan` is kept: `(invols, outvols)` with `invols = [scan1, scan2]` and
`outvols = [scan2(, scan1 if bidir), zeros]` (the zero flow target is allocated once and reused, as :98-101).
Random pairs are drawn like the reference (`np.random.randint`, :53) from a per-rank `numpy` generator, so ranks
draw different pairs (data parallel shards, `scripts/torch/train.py:128-129`).
"""
import glob
import os
import queue
import threading

import numpy as np
import torch


def load_volfile(filename, np_var='vol'):
    """npz / npy subset of `voxelmorph/py/utils.py:69-129` (preloaded arrays are passed through)."""
    if not isinstance(filename, (str, os.PathLike)):
        return np.asarray(filename)
    filename = str(filename)
    if not os.path.isfile(filename):
        raise ValueError("'%s' is not a file." % filename)
    if filename.endswith('.npy'):
        return np.load(filename)
    if filename.endswith('.npz'):
        npz = np.load(filename)
        return next(iter(npz.values())) if len(npz.keys()) == 1 else npz[np_var]
    if filename.endswith(('.nii', '.nii.gz', '.mgz')):
        raise ValueError("NIfTI / mgz need nibabel, which is not available here; convert '%s' to npz" % filename)
    raise ValueError('unknown filetype for %s' % filename)


def _resolve(vol_names):
    if isinstance(vol_names, str):                       # generators.py:38-42
        if os.path.isdir(vol_names):
            vol_names = os.path.join(vol_names, '*')
        vol_names = sorted(glob.glob(vol_names))
    return list(vol_names)


class PairLoader:
    """Infinite iterator of `(invols, outvols)` device batches for scan-to-scan registration."""

    def __init__(self, vol_names, batch_size=1, bidir=False, prob_same=0, no_warp=False, np_var='vol', device=None, rank=0,
                 seed=0, resident_bytes=64 << 30, prefetch=2):
        names = _resolve(vol_names)
        if not names:
            raise ValueError('no volumes given')
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise ValueError('PairLoader feeds the MI355X path: device must be a HIP device, got %s' % self.device)
        self.batch_size, self.bidir, self.prob_same, self.no_warp = batch_size, bidir, prob_same, no_warp
        self.rng = np.random.default_rng([seed, rank])
        vols = [np.ascontiguousarray(load_volfile(n, np_var), dtype=np.float32) for n in names]
        shape = vols[0].shape
        if any(v.shape != shape for v in vols):
            raise ValueError('all volumes must share one shape (pad them first), got %s' % sorted({v.shape for v in vols}))
        self.shape = tuple(shape)
        self.n = len(vols)
        host = torch.empty((self.n, 1) + self.shape, dtype=torch.float32).pin_memory()
        for i, v in enumerate(vols):
            host[i, 0].copy_(torch.from_numpy(v))
        self.host = host
        self.resident = host.numel() * 4 <= resident_bytes
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.zeros = None
        if self.resident:
            with torch.cuda.stream(self.copy_stream):
                self.dev = host.to(self.device, non_blocking=True)
            self.copy_stream.synchronize()
        else:
            self.q = queue.Queue(maxsize=max(1, prefetch))
            self.thread = threading.Thread(target=self._producer, daemon=True)
            self.thread.start()

    # ---- sampling, generators.py:53,86-95
    def _draw(self):
        i1 = self.rng.integers(self.n, size=self.batch_size)
        i2 = self.rng.integers(self.n, size=self.batch_size)
        if self.prob_same > 0 and self.rng.random() < self.prob_same:
            if self.rng.random() > 0.5:
                i1 = i2
            else:
                i2 = i1
        return i1, i2

    def _producer(self):
        """Streaming mode: gather into pinned staging on this thread, upload on the copy stream."""
        while True:
            i1, i2 = self._draw()
            stage = torch.empty((2, self.batch_size, 1) + self.shape, dtype=torch.float32).pin_memory()
            torch.index_select(self.host, 0, torch.from_numpy(i1), out=stage[0])
            torch.index_select(self.host, 0, torch.from_numpy(i2), out=stage[1])
            with torch.cuda.stream(self.copy_stream):
                dev = stage.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.q.put((dev, ev, stage))          # `stage` stays referenced until the copy has been consumed

    def __iter__(self):
        return self

    def __next__(self):
        if self.resident:
            i1, i2 = self._draw()
            scan1 = self.dev.index_select(0, torch.from_numpy(i1).to(self.device))
            scan2 = self.dev.index_select(0, torch.from_numpy(i2).to(self.device))
        else:
            dev, ev, _ = self.q.get()
            torch.cuda.current_stream(self.device).wait_event(ev)
            dev.record_stream(torch.cuda.current_stream(self.device))
            scan1, scan2 = dev[0], dev[1]
        invols = [scan1, scan2]
        outvols = [scan2, scan1] if self.bidir else [scan2]
        if not self.no_warp:
            if self.zeros is None:                # generators.py:98-101 (kept NCDHW like everything on this path)
                self.zeros = torch.zeros((self.batch_size, len(self.shape)) + self.shape, dtype=torch.float32, device=self.device)
            outvols.append(self.zeros)
        return invols, outvols


def scan_to_scan(vol_names, bidir=False, batch_size=1, prob_same=0, no_warp=False, **kwargs):
    """Drop-in for `voxelmorph.generators.scan_to_scan` (generators.py:71-107) yielding device tensors."""
    return PairLoader(vol_names, batch_size=batch_size, bidir=bidir, prob_same=prob_same, no_warp=no_warp, **kwargs)
