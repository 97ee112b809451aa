"""On-device data path for the VxmDense training step.

Replaces, for this path only, `voxelmorph/generators.py:9-194` (`volgen`, `scan_to_scan`, `scan_to_atlas`,
`semisupervised`) together with the host→device step of `scripts/torch/train.py:199-201`
(`torch.from_numpy(d).to(device).float().permute(0,4,1,2,3)`): the reference builds float64 numpy batches
`[B,*vol,C]` on the training thread, copies them pageable and converts / permutes on the device every step — at
tens of pairs/s per GPU that is the bottleneck.

Here each rank owns a loader built on `VolumeBank`s:
  * volumes (and label maps) are loaded once (nii / nii.gz / mgz / npz `vol` / `seg` / npy / preloaded arrays — `py/utils.py:69-129`;
    the NIfTI-1 and mgz readers are `voxelmorph_amd/nifti.py`, nibabel is not needed), converted to fp32 channels-first and kept in
    PINNED host memory;
    banks that fit are uploaded once and stay resident in HBM (288 GB per GPU: ~10,000 volumes of 160x192x224), so a
    step's "load" is a device-side gather into the batch tensors;
  * otherwise a background thread fills pinned staging batches and a dedicated HIP copy stream uploads batch k+1
    while batch k trains (`non_blocking=True` from pinned memory; an event orders the consumer stream);
  * batches come out already in the layout the kernels take: `[B, C, D, H, W]` fp32, contiguous.
The tuple contracts of the reference generators are kept — `(invols, outvols)`:
  scan_to_scan   (:71-107)   invols [scan1, scan2],            outvols [scan2(, scan1 if bidir)(, zeros)]
  scan_to_atlas  (:110-143)  invols [scan, atlas],             outvols [atlas | seg(, scan if bidir)(, zeros)]
  semisupervised (:146-194)  invols [src_vol, trg_vol, src_seg], outvols [trg_vol, zeros, trg_seg]
with the zero flow target allocated once and reused (as :98-101) and everything NCDHW on the device.  Random indices
are drawn like the reference (`np.random.randint`, :53) from a per-rank `numpy` generator, so ranks draw different
samples (data-parallel shards, `scripts/torch/train.py:128-129`).
"""
import glob
import os
import queue
import threading

import numpy as np
import torch


def load_volfile(filename, np_var='vol', add_batch_axis=False, add_feat_axis=False, pad_shape=None, resize_factor=1, ret_affine=False):
    """`voxelmorph/py/utils.py:69-129`: nii / nii.gz / mgz / npz / npy (a preloaded array is passed through; with ret_affine a
    preloaded `(vol, affine)` pair).  NIfTI-1 and mgz are read by `voxelmorph_amd.nifti` (no nibabel in this image): the volume as
    `np.squeeze(img.dataobj)` and the best affine of the header.  pad_shape / resize_factor as `py/utils.py:235-262`."""
    from . import nifti
    affine = None
    if not isinstance(filename, (str, os.PathLike)):
        if ret_affine:
            vol, affine = filename
        else:
            vol = filename
        vol = np.asarray(vol)
    else:
        filename = str(filename)
        if not os.path.isfile(filename):
            raise ValueError("'%s' is not a file." % filename)
        if filename.endswith(('.nii', '.nii.gz')):
            vol, affine = nifti.read_nifti(filename)
            vol = np.squeeze(vol)
        elif filename.endswith(('.mgz', '.mgh')):
            vol, affine = nifti.read_mgz(filename)
            vol = np.squeeze(vol)
        elif filename.endswith('.npy'):
            vol = np.load(filename)
        elif filename.endswith('.npz'):
            npz = np.load(filename)
            vol = next(iter(npz.values())) if len(npz.keys()) == 1 else npz[np_var]
        else:
            raise ValueError('unknown filetype for %s' % filename)
    if pad_shape:
        if vol.shape != tuple(pad_shape):                # py/utils.py:235-247: zero-pad, the array centred
            padded = np.zeros(pad_shape, dtype=vol.dtype)
            offsets = [int((p - v) / 2) for p, v in zip(pad_shape, vol.shape)]
            padded[tuple(slice(o, n + o) for o, n in zip(offsets, vol.shape))] = vol
            vol = padded
    if add_feat_axis:
        vol = vol[..., np.newaxis]
    if resize_factor != 1:                               # py/utils.py:250-262: nearest-neighbour zoom of every axis but the feature axis
        import scipy.ndimage
        vol = scipy.ndimage.zoom(vol, [resize_factor] * (vol.ndim - 1) + [1], order=0)
    if add_batch_axis:
        vol = vol[np.newaxis, ...]
    return (vol, affine) if ret_affine else vol


def save_volfile(array, filename, affine=None):
    """`voxelmorph/py/utils.py:132-158`: nii / nii.gz (the reference's default LIA affine centred on the volume when none is given) / npz"""
    from . import nifti
    filename = str(filename)
    array = np.asarray(array)
    if filename.endswith(('.nii', '.nii.gz')):
        if affine is None and array.ndim >= 3:
            affine = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=float)
            pcrs = np.append(np.array(array.shape[:3]) / 2, 1)
            affine[:3, 3] = -np.matmul(affine, pcrs)[:3]
        nifti.write_nifti(array, filename, affine)
    elif filename.endswith('.npz'):
        np.savez_compressed(filename, vol=array)
    else:
        raise ValueError('unknown filetype for %s' % filename)


def _resolve(vol_names):
    if isinstance(vol_names, str):                       # generators.py:38-42
        if os.path.isdir(vol_names):
            vol_names = os.path.join(vol_names, '*')
        vol_names = sorted(glob.glob(vol_names))
    return list(vol_names)


def _channels_first(vol, add_feat_axis):
    """`[*vol]` (add_feat_axis, the default of volgen :17) or `[*vol, C]` (multichannel data, train.py:101) -> `[C, *vol]` fp32."""
    vol = np.asarray(vol)
    if add_feat_axis:
        vol = vol[None]
    else:
        vol = np.moveaxis(vol, -1, 0)
    return np.ascontiguousarray(vol, dtype=np.float32)


class VolumeBank:
    """N same-shaped volumes as one pinned host tensor `[N, C, *vol]` fp32, and its HBM-resident copy when it fits."""

    def __init__(self, names, device, np_var='vol', add_feat_axis=True, resident_bytes=64 << 30, copy_stream=None):
        names = _resolve(names)
        if not names:
            raise ValueError('no volumes given')
        vols = [_channels_first(load_volfile(n, np_var), add_feat_axis) for n in names]
        shape = vols[0].shape
        if any(v.shape != shape for v in vols):
            raise ValueError('all volumes must share one shape (pad them first), got %s' % sorted({v.shape for v in vols}))
        self.n, self.channels, self.shape = len(vols), shape[0], tuple(shape[1:])
        host = torch.empty((self.n,) + tuple(shape), dtype=torch.float32)
        if torch.cuda.is_available():
            host = host.pin_memory()
        for i, v in enumerate(vols):
            host[i].copy_(torch.from_numpy(v))
        self.host = host
        self.device = device
        self.resident = host.numel() * 4 <= resident_bytes
        self.dev = None
        if self.resident:
            if copy_stream is not None:
                with torch.cuda.stream(copy_stream):
                    self.dev = host.to(device, non_blocking=True)
                copy_stream.synchronize()
            else:
                self.dev = host.to(device)

    def gather_device(self, idx):
        return self.dev.index_select(0, torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(self.device))

    def gather_host(self, idx):
        stage = torch.empty((len(idx),) + tuple(self.host.shape[1:]), dtype=torch.float32)
        if torch.cuda.is_available():
            stage = stage.pin_memory()
        torch.index_select(self.host, 0, torch.from_numpy(np.asarray(idx, dtype=np.int64)), out=stage)
        return stage


class _DeviceLoader:
    """Infinite iterator of `(invols, outvols)` device batches.  A subclass names what one batch gathers
    (`_draw() -> [(bank, indices), ...]`) and how the gathered tensors map onto the generator's tuples (`_assemble`)."""

    def __init__(self, device=None, rank=0, seed=0, prefetch=2):
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        if self.device.type != 'cuda':
            raise ValueError('%s feeds the MI355X path: device must be a HIP device, got %s' % (type(self).__name__, self.device))
        self.rng = np.random.default_rng([seed, rank])
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.prefetch = prefetch
        self.zeros = None
        self._q = None

    def _bank(self, names, **kw):
        return VolumeBank(names, self.device, copy_stream=self.copy_stream, **kw)

    def _start(self, banks):
        """Streaming mode (some bank is not resident): a producer thread gathers into pinned staging, the copy stream uploads."""
        self.resident = all(b.resident for b in banks)
        if not self.resident:
            self._q = queue.Queue(maxsize=max(1, self.prefetch))
            self._thread = threading.Thread(target=self._producer, daemon=True)
            self._thread.start()

    def _producer(self):
        while True:
            stages = [bank.gather_host(idx) for bank, idx in self._draw()]
            with torch.cuda.stream(self.copy_stream):
                devs = [s.to(self.device, non_blocking=True) for s in stages]
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self._q.put((devs, ev, stages))       # `stages` stay referenced until the copies have been consumed

    def _zero_field(self, batch, shape):
        if self.zeros is None:                    # generators.py:98-101 (kept NCDHW like everything on this path)
            self.zeros = torch.zeros((batch, len(shape)) + tuple(shape), dtype=torch.float32, device=self.device)
        return self.zeros

    def __iter__(self):
        return self

    def __next__(self):
        if self.resident:
            tensors = [bank.gather_device(idx) for bank, idx in self._draw()]
        else:
            tensors, ev, _ = self._q.get()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in tensors:
                t.record_stream(cur)
        return self._assemble(tensors)


class PairLoader(_DeviceLoader):
    """`scan_to_scan` (generators.py:71-107): two independent random draws per batch."""

    def __init__(self, vol_names, batch_size=1, bidir=False, prob_same=0, no_warp=False, np_var='vol', add_feat_axis=True,
                 device=None, rank=0, seed=0, resident_bytes=64 << 30, prefetch=2):
        super().__init__(device, rank, seed, prefetch)
        self.batch_size, self.bidir, self.prob_same, self.no_warp = batch_size, bidir, prob_same, no_warp
        self.bank = self._bank(vol_names, np_var=np_var, add_feat_axis=add_feat_axis, resident_bytes=resident_bytes)
        self.shape, self.channels, self.n = self.bank.shape, self.bank.channels, self.bank.n
        self._start([self.bank])

    def _draw(self):                              # generators.py:53,86-95
        i1 = self.rng.integers(self.n, size=self.batch_size)
        i2 = self.rng.integers(self.n, size=self.batch_size)
        if self.prob_same > 0 and self.rng.random() < self.prob_same:
            if self.rng.random() > 0.5:
                i1 = i2
            else:
                i2 = i1
        return [(self.bank, i1), (self.bank, i2)]

    def _assemble(self, tensors):
        scan1, scan2 = tensors
        invols = [scan1, scan2]
        outvols = [scan2, scan1] if self.bidir else [scan2]
        if not self.no_warp:
            outvols.append(self._zero_field(self.batch_size, self.shape))
        return invols, outvols


class AtlasLoader(_DeviceLoader):
    """`scan_to_atlas` (generators.py:110-143): random scans against ONE fixed atlas, which lives in HBM once and is
    handed out as a batch-expanded view (the reference `np.repeat`s it, :129); `segs` (True: the `seg` variable of the same
    npz files; or a parallel list) replaces the atlas in `outvols` for supervised training (:136-139)."""

    def __init__(self, vol_names, atlas, batch_size=1, bidir=False, no_warp=False, segs=None, np_var='vol', add_feat_axis=True,
                 device=None, rank=0, seed=0, resident_bytes=64 << 30, prefetch=2):
        super().__init__(device, rank, seed, prefetch)
        self.batch_size, self.bidir, self.no_warp = batch_size, bidir, no_warp
        self.bank = self._bank(vol_names, np_var=np_var, add_feat_axis=add_feat_axis, resident_bytes=resident_bytes)
        self.shape, self.channels, self.n = self.bank.shape, self.bank.channels, self.bank.n
        self.seg_bank = None
        if segs is True:
            self.seg_bank = self._bank(vol_names, np_var='seg', add_feat_axis=add_feat_axis, resident_bytes=resident_bytes)
        elif isinstance(segs, (list, tuple)):
            if len(segs) != self.n:               # generators.py:44-45
                raise ValueError('Number of image files must match number of seg files.')
            self.seg_bank = self._bank(list(segs), np_var=np_var, add_feat_axis=add_feat_axis, resident_bytes=resident_bytes)
        atlas = load_volfile(atlas, np_var) if isinstance(atlas, (str, os.PathLike)) else np.asarray(atlas)
        if atlas.ndim == len(self.shape) + 2:     # the reference passes [1, *vol, C] (train.py:105-106)
            atlas = atlas[0]
            atlas_c = _channels_first(atlas, add_feat_axis=False)
        else:
            atlas_c = _channels_first(atlas, add_feat_axis)
        if tuple(atlas_c.shape[1:]) != self.shape:
            raise ValueError('atlas shape %s does not match the scans %s' % (tuple(atlas_c.shape[1:]), self.shape))
        self.atlas = torch.from_numpy(atlas_c).to(self.device)[None].expand(batch_size, *atlas_c.shape).contiguous()
        self._start([b for b in (self.bank, self.seg_bank) if b is not None])

    def _draw(self):
        idx = self.rng.integers(self.n, size=self.batch_size)
        return [(self.bank, idx)] + ([(self.seg_bank, idx)] if self.seg_bank is not None else [])

    def _assemble(self, tensors):
        scan = tensors[0]
        invols = [scan, self.atlas]
        first = tensors[1] if self.seg_bank is not None else self.atlas
        outvols = [first, scan] if self.bidir else [first]
        if not self.no_warp:
            outvols.append(self._zero_field(self.batch_size, self.shape))
        return invols, outvols


class SemiSupervisedLoader(_DeviceLoader):
    """`semisupervised` (generators.py:146-194): (volume, label map) pairs; the discrete label map becomes a one-hot
    `[B, len(labels), *vol/downsize]` tensor ON THE DEVICE (`split_seg`, :161-165: `seg == label` per label, then every
    `downsize`-th voxel), from the resident label volume instead of a float64 host array of 30 full-size planes.  With
    `atlas_file` the target volume / one-hot target segmentation are fixed (:168-173).  The reference generator is
    batch-1; `batch_size` is this package's per-rank extension."""

    def __init__(self, vol_names, seg_names, labels, atlas_file=None, downsize=2, batch_size=1, device=None, rank=0, seed=0,
                 resident_bytes=64 << 30, prefetch=2):
        super().__init__(device, rank, seed, prefetch)
        self.batch_size, self.downsize = batch_size, int(downsize)
        self.bank = self._bank(vol_names, np_var='vol', resident_bytes=resident_bytes)
        segs = _resolve(seg_names)
        if len(segs) != self.bank.n:
            raise ValueError('Number of image files must match number of seg files.')
        self.seg_bank = self._bank(segs, np_var='vol', resident_bytes=resident_bytes)      # volgen(segs=list) loads np_var (:64-67)
        if self.seg_bank.shape != self.bank.shape:
            raise ValueError('segmentations %s do not match the volumes %s' % (self.seg_bank.shape, self.bank.shape))
        self.shape, self.channels, self.n = self.bank.shape, 1, self.bank.n
        self.labels = torch.as_tensor(np.asarray(labels, dtype=np.float32), device=self.device).view(1, -1, *([1] * len(self.shape)))
        self.nb_labels = self.labels.shape[1]
        self.trg_vol = self.trg_seg = None
        if atlas_file:
            vol = _channels_first(load_volfile(atlas_file, 'vol'), True)
            seg = _channels_first(load_volfile(atlas_file, 'seg'), True)
            self.trg_vol = torch.from_numpy(vol).to(self.device)[None].expand(batch_size, *vol.shape).contiguous()
            self.trg_seg = self.split_seg(torch.from_numpy(seg).to(self.device)[None].expand(batch_size, *seg.shape))
        self._start([self.bank, self.seg_bank])

    def split_seg(self, seg):
        """[B,1,*vol] label map -> [B,L,*vol/downsize] one-hot fp32 (generators.py:161-165)."""
        sl = (slice(None), slice(None)) + (slice(None, None, self.downsize),) * len(self.shape)
        return (seg[sl] == self.labels).to(torch.float32).contiguous()

    def _draw(self):
        i = self.rng.integers(self.n, size=self.batch_size)
        slots = [(self.bank, i), (self.seg_bank, i)]
        if self.trg_vol is None:
            j = self.rng.integers(self.n, size=self.batch_size)
            slots += [(self.bank, j), (self.seg_bank, j)]
        return slots

    def _assemble(self, tensors):
        src_vol, src_seg = tensors[0], self.split_seg(tensors[1])
        if self.trg_vol is None:
            trg_vol, trg_seg = tensors[2], self.split_seg(tensors[3])
        else:
            trg_vol, trg_seg = self.trg_vol, self.trg_seg
        invols = [src_vol, trg_vol, src_seg]
        outvols = [trg_vol, self._zero_field(self.batch_size, self.shape), trg_seg]
        return invols, outvols


def scan_to_scan(vol_names, bidir=False, batch_size=1, prob_same=0, no_warp=False, **kwargs):
    """Drop-in for `voxelmorph.generators.scan_to_scan` (generators.py:71-107) yielding device tensors."""
    return PairLoader(vol_names, batch_size=batch_size, bidir=bidir, prob_same=prob_same, no_warp=no_warp, **kwargs)


def scan_to_atlas(vol_names, atlas, bidir=False, batch_size=1, no_warp=False, segs=None, **kwargs):
    """Drop-in for `voxelmorph.generators.scan_to_atlas` (generators.py:110-143) yielding device tensors."""
    return AtlasLoader(vol_names, atlas, batch_size=batch_size, bidir=bidir, no_warp=no_warp, segs=segs, **kwargs)


def semisupervised(vol_names, seg_names, labels, atlas_file=None, downsize=2, **kwargs):
    """Drop-in for `voxelmorph.generators.semisupervised` (generators.py:146-194) yielding device tensors."""
    return SemiSupervisedLoader(vol_names, seg_names, labels, atlas_file=atlas_file, downsize=downsize, **kwargs)
