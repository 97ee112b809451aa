#!/usr/bin/env python
"""Scan-to-scan / scan-to-atlas VxmDense training on MI355X — the caller of the hot path, taking the command line of the reference's
`scripts/torch/train.py` (same flag names / defaults; :52-91) and keeping its loop semantics (:184-233: weighted loss
list, Adam lr 1e-4, a checkpoint every 20 epochs + the final one, `%04d.pt` names), with three differences that are
the point of this package:

  * data parallelism is one process per GPU (launch with `python -m torch.distributed.run --nproc-per-node N
    --master-addr 127.0.0.1 scripts/train.py ...`, or pass the reference's `--gpu 0,1,..`: the script then re-launches
    itself that way on those devices), each rank trains `--batch-size / N` pairs per step and the only exchange is one
    RCCL all-reduce of the flat gradient bucket (`FlatAdam.step`), instead of `torch.nn.DataParallel` (:151-154);
  * batches come from `voxelmorph_amd.data.PairLoader` (volumes pinned / resident in HBM, uploads on a copy stream)
    instead of numpy generators + a per-step pageable copy and permute (:199-201);
  * losses are accumulated on the device and read back once per epoch instead of three `.item()` syncs per step
    (:211,:215).

Volumes: npz (`vol`) / npy files listed one per line in --img-list (NIfTI needs nibabel, absent here).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


# The reference's flags (scripts/torch/train.py:52-91) with its defaults; only --save-every is new.  --cudnn-nondet is
# accepted and ignored: there is no cuDNN / MIOpen algorithm choice on this path (the kernels are fixed and deterministic).
_FLAGS = [
    # name, argparse keywords
    ('--img-list', dict(required=True, help='text file with one training volume (npz / npy) per line')),
    ('--img-prefix', dict(help='string put in front of every entry of --img-list')),
    ('--img-suffix', dict(help='string appended to every entry of --img-list')),
    ('--atlas', dict(help='atlas volume (npz `vol` / npy): scan-to-atlas training instead of scan-to-scan')),
    ('--model-dir', dict(default='models', help='where checkpoints go [models]')),
    ('--multichannel', dict(action='store_true', help='volumes carry a trailing feature axis [*vol, C]')),
    ('--gpu', dict(default=None, help='device id(s), comma-separated; several ids = one rank per id [LOCAL_RANK, else 0]')),
    ('--batch-size', dict(type=int, default=1, help='GLOBAL batch size, split evenly over the ranks [1]')),
    ('--epochs', dict(type=int, default=1500, help='epochs to train [1500]')),
    ('--steps-per-epoch', dict(type=int, default=100, help='optimiser steps per epoch [100]')),
    ('--load-model', dict(help='checkpoint to start from (this package\'s or the reference\'s .pt)')),
    ('--initial-epoch', dict(type=int, default=0, help='epoch counter to resume at [0]')),
    ('--lr', dict(type=float, default=1e-4, help='Adam learning rate [1e-4]')),
    ('--cudnn-nondet', dict(action='store_true', help='accepted for command-line compatibility; no effect here')),
    ('--enc', dict(type=int, nargs='+', help='U-Net encoder features per level [16 32 32 32]')),
    ('--dec', dict(type=int, nargs='+', help='U-Net decoder features, extra entries = full-resolution convs [32 32 32 32 32 16 16]')),
    ('--int-steps', dict(type=int, default=7, help='scaling-and-squaring steps, 0 = no integration [7]')),
    ('--int-downsize', dict(type=int, default=2, help='integrate the field at 1/N of the image resolution [2]')),
    ('--bidir', dict(action='store_true', help='also warp the target onto the source and average both image losses')),
    ('--image-loss', dict(default='mse', help="'mse' or 'ncc' [mse]")),
    ('--lambda', dict(type=float, dest='weight', default=0.01, help='weight of the smoothness (Grad) term [0.01]')),
    ('--save-every', dict(type=int, default=20, help='epochs between checkpoints [20, as the reference]')),
]


def parse(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser.parse_args(argv)


def read_file_list(path, prefix=None, suffix=None):
    with open(path) as f:
        names = [ln.strip() for ln in f if ln.strip()]
    return [(prefix or '') + n + (suffix or '') for n in names]


def main(argv=None):
    args = parse(argv)
    import voxelmorph_amd as vxm
    from voxelmorph_amd import data as vdata
    from voxelmorph_amd import dist as vdist
    from voxelmorph_amd.optim import FlatAdam

    gpus = [g for g in (args.gpu or '').split(',') if g != '']
    if len(gpus) > 1:           # the reference's DataParallel request (train.py:123-129,151-154) = one rank per listed device
        vdist.self_launch(len(gpus), os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), ','.join(gpus))
    elif len(gpus) == 1 and 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = gpus[0]
    rank, local, world = vdist.init_from_env()
    files = read_file_list(args.img_list, args.img_prefix, args.img_suffix)
    assert len(files) > 0, 'Could not find any training data.'
    lo, hi = vdist.shard_range(args.batch_size, rank, world)        # asserts batch % world == 0 like train.py:128-129
    dev = torch.device('cuda', local)
    add_feat_axis = not args.multichannel                           # train.py:101
    if args.atlas:                                                  # train.py:103-109
        loader = vdata.scan_to_atlas(files, args.atlas, batch_size=hi - lo, bidir=args.bidir, add_feat_axis=add_feat_axis,
                                     device=dev, rank=rank)
    else:
        loader = vdata.scan_to_scan(files, batch_size=hi - lo, bidir=args.bidir, add_feat_axis=add_feat_axis, device=dev, rank=rank)
    inshape = loader.shape

    enc = args.enc if args.enc else [16, 32, 32, 32]
    dec = args.dec if args.dec else [32, 32, 32, 32, 32, 16, 16]
    if args.load_model:
        model = vxm.networks.VxmDense.load(args.load_model, dev)
    else:
        # (the reference builds the model for one feature per image even with --multichannel and then fails in its first
        # conv for C > 1; here the channel count of the data sizes the network input)
        model = vxm.networks.VxmDense(inshape=inshape, nb_unet_features=[enc, dec], bidir=args.bidir, int_steps=args.int_steps,
                                      int_downsize=args.int_downsize, src_feats=loader.channels, trg_feats=loader.channels)
    model.to(dev)
    model.train()
    opt = FlatAdam(model, lr=args.lr, comm=vdist.native_comm())
    opt.broadcast_params(0)

    if args.image_loss == 'ncc':
        image_loss = vxm.losses.NCC().loss
    elif args.image_loss == 'mse':
        image_loss = vxm.losses.MSE().loss
    else:
        raise ValueError('Image loss should be "mse" or "ncc", but found "%s"' % args.image_loss)
    losses = [image_loss, image_loss] if args.bidir else [image_loss]
    weights = [0.5, 0.5] if args.bidir else [1.0]
    losses.append(vxm.losses.Grad('l2', loss_mult=args.int_downsize).loss)
    weights.append(args.weight)

    os.makedirs(args.model_dir, exist_ok=True)
    from voxelmorph_amd.pacing import InFlight
    from voxelmorph_amd.graph import GraphedStep
    pace = InFlight(2)
    # The loop body of the reference (train.py:194-223) as ONE hipGraph launch per step (voxelmorph_amd/graph.py): the batch lives in static
    # tensors that every step refills in place, the loss terms accumulate into a static tensor; the first two steps run launch by launch
    # (VXM_GRAPH=0: every step does).
    inputs, y_true = next(loader)
    static_in, static_true = [t.clone() for t in inputs], [t.clone() if torch.is_tensor(t) else t for t in y_true]
    terms = torch.zeros(len(losses) + 1, device=dev)

    def forward_loss():
        y_pred = model(*static_in)
        # train.py:205-212 (`loss += loss_function(y_true[n], y_pred[n]) * weights[n]`): products, sum and the per-epoch accumulators in one launch
        return vxm.losses.weighted_sum([fn(static_true[n], y_pred[n]) for n, fn in enumerate(losses)], weights, running=terms)

    step = GraphedStep(forward_loss, opt, eager_steps=2, enabled=os.environ.get('VXM_GRAPH', '1') != '0')
    fresh = False                            # the first batch is in the static tensors already
    for epoch in range(args.initial_epoch, args.epochs):
        if rank == 0 and epoch % args.save_every == 0:
            model.save(os.path.join(args.model_dir, '%04d.pt' % epoch))
        terms.zero_()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.steps_per_epoch):
            pace.wait()                      # at most two steps in flight (voxelmorph_amd/pacing.py)
            if fresh:
                inputs, y_true = next(loader)
                for dst, src in zip(static_in + static_true, list(inputs) + list(y_true)):
                    if torch.is_tensor(dst):
                        # the captured step reads these tensors by address: a batch of another shape / dtype (a short last batch, a loader
                        # that changed its mind) must not be broadcast into them silently
                        if not torch.is_tensor(src) or src.shape != dst.shape or src.dtype != dst.dtype:
                            raise RuntimeError('train.py: the loader returned %s where the static batch tensor is %s %s -- the graphed step '
                                               'needs batches of one shape (set VXM_GRAPH=0 for ragged batches)'
                                               % ('%s %s' % (tuple(src.shape), src.dtype) if torch.is_tensor(src) else type(src).__name__,
                                                  tuple(dst.shape), dst.dtype))
                        dst.copy_(src)
                    elif dst is not src and dst != src:
                        raise RuntimeError('train.py: a non-tensor entry of y_true changed between batches (%r -> %r); the graphed step '
                                           'captured the first value' % (dst, src))
            fresh = True
            step()
            pace.mark()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.steps_per_epoch
        if rank == 0:
            t = (terms / args.steps_per_epoch).tolist()
            print('Epoch %d/%d - %.4f sec/step - loss: %.4e  (%s)' % (epoch + 1, args.epochs, dt, t[-1], ', '.join('%.4e' % v for v in t[:-1])),
                  flush=True)
    if rank == 0:
        model.save(os.path.join(args.model_dir, '%04d.pt' % args.epochs))


if __name__ == '__main__':
    main()
