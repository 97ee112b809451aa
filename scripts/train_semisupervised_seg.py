#!/usr/bin/env python
"""Semi-supervised VxmDense training on MI355X (BASELINE.json configs[4]): the registration step plus an auxiliary Dice loss on
the warped, down-sampled one-hot source segmentation.

The reference has this entry point for its TensorFlow backend only (`scripts/tf/train_semisupervised_seg.py`); this script
takes that command line (:41-79, same flag names / defaults) and runs its loop (:117-150: losses `[image, Grad('l2',
loss_mult=int_downsize), Dice]`, weights `[1, --grad-loss-weight, --dice-loss-weight]`, Adam, a checkpoint of the starting
weights + every 20 epochs) on `voxelmorph_amd.networks.VxmDenseSemiSupervisedSeg` with batches from
`voxelmorph_amd.data.semisupervised` (generators.py:146-194: `[src_vol, trg_vol, src_seg] -> [trg_vol, zeros, trg_seg]`,
one-hot segmentations at half resolution built on the device).  Checkpoints are `%04d.pt` (torch) instead of `.h5`.
Data parallel as scripts/train.py: one process per GPU, `--batch-size` (an addition; the reference generator is batch-1)
split over the ranks, one RCCL all-reduce of the flat gradient bucket per step.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

_FLAGS = [
    ('--img-list', dict(required=True, help='text file with one training sample per line')),
    ('--img-suffix', dict(help='suffix turning a list entry into the image file')),
    ('--seg-suffix', dict(help='suffix turning a list entry into the label-map file')),
    ('--img-prefix', dict(help='prefix turning a list entry into the image file')),
    ('--seg-prefix', dict(help='prefix turning a list entry into the label-map file')),
    ('--labels', dict(required=True, help='npy file with the label values entering the Dice loss')),
    ('--model-dir', dict(default='models', help='where checkpoints go [models]')),
    ('--atlas', dict(help='npz with `vol` and `seg`: scan-to-atlas training')),
    ('--gpu', dict(default=None, help='device id(s), comma-separated; several ids = one rank per id [LOCAL_RANK, else 0]')),
    ('--batch-size', dict(type=int, default=1, help='GLOBAL batch size, split evenly over the ranks [1]')),
    ('--epochs', dict(type=int, default=1500, help='epochs to train [1500]')),
    ('--steps-per-epoch', dict(type=int, default=100, help='optimiser steps per epoch [100]')),
    ('--load-weights', dict(help='checkpoint (.pt of this script) to start from')),
    ('--initial-epoch', dict(type=int, default=0, help='epoch counter to resume at [0]')),
    ('--lr', dict(type=float, default=1e-4, help='Adam learning rate [1e-4]')),
    ('--enc', dict(type=int, nargs='+', help='U-Net encoder features per level [16 32 32 32]')),
    ('--dec', dict(type=int, nargs='+', help='U-Net decoder features [32 32 32 32 32 16 16]')),
    ('--int-steps', dict(type=int, default=7, help='scaling-and-squaring steps [7]')),
    ('--int-downsize', dict(type=int, default=2, help='integrate the field at 1/N of the image resolution [2]')),
    ('--image-loss', dict(default='mse', help="'mse' or 'ncc' [mse]")),
    ('--grad-loss-weight', dict(type=float, default=0.01, help='weight of the smoothness term (lambda) [0.01]')),
    ('--dice-loss-weight', dict(type=float, default=0.01, help='weight of the Dice term (gamma) [0.01]')),
    ('--save-every', dict(type=int, default=20, help='epochs between checkpoints [20, as the reference]')),
]


def parse(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for flag, kw in _FLAGS:
        parser.add_argument(flag, **kw)
    return parser.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    import voxelmorph_amd as vxm
    from voxelmorph_amd import data as vdata
    from voxelmorph_amd import dist as vdist
    from voxelmorph_amd.optim import FlatAdam
    from train import read_file_list

    if args.img_prefix == args.seg_prefix and args.img_suffix == args.seg_suffix:       # tf script :82-85
        print('Error: Must provide a differing file suffix and/or prefix for images and segs.')
        sys.exit(1)
    gpus = [g for g in (args.gpu or '').split(',') if g != '']
    if len(gpus) > 1:
        vdist.self_launch(len(gpus), os.path.abspath(__file__), sys.argv[1:] if argv is None else list(argv), ','.join(gpus))
    elif len(gpus) == 1 and 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = gpus[0]
    rank, local, world = vdist.init_from_env()
    imgs = read_file_list(args.img_list, args.img_prefix, args.img_suffix)
    segs = read_file_list(args.img_list, args.seg_prefix, args.seg_suffix)
    assert len(imgs) > 0, 'Could not find any training data.'
    labels = np.load(args.labels)
    lo, hi = vdist.shard_range(args.batch_size, rank, world)
    dev = torch.device('cuda', local)
    loader = vdata.semisupervised(imgs, segs, labels=labels, atlas_file=args.atlas, downsize=2, batch_size=hi - lo,
                                  device=dev, rank=rank)

    enc = args.enc if args.enc else [16, 32, 32, 32]
    dec = args.dec if args.dec else [32, 32, 32, 32, 32, 16, 16]
    if args.load_weights:
        model = vxm.networks.VxmDenseSemiSupervisedSeg.load(args.load_weights, dev)
    else:
        model = vxm.networks.VxmDenseSemiSupervisedSeg(inshape=loader.shape, nb_unet_features=[enc, dec], nb_labels=len(labels),
                                                       int_steps=args.int_steps, int_downsize=args.int_downsize)
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
    losses = [image_loss, vxm.losses.Grad('l2', loss_mult=args.int_downsize).loss, vxm.losses.Dice().loss]
    weights = [1.0, args.grad_loss_weight, args.dice_loss_weight]

    os.makedirs(args.model_dir, exist_ok=True)
    if rank == 0:
        model.save(os.path.join(args.model_dir, '%04d.pt' % args.initial_epoch))       # tf script :143
    from voxelmorph_amd.pacing import InFlight
    pace = InFlight(2)
    for epoch in range(args.initial_epoch, args.epochs):
        terms = torch.zeros(len(losses) + 1, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(args.steps_per_epoch):
            pace.wait()                      # at most two steps in flight (voxelmorph_amd/pacing.py)
            inputs, y_true = next(loader)
            y_pred = model(*inputs)
            loss = vxm.losses.weighted_sum([fn(y_true[n], y_pred[n]) for n, fn in enumerate(losses)], weights, running=terms)
            opt.zero_grad()
            loss.backward()
            opt.step()
            pace.mark()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / args.steps_per_epoch
        if rank == 0:
            t = (terms / args.steps_per_epoch).tolist()
            print('Epoch %d/%d - %.4f sec/step - loss: %.4e  (%s)' % (epoch + 1, args.epochs, dt, t[-1], ', '.join('%.4e' % v for v in t[:-1])),
                  flush=True)
            if (epoch + 1) % args.save_every == 0 or epoch + 1 == args.epochs:
                model.save(os.path.join(args.model_dir, '%04d.pt' % (epoch + 1)))


if __name__ == '__main__':
    main()
