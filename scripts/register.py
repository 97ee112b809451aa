#!/usr/bin/env python
"""Register a moving volume to a fixed one with a trained VxmDense on MI355X — the command line of the reference's
`scripts/torch/register.py` (:49-58: --moving --fixed --moved --model --warp -g/--gpu --multichannel); reference checkpoints load unchanged
(`LoadableModel.load`, modelio.py:69-77).  nii / nii.gz / mgz / npz / npy in, nii / nii.gz / npz / npy out (voxelmorph_amd/nifti.py: the
reference's nibabel is not needed); the moved image and the warp are saved with the FIXED image's affine, as register.py:75,88-92 does;
`--seg` additionally warps a label map with the bit-exact nearest-neighbour transformer, and `--jacobian`
reports the fraction of voxels with a non-positive Jacobian determinant of the deformation (py/utils.py:473-516)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def save_vol(arr, path, affine=None):
    from voxelmorph_amd import data as vdata
    if path.endswith('.npy'):
        np.save(path, arr)
    else:
        vdata.save_volfile(arr, path, affine)          # nii / nii.gz / npz (py/utils.py:132-158)


def jacobian_determinant(disp):
    """disp [3, D, H, W] in voxels -> det(I + grad disp) [D, H, W] with np.gradient's differences (central inside, one-sided at
    the borders), the cofactor expansion of py/utils.py:473-516; runs where the tensor lives."""
    g = [torch.gradient(disp[a], dim=(0, 1, 2)) for a in range(3)]
    J = torch.stack([torch.stack(list(g[a]), 0) for a in range(3)], 0)          # [3(a), 3(axis), D, H, W]
    J = J + torch.eye(3, device=disp.device, dtype=disp.dtype)[:, :, None, None, None]
    return (J[0, 0] * (J[1, 1] * J[2, 2] - J[1, 2] * J[2, 1]) - J[0, 1] * (J[1, 0] * J[2, 2] - J[1, 2] * J[2, 0])
            + J[0, 2] * (J[1, 0] * J[2, 1] - J[1, 1] * J[2, 0]))


def nonpositive_jacobian_fraction(disp):
    """fraction of voxels where the deformation folds (det <= 0)"""
    return float((jacobian_determinant(disp) <= 0).float().mean())


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--moving', required=True, help='moving image (source) filename')
    p.add_argument('--fixed', required=True, help='fixed image (target) filename')
    p.add_argument('--moved', required=True, help='warped image output filename')
    p.add_argument('--model', required=True, help='pytorch model for nonlinear registration')
    p.add_argument('--warp', help='output warp deformation filename')
    p.add_argument('--seg', help='label map of the moving image to warp with nearest-neighbour interpolation')
    p.add_argument('--moved-seg', help='output filename of the warped label map')
    p.add_argument('--jacobian', action='store_true', help='print the fraction of non-positive Jacobian determinants')
    p.add_argument('-g', '--gpu', default='0', help='GPU number (this path has no CPU fallback)')
    p.add_argument('--multichannel', action='store_true', help='volumes carry a trailing feature axis [*vol, C]')
    args = p.parse_args(argv)

    import voxelmorph_amd as vxm
    from voxelmorph_amd import data as vdata
    dev = torch.device('cuda', int(args.gpu))
    torch.cuda.set_device(dev)

    affines = {}

    def load(path, multichannel=False):
        """[*vol] (or [*vol, C] with --multichannel, register.py:69-72) -> [1, C, *vol] fp32 on the device"""
        vol, affines[path] = vdata.load_volfile(path, ret_affine=True)
        vol = np.asarray(vol)
        vol = np.moveaxis(vol, -1, 0) if multichannel else vol[None]
        return torch.from_numpy(np.ascontiguousarray(vol, dtype=np.float32))[None].to(dev)

    moving, fixed = load(args.moving, args.multichannel), load(args.fixed, args.multichannel)
    fixed_affine = affines[args.fixed]
    model = vxm.networks.VxmDense.load(args.model, dev)
    model.to(dev)
    model.eval()
    with torch.no_grad():
        moved, warp = model(moving, fixed, registration=True)
        save_vol((moved[0].permute(1, 2, 3, 0) if args.multichannel else moved).cpu().numpy().squeeze(), args.moved, fixed_affine)
        if args.warp:
            save_vol(warp.cpu().numpy().squeeze(), args.warp, fixed_affine)
        if args.seg:
            seg = load(args.seg)
            out = vxm.layers.SpatialTransformer(seg.shape[2:], mode='nearest').to(dev)(seg, warp)
            save_vol(out.cpu().numpy().squeeze(), args.moved_seg or (os.path.splitext(args.moved)[0] + '_seg.npz'), fixed_affine)
        if args.jacobian:
            print('non-positive Jacobian fraction: %.6f' % nonpositive_jacobian_fraction(warp[0]))


if __name__ == '__main__':
    main()
