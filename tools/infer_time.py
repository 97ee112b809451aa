#!/usr/bin/env python
"""Developer timing: registration (inference) and the semi-supervised step (BASELINE configs[4] wiring) at 160x192x224 on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import voxelmorph_amd as vxm
from voxelmorph_amd.optim import FlatAdam

FULL = (160, 192, 224)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


torch.manual_seed(0)
model = vxm.networks.VxmDense(FULL, int_steps=7, int_downsize=2).cuda()
src, trg = torch.rand(1, 1, *FULL, device="cuda"), torch.rand(1, 1, *FULL, device="cuda")
with torch.no_grad():
    ms = timed(lambda: model(src, trg, registration=True))
print("registration (forward, registration=True, no_grad): %.2f ms per pair" % ms)

semi = vxm.networks.VxmDenseSemiSupervisedSeg(FULL, nb_labels=30, seg_resolution=2, int_steps=7, int_downsize=2).cuda()
opt = FlatAdam(semi, lr=1e-4)
lab = torch.randint(0, 30, (1, 80, 96, 112), device="cuda")
seg_src = torch.nn.functional.one_hot(lab, 30).permute(0, 4, 1, 2, 3).float().contiguous()
seg_trg = torch.nn.functional.one_hot(torch.roll(lab, 3, 2), 30).permute(0, 4, 1, 2, 3).float().contiguous()
ncc, grad, dice = vxm.losses.NCC().loss, vxm.losses.Grad("l2", loss_mult=2).loss, vxm.losses.Dice().loss


def step():
    opt.zero_grad()
    y, pre, yseg = semi(src, trg, seg_src)
    loss = ncc(trg, y) + grad(None, pre) + 0.01 * dice(seg_trg, yseg)
    loss.backward()
    opt.step()


print("semi-supervised training step (30 labels at half resolution, NCC + Grad + 0.01 Dice): %.2f ms" % timed(step))
