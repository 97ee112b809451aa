#!/usr/bin/env python
"""Developer stress: the U-Net parity check of tests/test_gpu_parity.py::test_unet_vs_oracle[kw0], repeated, printing the
worst parameter-gradient error of every repetition and bitwise run-to-run differences."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import voxelmorph_amd as vxm
from oracle import vxm_oracle as orc


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
inshape = (16, 16, 32)
torch.manual_seed(0)
net = vxm.networks.Unet(inshape, infeats=2).cuda()
sd = {("unet_model." + k): v for k, v in net.state_dict().items()}
rng = np.random.default_rng(4)
x = rng.standard_normal((2, 2) + inshape).astype(np.float32)
gy = None
sdo = {k: v.detach().cpu().double().requires_grad_() for k, v in sd.items()}
xo = torch.from_numpy(x).double().requires_grad_()
yo = orc.unet_forward(xo, sdo)
gy = rng.standard_normal(tuple(yo.shape)).astype(np.float32)
yo.backward(torch.from_numpy(gy).double())
first = None
for rep in range(reps):
    for p in net.parameters():
        p.grad = None
    xg = torch.from_numpy(x).cuda().requires_grad_()
    y = net(xg)
    y.backward(torch.from_numpy(gy).cuda())
    torch.cuda.synchronize()
    errs = {name: rel_l2(p.grad.cpu().numpy(), sdo["unet_model." + name].grad.numpy()) for name, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    cur = {name: p.grad.clone() for name, p in net.named_parameters()}
    cur["y"] = y.detach().clone()
    cur["gx"] = xg.grad.clone()
    diff = []
    if first is None:
        first = cur
    else:
        diff = [k for k in cur if not torch.equal(cur[k], first[k])]
    print("rep %d: y err %.2e gx err %.2e worst %s %.2e  bitwise-different vs rep 0: %s" % (
        rep, rel_l2(y.detach().cpu().numpy(), yo.detach().numpy()), rel_l2(xg.grad.cpu().numpy(), xo.grad.numpy()), worst, errs[worst], diff), flush=True)
