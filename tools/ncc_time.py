import os, sys, torch
sys.path.insert(0, os.getcwd())
import voxelmorph_amd as vxm
torch.manual_seed(0)
I = torch.rand(1, 1, 160, 192, 224, device="cuda"); J = torch.rand_like(I).requires_grad_()
def step():
    l = vxm.losses.NCC().loss(I, J); l.backward(); return l
for _ in range(3): step()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): l = step()
e.record(); torch.cuda.synchronize()
print("seg", os.environ.get("VXM_NCC_SEG", "auto"), "fwd+bwd %.3f ms" % (s.elapsed_time(e) / 20), "loss %.9f" % float(l), "gsum %.9e" % float(J.grad.double().abs().sum()))
