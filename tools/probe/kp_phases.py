import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from voxelmorph_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["VXM_LIB"])
from voxelmorph_amd.torch import functional as VF
D, H, W = 160, 192, 224; V = D * H * W
a, b2 = torch.rand(1, 1, D, H, W, device="cuda"), torch.rand(1, 1, D, H, W, device="cuda")
w = torch.randn(16, 2, 3, 3, 3, device="cuda") * 0.1; bias = torch.randn(16, device="cuda")
y = torch.empty(1, 16, D, H, W, device="cuda")
for _ in range(5):
    VF.conv_forward(a, 1, V, False, b2, 1, V, w, bias, y, 16 * V, 16, 0.2, 1, D, H, W)
torch.cuda.synchronize()
h = _lib.lib(); h.vxm_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(8 * 8 * 4096, dtype=np.int64); assert h.vxm_debug_read(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(512, 8, 8, 8).astype(np.float64)      # block, iter, wave, slot
ok = t[:, 1:7, :, 0] > 0
names = ["issue next loads", "mfma phase", "stage next (vmcnt)", "epilogue", "barrier"]
for i, nm in enumerate(names):
    d = (t[:, 1:7, :, i + 1] - t[:, 1:7, :, i])[ok]
    print("%-22s mean %7.0f p50 %7.0f p90 %7.0f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
tot = (t[:, 2:7, :, 0] - t[:, 1:6, :, 0])[t[:, 2:7, :, 0] > 0]
print("iteration total        mean %7.0f p50 %7.0f" % (tot.mean(), np.median(tot)))
