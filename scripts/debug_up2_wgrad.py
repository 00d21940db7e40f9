import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from tests.util import src, P, stream, rup, nhwc, unpack_w
from unet_watermark_amd import _lib as L
cuda = torch.device("cuda:0")
g = torch.Generator().manual_seed(39)
n, hs, ws = 2, 8, 16
d = torch.randn(n, 32, hs, ws, generator=g)
wt = (torch.randn(16, 32, 3, 3, generator=g) * 0.05).requires_grad_()
y = F.conv2d(F.interpolate(d, scale_factor=2, mode="nearest"), wt, None, 1, 1)
dy = torch.randn(y.shape, generator=g)
y.backward(dy)
dd, dyd = nhwc(d).to(cuda), nhwc(dy).to(cuda)
kpad = 288
s0 = src(dd, up=1)
dw = torch.zeros(16, kpad, device=cuda)
L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dyd), n, 2 * hs, 2 * ws, 16, 16, kpad, 3, 3, 1, 1, P(dw), 0, stream()))
torch.cuda.synchronize()
got = unpack_w(dw.cpu(), 16, 32, 3, 3); ref = wt.grad
print("nonzero", int((dw != 0).sum()), "of", dw.numel(), "max", float(dw.abs().max()), "ref max", float(ref.abs().max()))
err = (got - ref).abs()
print("max err", float(err.max()))
for r in range(3):
    for s in range(3):
        print("tap", r, s, "err", float(err[:, :, r, s].max()), "got max", float(got[:, :, r, s].abs().max()), "ref", float(ref[:, :, r, s].abs().max()))
print("by co", [round(float(err[co].max()), 2) for co in range(16)])
print("by c", [round(float(err[:, c].max()), 2) for c in range(32)])
