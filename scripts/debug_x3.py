"""Debug harness for conv_wino_x3 (cfg 400): structured inputs that expose layout mistakes."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import nhwc, nchw, pack_w, src, P, stream
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
def run(x, wt, cfg):
    n, cin, h, w = x.shape; cout = wt.shape[0]
    xd = nhwc(x.to(dev)); wp, kpad = pack_w(wt.to(dev))
    out = torch.full((n, h, w, cout), float("nan"), device=dev)
    s0 = src(xd)
    L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wp), cout, kpad, 3, 3, 1, 1, n, cout, None, P(out), None, cfg, stream()))
    torch.cuda.synchronize()
    return nchw(out.cpu(), cout)
n, cin, cout, h, w = 1, 16, 32, 16, 32
wt = torch.zeros(cout, cin, 3, 3)
for c in range(cin): wt[c, c, 1, 1] = 1.0
print("effective out[co] response to a unit constant image in channel k (centre-tap identity weights): rows k, nonzero (co, value)")
for k in range(16):
    x = torch.zeros(n, cin, h, w); x[:, k] = 1.0
    y = run(x, wt, 400)[0, :, 5, 7]
    print(k, [(int(c), round(float(y[c]), 3)) for c in range(cout) if abs(float(y[c])) > 1e-6])
print("weights single (co=2, ci=k) centre tap, image all ones: out[2] should be 1")
for k in range(8):
    wt2 = torch.zeros(cout, cin, 3, 3); wt2[2, k, 1, 1] = 1.0
    x = torch.ones(n, cin, h, w)
    y = run(x, wt2, 400)[0, :, 5, 7]
    print(k, [(int(c), round(float(y[c]), 3)) for c in range(cout) if abs(float(y[c])) > 1e-6])
print("image value scan (identity weights, channel 0 = v): out[0]")
for v in (0.5, 1.0, 2.0, 3.0, 1.5):
    x = torch.zeros(n, cin, h, w); x[:, 0] = v
    y = run(x, wt, 400)[0, :, 5, 7]
    print(v, [(int(c), round(float(y[c]), 4)) for c in range(cout) if abs(float(y[c])) > 1e-6])
