"""Timing-only ablation of conv_patch on the layer1 shape (16x128x128, 64->64): set UWM_DBG before import."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, stream
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
n, cin, cout, h, w = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 64, 128, 128
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 164
x = torch.randn(n, h, w, cin, device=dev); wt = torch.randn(cout, 9 * cin, device=dev) * 0.05
y = torch.empty(n, h, w, cout, device=dev)
s0 = src(x)
def run():
    L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wt), cout, 9 * cin, 3, 3, 1, 1, n, cout, None, P(y), None, cfg, stream()))
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = 2.0 * n * h * w * cout * cin * 9
print(f"UWM_DBG={os.environ.get('UWM_DBG','0'):>2s} cfg={cfg} cin={cin} cout={cout}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TF/s")
