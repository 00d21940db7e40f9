"""Time decoder block 4 conv1's dgrad (16 -> 32 channels at 512^2 with the concat-split / 2x2 pooling epilogue) through the op
entry point: auto route vs UWM_NO_UP2=1 (Winograd + fused split)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import P, stream, rup
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
n, h, w, c0, cout = 16, 256, 256, 32, 16
dy = torch.randn(n, 2 * h, 2 * w, cout, device=dev)
kpadd = rup(9 * cout, 32)
wd = torch.randn(c0, kpadd, device=dev) * 0.05
pm = torch.randn(n, h, w, c0, device=dev); sc = torch.rand(c0, device=dev) + 0.5; sh = torch.randn(c0, device=dev) * 0.1
gp = torch.empty(n, h, w, c0, device=dev)
def run():
    L.check(L.lib().uwm_op_dgrad_upsplit(P(dy), n, 2 * h, 2 * w, cout, P(wd), c0, 0, kpadd, P(gp), P(pm), P(sc), P(sh), None, stream()))
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"dec4c1 dgrad+split UWM_NO_UP2={os.environ.get('UWM_NO_UP2', '0')}: {e0.elapsed_time(e1) * 100:.1f} us (includes the op's filter-bank prepare launch)")
