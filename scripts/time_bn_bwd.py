"""HBM rate of the BatchNorm-backward passes (reduce: reads g, y; apply: reads g, y, writes dy) on layer-sized tensors."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import P, stream
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
for c, h in ((16, 512), (32, 256), (64, 256), (64, 128), (128, 64), (256, 32), (512, 16)):
    npix = 16 * h * h
    g = torch.randn(npix, c, device=dev); y = torch.randn(npix, c, device=dev); dy = torch.empty_like(g)
    mean = torch.zeros(c, device=dev); rstd = torch.ones(c, device=dev); gamma = torch.ones(c, device=dev)
    scr = torch.zeros(2 * c, dtype=torch.float64, device=dev); dg = torch.zeros(c, device=dev); db = torch.zeros(c, device=dev)
    def run():
        L.check(L.lib().uwm_op_bn_backward(P(g), P(y), P(mean), P(rstd), P(gamma), P(scr), P(dy), P(dg), P(db), npix, c, stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = 5 * npix * c * 4
    print(f"C={c:4d} {h}x{h}: tensor {npix*c*4/1e6:7.1f} MB  reduce+apply {ms*1e3:7.1f} us  {byt/ms/1e9:6.2f} TB/s (5 passes)")
