"""Kernel-at-a-time gate for wgrad_f16x3.hip against the fp32 Winograd-domain weight gradient (uwm_op_wgrad, resnet34 / decoder
shapes at 16 x 512^2; both include their partial-sum reduce, the fp16x3 one also the op entry's max|dy| reduction)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, stream, rup
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
N = 16


def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, cin, cout, h in (("layer1", 64, 64, 128), ("layer2", 128, 128, 64), ("layer3", 256, 256, 32), ("dec2.c2", 64, 64, 128),
                           ("dec1.c1", 384, 128, 64), ("dec0.c1", 768, 256, 32), ("dec3.c1", 128, 32, 256), ("dec3.c2", 32, 32, 256)):
    x = torch.randn(N, h, h, cin, device=dev); dy = torch.randn(N, h, h, cout, device=dev) * 1e-5
    kpad = rup(9 * cin, 32)
    dw = torch.zeros(cout, kpad, device=dev)
    s0 = src(x)
    res = {}
    for tag, force in (("wino", 0), ("f16x3", 6)):
        f = lambda: L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dy), N, h, h, cout, cout, kpad, 3, 3, 1, 1, P(dw), force, stream()))
        res[tag] = timeit(f)
    fl = 2.0 * N * h * h * cin * cout * 9
    print(f"{name:8s} {cin:4d}->{cout:4d} {h:3d}^2: wino {res['wino']:7.1f} us ({fl / res['wino'] / 1e6:6.1f} TF alg) | f16x3 {res['f16x3']:7.1f} us "
          f"({fl / res['f16x3'] / 1e6:6.1f} TF alg, {3 * fl / res['f16x3'] / 1e6 / 2500:.3f} of the f16 peak)")
