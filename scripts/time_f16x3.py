"""Kernel-at-a-time gate for conv_f16x3.hip (fp16x3 direct 3x3 on v_mfma_f32_16x16x32_f16) against the fp32 Winograd kernels,
forward convolution with BatchNorm statistics, resnet34 / decoder shapes at 16 x 512^2 (uwm_op_conv; the op entry's filter-bank
launches are subtracted by timing them alone)."""
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
    return e0.elapsed_time(e1) * 1e3 / reps      # us


for name, cin, cout, h in (("layer1", 64, 64, 128), ("layer2", 128, 128, 64), ("layer3", 256, 256, 32), ("layer4", 512, 512, 16),
                           ("dec2.c2", 64, 64, 128), ("dec1.c1", 384, 128, 64), ("dec0.c1", 768, 256, 32)):
    x = torch.randn(N, h, h, cin, device=dev); y = torch.empty(N, h, h, cout, device=dev)
    w = torch.randn(cout, rup(9 * cin, 32), device=dev) * 0.05
    st = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    s0 = src(x)
    res = {}
    for tag, cfg in (("wino", -1), ("f16x3", int(sys.argv[1]) if len(sys.argv) > 1 else 600)):      # 601 / 602 / 603: force a kernel variant
        f = lambda: L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(w), cout, rup(9 * cin, 32), 3, 3, 1, 1, N, cout, None, P(y), P(st), cfg, stream()))
        res[tag] = timeit(f)
    fl = 2.0 * N * h * h * cin * cout * 9
    print(f"{name:8s} {cin:4d}->{cout:4d} {h:3d}^2: wino {res['wino']:7.1f} us ({fl / res['wino'] / 1e6:6.1f} TF alg) | f16x3 {res['f16x3']:7.1f} us "
          f"({fl / res['f16x3'] / 1e6:6.1f} TF alg, {3 * fl / res['f16x3'] / 1e6 / 2500:.3f} of the f16 peak)   (both include their filter-bank launch)")
