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


CFGS = [int(v) for v in sys.argv[1:]] or [600]      # 600 auto | 601 / 602 / 603: conv_f16x3.hip variants | 604 / 605: conv_f16x3v2.hip 64- / 32-channel tiles
for name, cin, cout, h in (("layer1", 64, 64, 128), ("layer2", 128, 128, 64), ("layer3", 256, 256, 32), ("layer4", 512, 512, 16),
                           ("dec2.c2", 64, 64, 128), ("dec1.c1", 384, 128, 64), ("dec0.c1", 768, 256, 32), ("dec3.c1", 128, 32, 256), ("dec3.c2", 32, 32, 256)):
    x = torch.randn(N, h, h, cin, device=dev); y = torch.empty(N, h, h, cout, device=dev)
    w = torch.randn(cout, rup(9 * cin, 32), device=dev) * 0.05
    st = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    s0 = src(x)
    fl = 2.0 * N * h * h * cin * cout * 9
    ref = None
    line = f"{name:8s} {cin:4d}->{cout:4d} {h:3d}^2:"
    for cfg in [-1] + CFGS:
        f = lambda: L.lib().uwm_op_conv(C.byref(s0), None, P(w), cout, rup(9 * cin, 32), 3, 3, 1, 1, N, cout, None, P(y), P(st), cfg, stream())
        y.zero_()
        if f() != 0:
            line += f" | {cfg}: n/a"
            continue
        if cfg >= 600:                                  # kernel only: the bank of the call above is reused (cfg + 1000)
            g = lambda: L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(w), cout, rup(9 * cin, 32), 3, 3, 1, 1, N, cout, None, P(y), P(st), cfg + 1000, stream()))
            t = timeit(g)
        else:
            t = timeit(lambda: L.check(f()))
        if ref is None:
            ref = y.clone()
            line += f" wino {t:6.1f} us ({fl / t / 1e6:5.1f} TF)"
        else:
            err = float((y - ref).abs().max() / ref.abs().max())
            line += f" | {cfg}: {t:6.1f} us ({fl / t / 1e6:5.1f} TF alg, util {3 * fl / t / 1e6 / 2500:.3f}, relerr {err:.1e})"
    print(line + "   (wino: op entry with its filter transform; fp16x3: kernel alone)", flush=True)
