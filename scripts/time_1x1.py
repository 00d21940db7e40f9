"""Time the 1x1 convs of the EfficientNet-b4 MBConv blocks (BASELINE config 4 shapes, 4x3x1024x1024) through the op entry
points: forward conv, dgrad, wgrad.  Prints us, algorithmic GB/s and TF/s per layer and the per-step totals."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, stream, rup
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
N = 4
# (name, Cin, Cout, H=W, count)
LAYERS = [("exp2", 24, 144, 512, 1), ("exp3-5", 32, 192, 256, 3), ("exp6", 32, 192, 256, 1), ("exp7-9", 56, 336, 128, 3), ("exp10", 56, 336, 128, 1),
          ("exp11-15", 112, 672, 64, 5), ("exp16", 112, 672, 64, 1), ("exp17-21", 160, 960, 64, 5), ("exp22", 160, 960, 64, 1),
          ("exp23-29", 272, 1632, 32, 7), ("exp30", 272, 1632, 32, 1), ("exp31", 448, 2688, 32, 1),
          ("prj0", 48, 24, 512, 1), ("prj1", 24, 24, 512, 1), ("prj2", 144, 32, 256, 1), ("prj3-5", 192, 32, 256, 3), ("prj6", 192, 56, 128, 1),
          ("prj7-9", 336, 56, 128, 3), ("prj10", 336, 112, 64, 1), ("prj11-15", 672, 112, 64, 5), ("prj16", 672, 160, 64, 1),
          ("prj17-21", 960, 160, 64, 5), ("prj22", 960, 272, 32, 1), ("prj23-29", 1632, 272, 32, 7), ("prj30", 1632, 448, 32, 1), ("prj31", 2688, 448, 32, 1)]
# resnet50 bottleneck 1x1s at BASELINE config-3 size (16x3x512x512): python scripts/time_1x1.py r50 [sweep|cfg=N]
R50 = [("l1.c1", 256, 64, 128, 2), ("l1.c3", 64, 256, 128, 3), ("l2.c1a", 256, 128, 128, 1), ("l2.c1", 512, 128, 64, 3), ("l2.c3", 128, 512, 64, 4),
       ("l3.c1a", 512, 256, 64, 1), ("l3.c1", 1024, 256, 32, 5), ("l3.c3", 256, 1024, 32, 6), ("l4.c1a", 1024, 512, 32, 1),
       ("l4.c1", 2048, 512, 16, 2), ("l4.c3", 512, 2048, 16, 3)]
if len(sys.argv) > 1 and sys.argv[1].startswith("r50"):
    LAYERS = R50; N = 16
    sys.argv[1] = sys.argv[1][4:] if len(sys.argv[1]) > 4 else "all"
only = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else None
fcfg = -1
for a_ in sys.argv[2:]:
    if a_.startswith("cfg="): fcfg = int(a_[4:])
sweep = len(sys.argv) > 2 and sys.argv[2] == "sweep"          # forward conv on every implicit-GEMM tile config
tot = [0.0, 0.0, 0.0]
def timeit(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 100      # us
for name, cin, cout, h, cnt in LAYERS:
    if only and name not in only: continue
    x = torch.randn(N, h, h, cin, device=dev); dy = torch.randn(N, h, h, cout, device=dev)
    kpad = rup(cin, 32); kpadd = rup(cout, 32)
    w = torch.randn(cout, kpad, device=dev); wd = torch.randn(cin, kpadd, device=dev)
    y = torch.empty(N, h, h, cout, device=dev); dx = torch.empty(N, h, h, cin, device=dev); dw = torch.zeros(cout, kpad, device=dev)
    stats = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    s0 = src(x)
    f_fwd = lambda: L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(w), cout, kpad, 1, 1, 1, 0, N, cout, None, P(y), P(stats), fcfg, stream()))
    f_dg = lambda: L.check(L.lib().uwm_op_dgrad(P(dy), N, h, h, cout, P(wd), cin, kpadd, 1, 1, 1, 0, h, h, None, None, None, None, P(dx), stream()))
    f_wg = lambda: L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dy), N, h, h, cout, cout, kpad, 1, 1, 1, 0, P(dw), 0, stream()))
    byt = (x.numel() + y.numel()) * 4
    fl = 2.0 * N * h * h * cin * cout
    out = []
    for i, f in enumerate((f_fwd, f_dg, f_wg)):
        us = timeit(f); tot[i] += us * cnt
        out.append(f"{us:7.1f} us {byt / us / 1e3:5.0f} GB/s {fl / us / 1e6:5.1f} TF")
    print(f"{name:9s} {cin:4d}->{cout:4d} {h:3d}^2 x{cnt}: fwd {out[0]} | dgrad {out[1]} | wgrad {out[2]}")
    if sweep:
        res = []
        for cfg in (0, 1, 2, 4, 5):
            f = lambda: L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(w), cout, kpad, 1, 1, 1, 0, N, cout, None, P(y), P(stats), cfg, stream()))
            res.append(f"cfg{cfg} {timeit(f):6.1f}")
        print("          fwd sweep: " + "  ".join(res))
print(f"per step: fwd {tot[0] / 1e3:.2f} ms, dgrad {tot[1] / 1e3:.2f} ms, wgrad {tot[2] / 1e3:.2f} ms")
