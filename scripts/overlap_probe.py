"""How much of a short-K fp16x3 convolution launch is exposed prologue / epilogue?  The same layer launched back to back on ONE stream
against two independent copies launched on TWO streams (kernel only, filter bank reused): if the pair finishes in much less than twice
the single time, workgroups of one launch do not cover each other's load -> MFMA -> store chain and a persistent form has room."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, rup
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
N = 16
for name, cin, cout, h in (("layer1", 64, 64, 128), ("layer2", 128, 128, 64), ("layer3", 256, 256, 32), ("dec3.c2", 32, 32, 256)):
    bufs = []
    for _ in range(2):
        x = torch.randn(N, h, h, cin, device=dev); y = torch.empty(N, h, h, cout, device=dev)
        bufs.append((x, y, src(x)))
    w = torch.randn(cout, rup(9 * cin, 32), device=dev) * 0.05
    st = torch.zeros(2 * cout, dtype=torch.float64, device=dev)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def launch(i, s, cfg=1600):
        x, y, s0 = bufs[i]
        L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(w), cout, rup(9 * cin, 32), 3, 3, 1, 1, N, cout, None, P(y), P(st), cfg, C.c_void_p(s.cuda_stream)))
    launch(0, streams[0], 600); torch.cuda.synchronize()          # packs the bank once
    def run(two, reps=30):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[0])
        streams[1].wait_event(e0)
        for _ in range(reps):
            launch(0, streams[0]); launch(1, streams[1] if two else streams[0])
        e2 = torch.cuda.Event(); e2.record(streams[1]); streams[0].wait_event(e2)
        e1.record(streams[0]); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    run(False, 5); run(True, 5)
    a, b = run(False), run(True)
    print(f"{name:8s} {cin:4d}->{cout:4d} {h:3d}^2: two launches on one stream {a:7.1f} us, on two streams {b:7.1f} us ({b / a:.2f}x)", flush=True)
