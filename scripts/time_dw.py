"""Time the depthwise kernels of mbconv.hip at the EfficientNet-b4 layer shapes of BASELINE config 4 (4x3x1024x1024):
   python scripts/time_dw.py  -> per (layer, pass): us, GB/s of algorithmic traffic."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from unet_watermark_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")
P = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
N = 4
# (mid channels, input H=W, k, stride, pad_begin, how many blocks of this shape)
LAYERS = [(48, 512, 3, 1, 1, 1), (24, 512, 3, 1, 1, 1), (144, 512, 3, 2, 0, 1), (192, 256, 3, 1, 1, 3), (192, 256, 5, 2, 2, 1),
          (336, 128, 5, 1, 2, 3), (336, 128, 3, 2, 0, 1), (672, 64, 3, 1, 1, 5), (672, 64, 5, 1, 2, 1), (960, 64, 5, 1, 2, 5),
          (960, 64, 5, 2, 1, 1), (1632, 32, 5, 1, 2, 7), (1632, 32, 3, 1, 1, 1), (2688, 32, 3, 1, 1, 1)]
tot = [0.0, 0.0, 0.0]
for c, h, k, s, pb, cnt in LAYERS:
    ho = h // s
    x = torch.randn(N, h, h, c, device=dev); dy = torch.randn(N, ho, ho, c, device=dev)
    w = torch.randn(k * k, c, device=dev); y = torch.empty_like(dy); dx = torch.empty_like(x); dw = torch.zeros_like(w)
    scr = torch.empty(max(1, lib.uwm_op_depthwise_scratch_floats(k, N, c, ho, ho)), device=dev)
    calls = [lambda: lib.uwm_op_depthwise(0, P(x), P(w), k, s, pb, N, h, h, c, ho, ho, None, P(y), None, st()),
             lambda: lib.uwm_op_depthwise(1, P(dy), P(w), k, s, pb, N, h, h, c, ho, ho, None, P(dx), None, st()),
             lambda: lib.uwm_op_depthwise(2, P(x), P(dy), k, s, pb, N, h, h, c, ho, ho, None, P(dw), P(scr), st())]
    byt = [(x.numel() + y.numel()) * 4, (x.numel() + y.numel()) * 4, (x.numel() + y.numel()) * 4]
    res = []
    for i, f in enumerate(calls):
        for _ in range(3): L.check(f())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tot[i] += us * cnt
        res.append(f"{us:7.1f} us {byt[i] / us / 1e3:6.0f} GB/s")
    print(f"C={c:5d} {h:4d}^2 k{k} s{s} x{cnt}: fwd {res[0]} | dgrad {res[1]} | wgrad {res[2]}")
print(f"per step: fwd {tot[0] / 1e3:.2f} ms, dgrad {tot[1] / 1e3:.2f} ms, wgrad {tot[2] / 1e3:.2f} ms")
