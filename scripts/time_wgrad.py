"""Time wgrad (auto-routed) on named layer shapes: N Cin Cout H W"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, stream
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
shapes = {"layer1": (16, 64, 64, 128, 128), "layer2": (16, 128, 128, 64, 64), "layer3": (16, 256, 256, 32, 32),
          "layer4": (16, 512, 512, 16, 16), "dec0c1": (16, 768, 256, 32, 32), "dec1c1": (16, 384, 128, 64, 64),
          "dec2c1": (16, 192, 64, 128, 128), "dec3c1": (16, 128, 32, 256, 256), "dec4c1": (16, 32, 16, 512, 512), "dec4c2": (16, 16, 16, 512, 512), "dec4c1u": (16, 32, 16, 512, 512), "stem": (16, 3, 64, 512, 512)}
force = int(sys.argv[1]) if len(sys.argv) > 1 else 0
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, (n, cin, cout, h, w) in shapes.items():
    if only and name not in only: continue
    up = 1 if name.endswith("u") else 0
    stem = name == "stem"                      # 7x7 / stride 2 / pad 3, 3 channels stored as 4: h, w are the INPUT size
    k, st, pd = (7, 2, 3) if stem else (3, 1, 1)
    cinp = (cin + 3) // 4 * 4
    x = torch.randn(n, h >> up, w >> up, cinp, device=dev); dy = torch.randn(n, h // st, w // st, cout, device=dev)
    kpad = (k * k * cinp + 31) // 32 * 32
    dw = torch.zeros(cout, kpad, device=dev)
    s0 = src(x, up=up)
    def run():
        L.check(L.lib().uwm_op_wgrad(C.byref(s0), None, P(dy), n, h // st, w // st, cout, cout, kpad, k, k, st, pd, P(dw), force, stream()))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * n * (h // st) * (w // st) * cout * cin * k * k
    print(f"{name:8s} force_igemm={force} {ms*1e3:8.1f} us  {fl/ms/1e9:7.2f} TF/s")
