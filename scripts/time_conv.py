"""Time the forward 3x3 conv on named layer shapes (N Cin Cout H W) for a list of kernel configs.
usage: time_conv.py [cfg,cfg,...] [layer,layer,...]   cfg: -1 auto, 300 Winograd, 100+BN direct patch, 200 patch16.
The op-level Winograd path includes the (small) filter transform launch; TF/s are direct-conv FLOPs / time."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.util import src, P, stream
from unet_watermark_amd import _lib as L
dev = torch.device("cuda:0")
shapes = {"layer1": (16, 64, 64, 128, 128), "layer2": (16, 128, 128, 64, 64), "layer3": (16, 256, 256, 32, 32),
          "layer4": (16, 512, 512, 16, 16), "dec0c1": (16, 768, 256, 32, 32), "dec0c2": (16, 256, 256, 32, 32),
          "dec1c1": (16, 384, 128, 64, 64), "dec2c1": (16, 192, 64, 128, 128), "dec3c1": (16, 128, 32, 256, 256),
          "dec3c2": (16, 32, 32, 256, 256), "dec4c1": (16, 32, 16, 512, 512), "dec4c2": (16, 16, 16, 512, 512),
          "dec4c1u": (16, 32, 16, 512, 512)}     # the real decoder block 4 conv1: input is a 256x256 tensor, nearest-x2 upsampled by the loader
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [300, -2]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, (n, cin, cout, h, w) in shapes.items():
    if only and name not in only: continue
    up = 1 if name.endswith("u") else 0
    x = torch.randn(n, h >> up, w >> up, cin, device=dev)
    kpad = (9 * cin + 31) // 32 * 32
    wt = torch.randn(cout, kpad, device=dev) * 0.05
    y = torch.empty(n, h, w, cout, device=dev)
    s0 = src(x, up=up)
    line = f"{name:8s}"
    for cfg in cfgs:
        c = cfg
        if cfg == -2:   # best direct kernel for the shape
            c = 200 if cin == 16 else 100 + (128 if cout >= 128 else (64 if cout > 32 else (32 if cout > 16 else 16)))
        def run():
            L.check(L.lib().uwm_op_conv(C.byref(s0), None, P(wt), cout, kpad, 3, 3, 1, 1, n, cout, None, P(y), None, c, stream()))
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 2.0 * n * h * w * cout * cin * 9
        line += f" | cfg {c:4d}: {ms*1e3:8.1f} us {fl/ms/1e9:7.2f} TF/s"
    print(line)
