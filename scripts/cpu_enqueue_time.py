import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
import unet_watermark_amd as U
from unet_watermark_amd.train import Trainer
dev = torch.device("cuda:0")
m = U.Unet("resnet34").to(dev)
tr = Trainer(m, w_dice=1.0, w_bce=0.0, smooth=1e-5, lr=1e-4, weight_decay=1e-4)
x = torch.randn(16, 3, 512, 512, device=dev); t = torch.zeros(16, 512, 512, dtype=torch.int64, device=dev); t[:, 100:200, 100:300] = 1
for _ in range(3): tr.step(x, t)
torch.cuda.synchronize()
for n in (1, 3, 6):
    t0 = time.perf_counter()
    for _ in range(n): tr.step(x, t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e3*(t1-t0)/n:.2f} ms/step (CPU), total {1e3*(t2-t0)/n:.2f} ms/step")
