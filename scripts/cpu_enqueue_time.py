"""Host enqueue time per train step (one thread issues every launch of a step): the launch-bound risk of 8 ranks x 16 cores.
Enqueue time = wall time until the last launch of N back-to-back steps has been ISSUED (no sync) / N, measured after the queue
has drained, next to the GPU time per step.  Unet-resnet34 512x512 bs16 and Unet-efficientnet-b4 1024x1024 bs4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_watermark_amd as U
from unet_watermark_amd.train import Trainer
dev = torch.device("cuda:0")
for enc, n, s in (("resnet34", 16, 512), ("efficientnet-b4", 4, 1024)):
    m = U.Unet(enc).to(dev)
    tr = Trainer(m, w_dice=1.0, w_bce=0.0, smooth=1e-5, lr=1e-4, weight_decay=1e-4)
    x = torch.randn(n, 3, s, s, device=dev); t = torch.zeros(n, s, s, dtype=torch.int64, device=dev); t[:, 100:200, 100:300] = 1
    for _ in range(3): tr.step(x, t)
    torch.cuda.synchronize()
    for k in (1, 3):
        t0 = time.perf_counter()
        for _ in range(k): tr.step(x, t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"Unet-{enc} {s}x{s} bs{n}: {k} step(s): host enqueue {1e3*(t1-t0)/k:.2f} ms/step, GPU-complete {1e3*(t2-t0)/k:.2f} ms/step")
    del m, tr, x, t
    torch.cuda.empty_cache()
