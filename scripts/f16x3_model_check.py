import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_watermark_amd as U
from oracle import unet_oracle as O
from unet_watermark_amd.train import Trainer
dev = torch.device("cuda:0")
ref = O.build("resnet34", seed=42)
x, t = O.synthetic_batch(2, 256, 256, seed=42)
ref.train()
o_ref = ref(x); l_ref = O.DiceLoss(smooth=1e-5)(o_ref, t.unsqueeze(1)); l_ref.backward()
gref = dict(ref.named_parameters())
for mode in ("f32", "f16x3", "f16x3_all"):
    m = U.Unet("resnet34").to(dev); m.load_state_dict(ref.state_dict()); m.train(); m.set_precision(mode)
    o = m(x.to(dev)); l = U.DiceLoss(mode="binary", smooth=1e-5)(o, t.unsqueeze(1).to(dev)); l.backward()
    err = float((o.detach().cpu() - o_ref.detach()).abs().max())
    cmin = 1.0
    for n_, p_ in m.named_parameters():
        g1, g2 = p_.grad.detach().cpu().double().flatten(), gref[n_].grad.double().flatten()
        cmin = min(cmin, float(g1 @ g2 / (g1.norm() * g2.norm())))
    print(mode, "logit err vs oracle %.3e" % err, "loss diff %.2e" % abs(float(l) - float(l_ref)), "min grad cos %.6f" % cmin)
torch.manual_seed(0)
xb = torch.randn(16, 3, 512, 512, device=dev); tb = (torch.rand(16, 512, 512, device=dev) > 0.8).to(torch.uint8)
for mode in ("f32", "f16x3", "f16x3_all"):
    m = U.Unet("resnet34").to(dev); m.set_precision(mode)
    tr = Trainer(m, lr=1e-4)
    for _ in range(10): tr.step(xb, tb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): tr.step(xb, tb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "train img/s %.1f  ms/step %.3f" % (16 * 40 / dt, 1e3 * dt / 40))
    m.eval()
    with torch.no_grad():
        for _ in range(5): m(xb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m(xb)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "eval forward img/s %.1f" % (16 * 20 / dt))
