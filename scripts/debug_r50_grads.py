import sys, os
sys.path.insert(0, os.getcwd())
import torch
import unet_watermark_amd as U
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
for (n,h,w) in ((4,128,160),(4,192,192)):
    ref = O.build("resnet50", seed=42); m = U.Unet("resnet50").to(dev); m.load_state_dict(ref.state_dict())
    ref64 = O.build("resnet50", seed=42).double(); ref64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    x, t = O.synthetic_batch(n, h, w, seed=13)
    crit_ref = O.CombinedLoss([O.BCEWithLogits(), O.DiceLoss(smooth=1e-5)], [0.5, 0.5])
    crit = U.CombinedLoss([U.BCEWithLogitsLoss(), U.DiceLoss(smooth=1e-5)], [0.5, 0.5])
    m.train(); ref.train(); ref64.train()
    o = ref(x); crit_ref(o, t.unsqueeze(1)).backward()
    o64 = ref64(x.double()); crit_ref(o64, t.unsqueeze(1)).backward()
    og = m(x.to(dev)); crit(og, t.unsqueeze(1).to(dev)).backward()
    print(n,h,w, "logit err", float((og.detach().cpu()-o.detach()).abs().max()))
    worst = []
    g64 = dict(ref64.named_parameters()); g32 = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        g = p.grad.detach().cpu().double(); r = g32[name].grad.double(); r64 = g64[name].grad
        if r.norm() == 0: continue
        worst.append((float((g-r).norm()/r.norm()), float((g-r64).norm()/r64.norm()), float((r-r64).norm()/r64.norm()), name))
    worst.sort(reverse=True)
    for wv in worst[:6]: print("   hip-vs-32 %.4f  hip-vs-64 %.4f  torch32-vs-64 %.4f  %s" % wv)
