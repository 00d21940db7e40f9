"""Side by side: (ours vs fp64 oracle) and (torch fp32 oracle vs fp64 oracle) for every intermediate gradient."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import unet_watermark_amd as U
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
enc, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lossname = sys.argv[5] if len(sys.argv) > 5 else "combo"
def run(dtype):
    ref = O.build(enc, seed=42).to(dtype); ref.train(); acts = {}
    def hook(name):
        def f(mod, inp, out):
            out.retain_grad(); acts[name] = out
        return f
    nb = 0
    for name, mod in ref.named_modules():
        if isinstance(mod, nn.Conv2d): mod.register_forward_hook(hook("y:" + name))
        if isinstance(mod, O.BasicBlock): mod.register_forward_hook(hook("xn:%d" % nb)); nb += 1
        if isinstance(mod, nn.MaxPool2d): mod.register_forward_hook(hook("pool"))
    crit = O.CombinedLoss([O.BCEWithLogits(), O.DiceLoss(smooth=1e-5)], [0.5, 0.5]) if lossname == "combo" else O.DiceLoss(smooth=1e-5)
    x, t = O.synthetic_batch(n, h, w, seed=7)
    crit(ref(x.to(dtype)), t.unsqueeze(1)).backward()
    return acts
a64, a32 = run(torch.float64), run(torch.float32)
m = U.Unet(enc).to(dev); m.load_state_dict(O.build(enc, seed=42).state_dict()); m.train()
x, t = O.synthetic_batch(n, h, w, seed=7)
crit = U.CombinedLoss([U.BCEWithLogitsLoss(), U.DiceLoss(smooth=1e-5)], [0.5, 0.5]) if lossname == "combo" else U.DiceLoss(smooth=1e-5)
crit(m(x.to(dev)), t.unsqueeze(1).to(dev)).backward(); torch.cuda.synchronize()
def refgrad(acts, key):
    a = acts[key]
    if key.startswith("xn:"): return a.grad * (a.detach() > 0)
    return a.grad
def mykey(key):
    return "g:" + key[2:] if key.startswith("y:") else ("gx:" + key[3:] if key.startswith("xn:") else "g_pool")
print(f"{'buffer':45s} ours-l2   torch32-l2   mask-flips ours/torch32")
for key in a64:
    if key.startswith("y:segmentation"): continue
    r64 = refgrad(a64, key).double(); r32 = refgrad(a32, key).double()
    mine = m.debug_buffer(mykey(key)).cpu().double().reshape(r64.permute(0,2,3,1).shape).permute(0,3,1,2)
    lo = ((mine - r64).norm() / r64.norm()).item(); lt = ((r32 - r64).norm() / r64.norm()).item()
    fl = ""
    if key.startswith("xn:"):
        mx = m.debug_buffer(key).cpu().reshape(r64.permute(0,2,3,1).shape).permute(0,3,1,2)
        fo = ((mx > 0) != (a64[key].detach() > 0)).sum().item(); ft = ((a32[key].detach() > 0) != (a64[key].detach() > 0)).sum().item()
        fl = f"{fo}/{ft}"
    print(f"{key:45s} {lo:9.2e} {lt:9.2e}   {fl}")
