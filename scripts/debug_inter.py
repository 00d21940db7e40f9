"""Compare every planned workspace intermediate (activations and gradients) with the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import unet_watermark_amd as U
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
enc, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ref = O.build(enc, seed=42).double(); m = U.Unet(enc).to(dev); m.load_state_dict(O.build(enc, seed=42).state_dict())
x, t = O.synthetic_batch(n, h, w, seed=7)
m.train(); ref.train()
acts = {}
def hook(name):
    def f(mod, inp, out):
        out.retain_grad(); acts[name] = out
    return f
blocks = []
for name, mod in ref.named_modules():
    if isinstance(mod, nn.Conv2d): mod.register_forward_hook(hook("y:" + name))
    if isinstance(mod, O.BasicBlock): blocks.append(name); mod.register_forward_hook(hook("xn:%d" % (len(blocks)-1)))
    if isinstance(mod, nn.MaxPool2d): mod.register_forward_hook(hook("pool"))
crit_ref = O.CombinedLoss([O.BCEWithLogits(), O.DiceLoss(smooth=1e-5)], [0.5, 0.5])
crit = U.CombinedLoss([U.BCEWithLogitsLoss(), U.DiceLoss(smooth=1e-5)], [0.5, 0.5])
out_ref = ref(x.double()); crit_ref(out_ref, t.unsqueeze(1)).backward()
out = m(x.to(dev)); crit(out, t.unsqueeze(1).to(dev)).backward()
torch.cuda.synchronize()
def cmp(key, reft, kind):
    buf = m.debug_buffer(key).cpu().double()
    r = reft.permute(0, 2, 3, 1).reshape(-1)
    if buf.numel() != r.numel(): print(key, "SIZE MISMATCH", buf.numel(), r.numel()); return
    err = (buf - r).abs().max().item() / max(1e-30, r.abs().max().item())
    l2 = ((buf - r).norm() / max(1e-30, r.norm())).item()
    nbad = ((buf - r).abs() > 1e-3 * r.abs().max()).sum().item()
    print(f"{kind} {key:45s} max-rel {err:9.2e}  l2 {l2:9.2e}  nbad {nbad}")
for key, a in acts.items():
    if key.startswith("y:segmentation"): continue
    cmp(key, a.detach(), "act ")
for key, a in acts.items():
    if key.startswith("y:segmentation"): continue
    if key.startswith("y:"): cmp("g:" + key[2:], a.grad, "grad")
    elif key.startswith("xn:"): cmp("gx:" + key[3:], a.grad * (a.detach() > 0), "grad")
    elif key == "pool": cmp("g_pool", a.grad, "grad")
# ---- isolate single steps using OUR buffers as inputs (fp64 on CPU)
import torch.nn.functional as F
sd = {k: v.double() for k, v in O.build(enc, seed=42).state_dict().items()}
def buf(key, shape):  # NHWC -> NCHW double
    nn_, hh, ww, cc = shape
    return m.debug_buffer(key).cpu().double().reshape(nn_, hh, ww, cc).permute(0, 3, 1, 2)
hh, ww = h // 32, w // 32
dy1 = buf("g:encoder.layer4.1.conv1", (n, hh, ww, 512))
dz7 = buf("gx:7", (n, hh, ww, 512))
x6 = buf("xn:6", (n, hh, ww, 512))
mine = buf("gx:6", (n, hh, ww, 512))
W1 = sd["encoder.layer4.1.conv1.weight"]
exp = (F.conv_transpose2d(dy1, W1, None, 1, 1) + dz7) * (x6 > 0)
d = (mine - exp).abs()
print("STEP gx:6 from own inputs: max-rel", (d.max() / exp.abs().max()).item(), "l2", ((mine - exp).norm() / exp.norm()).item(), "nbad", (d > 1e-4 * exp.abs().max()).sum().item())
bad = (d > 1e-4 * exp.abs().max()).nonzero()
print(bad[:20])
# oracle's own chain in double for the same step
a = acts["xn:6"]; r = a.grad * (a.detach() > 0)
e2 = (F.conv_transpose2d(acts["y:encoder.layer4.1.conv1"].grad, W1, None, 1, 1) + acts["xn:7"].grad * (acts["xn:7"].detach() > 0)) * (a.detach() > 0)
print("oracle self-consistency", ((e2 - r).abs().max() / r.abs().max()).item())
