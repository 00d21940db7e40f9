import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_watermark_amd as U
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
enc, n, h, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ref = O.build(enc, seed=42); m = U.Unet(enc).to(dev); m.load_state_dict(ref.state_dict())
ref64 = copy.deepcopy(ref).double()
x, t = O.synthetic_batch(n, h, w, seed=7)
m.train(); ref.train(); ref64.train()
crit_ref = O.CombinedLoss([O.BCEWithLogits(), O.DiceLoss(smooth=1e-5)], [0.5, 0.5])
crit = U.CombinedLoss([U.BCEWithLogitsLoss(), U.DiceLoss(smooth=1e-5)], [0.5, 0.5])
out_ref = ref(x); loss_ref = crit_ref(out_ref, t.unsqueeze(1)); loss_ref.backward()
out64 = ref64(x.double()); loss64 = crit_ref(out64, t.unsqueeze(1)); loss64.backward()
out = m(x.to(dev)); loss = crit(out, t.unsqueeze(1).to(dev)); loss.backward()
print("logits err vs f32", (out.detach().cpu()-out_ref.detach()).abs().max().item(), "ours vs f64", (out.detach().cpu().double()-out64.detach()).abs().max().item(), "f32 vs f64", (out_ref.detach().double()-out64.detach()).abs().max().item())
gref = dict(ref.named_parameters()); g64 = dict(ref64.named_parameters())
wo = wr = 0
for nme, p in m.named_parameters():
    g, r, r64 = p.grad.cpu().double(), gref[nme].grad.double(), g64[nme].grad
    s = max(1e-30, r64.abs().max().item())
    eo, er = (g-r64).abs().max().item()/s, (r-r64).abs().max().item()/s
    wo, wr = max(wo, eo), max(wr, er)
    print(f"{nme:50s} ours-vs-f64 {eo:9.2e}   torchf32-vs-f64 {er:9.2e}")
print("WORST ours", wo, "torch f32", wr)
