"""EfficientNet-b4 encoder bring-up: forward / backward parity of Unet(efficientnet-b4) against the CPU oracle."""
import sys, time
import torch
sys.path.insert(0, ".")
import unet_watermark_amd as U
from oracle import unet_oracle as O

dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "Unet"
n, h, w = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (2, 64, 96)))
ref = O.build("efficientnet-b4", seed=3, arch=arch)
m = getattr(U, arch)("efficientnet-b4").to(dev)
print("keys equal:", list(m.state_dict().keys()) == list(ref.state_dict().keys()), "params", m.num_parameters(),
      sum(p.numel() for p in ref.parameters()))
m.load_state_dict(ref.state_dict())
x, t = O.synthetic_batch(n, h, w, seed=13)
torch.manual_seed(0)
nb = len(ref.encoder._blocks)
keep = (torch.rand(nb, n) > 0.3).float()
for use_keep in (False, "ones", True):
    if use_keep == "ones":
        keep_saved, keep = keep, torch.ones(nb, n)
    elif use_keep is True:
        keep = keep_saved
    m.train(); ref.train()
    ref.zero_grad()
    m.drop_connect = use_keep
    m._keep_override = keep if use_keep else None
    out_ref = ref(x, [keep[i] for i in range(nb)] if use_keep else None)
    crit_ref = O.CombinedLoss([O.BCEWithLogits(), O.DiceLoss(smooth=1e-5)], [0.5, 0.5])
    crit = U.CombinedLoss([U.BCEWithLogitsLoss(), U.DiceLoss(smooth=1e-5)], [0.5, 0.5])
    loss_ref = crit_ref(out_ref, t.unsqueeze(1)); loss_ref.backward()
    out = m(x.to(dev)); loss = crit(out, t.unsqueeze(1).to(dev)); loss.backward()
    torch.cuda.synchronize()
    print(f"[keep={use_keep}] logits max abs err", (out.detach().cpu() - out_ref.detach()).abs().max().item(), "ref absmax",
          out_ref.abs().max().item(), "loss", loss.item(), loss_ref.item())
    gref = dict(ref.named_parameters())
    worst = []
    for name, p in m.named_parameters():
        g, r = p.grad.detach().cpu().double(), gref[name].grad.double()
        if r.norm() == 0:
            worst.append((float(g.norm()), 1.0, name + " (zero ref)")); continue
        l2 = ((g - r).norm() / r.norm()).item()
        cos = ((g.flatten() @ r.flatten()) / (g.norm() * r.norm() + 1e-300)).item()
        worst.append((l2, cos, name))
    worst.sort(reverse=True)
    shown = 0
    for l2, cos, name in worst:
        if name.endswith("_bn2.bias") or shown >= 14:
            continue
        shown += 1
        g, r = dict(m.named_parameters())[name].grad, gref[name].grad
        blk = int(name.split(".")[2]) if "_blocks" in name else -1
        print(f"   {name}: relL2 {l2:.3e} cos {cos:.6f} |g| {g.norm().item():.3e} |ref| {r.norm().item():.3e} keep {keep[blk].tolist() if blk >= 0 else None}")
    # running stats
    bref = dict(ref.named_buffers())
    e = max((b.detach().cpu() - bref[k]).abs().max().item() for k, b in m.named_buffers() if "num_batches" not in k)
    print("   running-stat max abs err", e)
m.eval(); ref.eval()
with torch.no_grad():
    print("eval logits err", (m(x.to(dev)).cpu() - ref(x)).abs().max().item())
