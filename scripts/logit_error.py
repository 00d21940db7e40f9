"""Logit max-abs error of the HIP path against the fp32 CPU oracle and an fp64 run of it (train- and eval-mode
BatchNorm), with Winograd on and off and in the split-precision forward modes (fp16x3, bf16x3).  Needs an MI355X.  usage: logit_error.py [encoder] [N] [H] [W]"""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_watermark_amd as U
from unet_watermark_amd import _lib as L
from oracle import unet_oracle as O
enc = sys.argv[1] if len(sys.argv) > 1 else "resnet34"
n, h, w = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2, 256, 256)
dev = torch.device("cuda:0")
torch.manual_seed(0)
ref = O.build(enc, seed=42)
m = U.Unet(enc)
m.load_state_dict(ref.state_dict()); m = m.to(dev)
x, t = O.synthetic_batch(n, h, w, seed=7)
ref64 = copy.deepcopy(ref).double()
for mode in ("eval", "train"):      # eval first: the train-mode forwards update the running statistics
    getattr(ref, mode)(); getattr(ref64, mode)(); getattr(m, mode)()
    with torch.no_grad():
        o32 = ref(x); o64 = ref64(x.double())
    for wino, prec in ((1, "f32"), (0, "f32"), (1, "f16x3"), (1, "bf16x3_all")):
        L.check(L.lib().uwm_set_winograd_mode(m._h, wino))
        m.set_precision(prec, min_workgroups=1)             # fill threshold 1: every eligible layer on the split-product kernels
        with torch.no_grad():
            o = m(x.to(dev)).cpu()
        print(f"{enc} {n}x{h}x{w} {mode:5s} winograd={wino} precision={prec:10s}: |hip-oracle32| {float((o - o32).abs().max()):.2e}  "
              f"|hip-oracle64| {float((o.double() - o64).abs().max()):.2e}  |oracle32-oracle64| {float((o32.double() - o64).abs().max()):.2e}  "
              f"logit range {float(o32.abs().max()):.2f}")
