#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/unet_oracle.py, plain torch fp32).

PARITY UNPINNED at the smp boundary: the reference holds no golden vector / test for this path and
`segmentation_models_pytorch` is not installed here, so these fixtures pin the
ORACLE ITSELF (so that it cannot drift between rounds / images) — not the real smp package.
Re-run:  python tests/golden/make_golden.py   (deterministic: seeds are fixed; needs only torch CPU)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def unet_case(name, enc, n, h, w, seed, loss="dice", arch="Unet"):
    model = O.build(enc, seed=seed, arch=arch)
    model.train()
    x, t = O.synthetic_batch(n, h, w, seed=seed)
    crit = O.DiceLoss(smooth=1e-5) if loss == "dice" else O.CombinedLoss(
        [O.BCEWithLogits(), O.DiceLoss(smooth=1e-5)], [0.5, 0.5])
    out = model(x)
    dice = O.DiceLoss(smooth=1e-5)(out, t.unsqueeze(1))
    bce = O.BCEWithLogits()(out, t.unsqueeze(1))
    l = crit(out, t.unsqueeze(1))
    l.backward()
    names, gnorm, gsample = [], [], []
    g = torch.Generator().manual_seed(123)
    for k, p in model.named_parameters():
        names.append(k)
        gnorm.append(float(p.grad.double().norm()))
        idx = torch.randint(0, p.numel(), (3,), generator=g)
        gsample.append(p.grad.flatten()[idx].numpy())
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.step()
    psample = np.stack([p.detach().flatten()[:3].numpy() if p.numel() >= 3 else
                        np.pad(p.detach().flatten().numpy(), (0, 3 - p.numel())) for p in model.parameters()])
    metrics = O.compute_metrics(torch.sigmoid(out.detach()).squeeze(1), t)
    o = out.detach()
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        encoder=enc, arch=arch, n=n, h=h, w=w, seed=seed, loss=loss,
        logits_crop=o[:, :, :32, :32].numpy(), logits_sum=float(o.double().sum()),
        logits_abs_sum=float(o.double().abs().sum()), loss_total=float(l), loss_dice=float(dice), loss_bce=float(bce),
        param_names=np.array(names), grad_norm=np.array(gnorm), grad_sample=np.stack(gsample),
        param_after_adam=psample, metrics=np.array([metrics[k] for k in ("iou", "f1", "accuracy", "recall", "precision")]),
        mask_sum=int(O.predict_mask(o, 0.5).long().sum()), target_sum=int(t.sum()),
        n_params=sum(p.numel() for p in model.parameters()))


def kernel_cases():
    g = torch.Generator().manual_seed(7)
    d = {}
    # conv_l1: layer1.0.conv1 shape class at reduced size (3x3 s1 p1, 64->64)
    x = torch.randn(1, 64, 12, 12, generator=g); w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    d.update(conv_x=x.numpy(), conv_w=w.numpy(), conv_y=torch.nn.functional.conv2d(x, w, None, 1, 1).numpy())
    # bn_train: batch statistics + normalisation
    y = torch.randn(2, 8, 6, 6, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(8); bn.weight.data = torch.rand(8, generator=g) + 0.5; bn.bias.data = torch.randn(8, generator=g)
    d.update(bn_x=y.numpy(), bn_gamma=bn.weight.detach().numpy(), bn_beta=bn.bias.detach().numpy(),
             bn_y=bn(y).detach().numpy(), bn_running_mean=bn.running_mean.numpy(), bn_running_var=bn.running_var.numpy())
    # dice_bce: logits/targets -> losses and dL/dlogits
    lg = (torch.randn(2, 1, 16, 16, generator=g) * 3).requires_grad_()
    tg = (torch.rand(2, 1, 16, 16, generator=g) > 0.7).long()
    dice = O.DiceLoss(smooth=1e-5)(lg, tg); bce = O.BCEWithLogits()(lg, tg)
    (0.5 * dice + 0.5 * bce).backward()
    d.update(loss_logits=lg.detach().numpy(), loss_target=tg.numpy(), loss_dice=float(dice), loss_bce=float(bce),
             loss_grad=lg.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **d)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "effb4":      # add this one fixture without regenerating the others
        unet_case("unet_effb4_64", "efficientnet-b4", 2, 64, 64, 5, "combo")
        sys.exit(0)
    unet_case("unet_r18_256", "resnet18", 1, 256, 256, 42, "dice")      # BASELINE config 1
    unet_case("unet_r18_64_combo", "resnet18", 2, 64, 64, 7, "combo")
    unet_case("unet_r34_64", "resnet34", 1, 64, 64, 3, "dice")
    unet_case("unetpp_r18_64_combo", "resnet18", 2, 64, 64, 11, "combo", arch="UnetPlusPlus")
    unet_case("unet_effb4_64", "efficientnet-b4", 2, 64, 64, 5, "combo")                       # BASELINE config 4 encoder
    kernel_cases()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
