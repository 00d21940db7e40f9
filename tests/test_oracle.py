"""CPU tests of the oracle (test infrastructure): self-checks that stand in for the upstream tests the
reference lacks + the committed golden vectors that pin the oracle itself."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_parameter_counts_and_state_dict_keys():
    m34, m18 = O.build("resnet34"), O.build("resnet18")
    assert sum(p.numel() for p in m34.parameters()) == 24_436_369
    assert sum(p.numel() for p in m18.parameters()) == 14_328_209
    keys = list(m34.state_dict().keys())
    assert keys[0] == "encoder.conv1.weight" and keys[-1] == "segmentation_head.0.bias"
    for k in ("encoder.bn1.num_batches_tracked", "encoder.layer2.0.downsample.0.weight", "encoder.layer4.2.conv2.weight",
              "decoder.blocks.0.conv1.0.weight", "decoder.blocks.4.conv2.1.running_var", "segmentation_head.0.weight"):
        assert k in keys, k
    sd = m34.state_dict()
    assert sd["decoder.blocks.0.conv1.0.weight"].shape == (256, 768, 3, 3)
    assert sd["decoder.blocks.4.conv1.0.weight"].shape == (16, 32, 3, 3)
    assert sd["segmentation_head.0.weight"].shape == (1, 16, 3, 3)
    assert not any(k.startswith("encoder.fc") for k in keys)


def test_unetplusplus_structure():
    """smp.UnetPlusPlus decoder restatement: parameter counts of the public smp models, ModuleDict key order,
    block shapes of the dense grid (the reference's default MODEL.NAME)."""
    m34, m18 = O.build("resnet34", arch="UnetPlusPlus"), O.build("resnet18", arch="UnetPlusPlus")
    assert sum(p.numel() for p in m34.parameters()) == 26_078_609
    assert sum(p.numel() for p in m18.parameters()) == 15_970_449
    sd = m34.state_dict()
    blocks = []
    for k in sd:
        if k.startswith("decoder.blocks.") and k.endswith(".conv1.0.weight"):
            blocks.append(k.split(".")[2])
    assert blocks == ["x_0_0", "x_0_1", "x_1_1", "x_0_2", "x_1_2", "x_2_2", "x_0_3", "x_1_3", "x_2_3", "x_3_3", "x_0_4"]
    shapes = {b: tuple(sd[f"decoder.blocks.{b}.conv1.0.weight"].shape[:2]) for b in blocks}
    assert shapes == {"x_0_0": (256, 768), "x_0_1": (128, 512), "x_1_1": (128, 384), "x_0_2": (64, 320), "x_1_2": (64, 256),
                      "x_2_2": (64, 192), "x_0_3": (32, 320), "x_1_3": (64, 256), "x_2_3": (64, 192), "x_3_3": (64, 128),
                      "x_0_4": (16, 32)}
    assert O.conv_flops("resnet34", 512, 512, arch="UnetPlusPlus") == (147_069_075_456, 439_974_100_992)
    x = torch.randn(1, 3, 64, 96)
    assert m18(x).shape == (1, 1, 64, 96)


def test_resnet50_structure():
    """Bottleneck encoder (unet_watermark_large.yaml): the public smp parameter counts for classes=1."""
    m = O.build("resnet50")
    assert sum(p.numel() for p in m.parameters()) == 32_521_105
    assert sum(p.numel() for p in O.build("resnet50", arch="UnetPlusPlus").parameters()) == 48_985_745
    sd = m.state_dict()
    assert sd["encoder.layer1.0.conv1.weight"].shape == (64, 64, 1, 1) and sd["encoder.layer1.0.conv3.weight"].shape == (256, 64, 1, 1)
    assert sd["encoder.layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["encoder.layer2.0.conv2.weight"].shape == (128, 128, 3, 3)
    assert sd["decoder.blocks.0.conv1.0.weight"].shape == (256, 3072, 3, 3)
    assert m.encoder.out_channels == (3, 64, 256, 512, 1024, 2048)


def test_efficientnet_b4_structure():
    """EfficientNet-b4 encoder restatement (BASELINE config 4): the encoder holds the published efficientnet-b4
    parameter count minus its classifier head (19,341,616 - conv_head 802,816 - bn1 3,584 - fc 1,793,000), smp's
    out_channels, feature strides, the static "same" paddings along the 380-pixel chain and the drop-connect ramp."""
    m = O.build("efficientnet-b4")
    enc = m.encoder
    assert sum(p.numel() for p in enc.parameters()) == 19_341_616 - 802_816 - 3_584 - 1_793_000 == 16_742_216
    assert sum(p.numel() for p in m.parameters()) == 19_419_289
    assert sum(p.numel() for p in O.build("efficientnet-b4", arch="UnetPlusPlus").parameters()) == 20_006_713
    assert enc.out_channels == (3, 48, 32, 56, 160, 448) and len(enc._blocks) == 32
    assert enc._conv_stem.pad == (0, 1)
    pads = {(b._depthwise_conv.kernel_size[0], b._depthwise_conv.stride[0], b._depthwise_conv.pad) for b in enc._blocks}
    assert pads == {(3, 1, (1, 1)), (3, 2, (0, 1)), (5, 2, (2, 2)), (5, 2, (1, 2)), (5, 1, (2, 2))}
    assert abs(enc._blocks[31].drop_rate - 0.2 * 31 / 32) < 1e-12 and enc._blocks[0].drop_rate == 0
    sd = m.state_dict()
    assert sd["encoder._blocks.0._depthwise_conv.weight"].shape == (48, 1, 3, 3)
    assert "encoder._blocks.0._expand_conv.weight" not in sd and sd["encoder._blocks.2._expand_conv.weight"].shape == (144, 24, 1, 1)
    assert sd["encoder._blocks.2._se_reduce.weight"].shape == (6, 144, 1, 1) and sd["encoder._blocks.2._se_reduce.bias"].shape == (6,)
    assert sd["decoder.blocks.0.conv1.0.weight"].shape == (256, 448 + 160, 3, 3)
    assert enc._blocks[2]._bn0.eps == 1e-3 and enc._blocks[2]._bn0.momentum == 0.01
    x = torch.randn(2, 3, 64, 96)
    feats = enc(x)
    assert [tuple(f.shape[1:]) for f in feats] == [(3, 64, 96), (48, 32, 48), (32, 16, 24), (56, 8, 12), (160, 4, 6), (448, 2, 3)]
    # drop-connect: a dropped sample passes through the block unchanged
    blk = enc._blocks[1].train()
    xin = torch.randn(2, 24, 8, 8)
    out = blk(xin, torch.tensor([0.0, 1.0]))
    assert torch.equal(out[0], xin[0]) and not torch.equal(out[1], xin[1])
    assert O.conv_flops("efficientnet-b4", 512, 512) == (36_848_411_328, 110_375_364_672)


def test_conv_flops_match_survey():
    assert O.conv_flops("resnet34", 512, 512) == (62_511_906_816, 186_302_595_072)
    assert O.conv_flops("resnet18", 256, 256) == (10_796_138_496, 32_080_134_144)


def test_input_shape_check():
    m = O.build("resnet18")
    with pytest.raises(RuntimeError, match="divisible by 32"):
        m(torch.zeros(1, 3, 100, 64))


def test_dice_closed_forms():
    dice = O.DiceLoss(smooth=0.0)
    big = 40.0
    x = torch.tensor([[[[big, big], [-big, -big]]]])
    t = torch.tensor([[[[1, 0], [1, 0]]]])
    assert abs(float(dice(x, t)) - 0.5) < 1e-6                 # I=1, sum p=2, sum t=2
    assert abs(float(dice(x, torch.tensor([[[[1, 1], [0, 0]]]])))) < 1e-6        # perfect overlap
    # all-negative batch => loss 0 and zero gradient
    xr = torch.randn(2, 1, 8, 8, requires_grad=True)
    l = O.DiceLoss(smooth=1e-5)(xr, torch.zeros(2, 1, 8, 8, dtype=torch.long))
    l.backward()
    assert float(l) == 0.0 and float(xr.grad.abs().max()) == 0.0
    # the reduction spans batch AND pixels (one Dice for the whole batch)
    x2 = torch.randn(2, 1, 4, 4); t2 = (torch.rand(2, 1, 4, 4) > 0.5).long()
    p = torch.sigmoid(x2)
    exp = 1 - (2 * (p * t2).sum() + 1e-5) / ((p + t2).sum() + 1e-5)
    assert abs(float(O.DiceLoss(smooth=1e-5)(x2, t2)) - float(exp)) < 1e-6


def test_metrics_closed_forms():
    out = torch.tensor([[[0.9, 0.2], [0.6, 0.5]]])            # >= 0.5 -> [[1,0],[1,1]]
    tgt = torch.tensor([[[1, 0], [0, 1]]])
    tp, fp, fn, tn = O.get_stats(out, tgt)
    assert (int(tp), int(fp), int(fn), int(tn)) == (2, 1, 0, 1)
    m = O.micro_metrics(tp, fp, fn, tn)
    assert abs(m["iou"] - 2 / 3) < 1e-12 and abs(m["precision"] - 2 / 3) < 1e-12 and m["recall"] == 1.0
    z = O.compute_metrics(torch.zeros(1, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))
    assert z["iou"] == 1.0 and z["f1"] == 1.0               # zero_division = 1.0
    # predict.py thresholds RAW logits (no sigmoid)
    lg = torch.tensor([[[[0.4, 0.6], [-1.0, 2.0]]]])
    assert O.predict_mask(lg, 0.5).flatten().tolist() == [0, 255, 0, 255]
    assert O.predict_mask(lg, 0.5, apply_sigmoid=True).flatten().tolist() == [255, 255, 0, 255]


@pytest.mark.parametrize("name", ["unet_r18_256", "unet_r18_64_combo", "unet_r34_64", "unetpp_r18_64_combo", "unet_effb4_64"])
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    enc, n, h, w, seed, loss = str(g["encoder"]), int(g["n"]), int(g["h"]), int(g["w"]), int(g["seed"]), str(g["loss"])
    model = O.build(enc, seed=seed, arch=str(g["arch"]) if "arch" in g.files else "Unet")
    model.train()
    x, t = O.synthetic_batch(n, h, w, seed=seed)
    out = model(x)
    assert out.shape == (n, 1, h, w)
    assert int(g["n_params"]) == sum(p.numel() for p in model.parameters())
    assert int(t.sum()) == int(g["target_sum"])
    # summation order inside oneDNN may change with the thread count: fp32-noise tolerances
    assert np.allclose(out.detach()[:, :, :32, :32].numpy(), g["logits_crop"], atol=2e-4)
    assert abs(float(out.double().sum()) - float(g["logits_sum"])) < 1e-3 * float(g["logits_abs_sum"])
    dice = O.DiceLoss(smooth=1e-5)(out, t.unsqueeze(1)); bce = O.BCEWithLogits()(out, t.unsqueeze(1))
    assert abs(float(dice) - float(g["loss_dice"])) < 1e-5 and abs(float(bce) - float(g["loss_bce"])) < 1e-5
    l = dice if loss == "dice" else 0.5 * bce + 0.5 * dice
    l.backward()
    gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
    # (a per-channel shift ahead of a 1x1 conv + BatchNorm has no effect: EfficientNet's _bn2.bias gradients are pure
    # rounding noise, not comparable between two runs)
    live = np.array([not k.endswith("_bn2.bias") for k, _ in model.named_parameters()])
    assert np.allclose(gn[live], g["grad_norm"][live], rtol=3e-2, atol=1e-7)
    assert [k for k, _ in model.named_parameters()] == list(g["param_names"])


def test_kernel_vectors():
    g = np.load(os.path.join(GOLD, "kernels.npz"))
    y = torch.nn.functional.conv2d(torch.from_numpy(g["conv_x"]), torch.from_numpy(g["conv_w"]), None, 1, 1)
    assert np.allclose(y.numpy(), g["conv_y"], atol=1e-5)
    lg = torch.from_numpy(g["loss_logits"]).requires_grad_(); tg = torch.from_numpy(g["loss_target"])
    dice, bce = O.DiceLoss(smooth=1e-5)(lg, tg), O.BCEWithLogits()(lg, tg)
    assert abs(float(dice) - float(g["loss_dice"])) < 1e-6 and abs(float(bce) - float(g["loss_bce"])) < 1e-6
    (0.5 * dice + 0.5 * bce).backward()
    assert np.allclose(lg.grad.numpy(), g["loss_grad"], atol=1e-8)
    # BatchNorm train: biased variance normalises, UNBIASED variance goes to running_var
    x = torch.from_numpy(g["bn_x"]).double()
    mu, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
    yy = (x - mu[None, :, None, None]) / torch.sqrt(var + 1e-5)[None, :, None, None] * torch.from_numpy(g["bn_gamma"]).double()[None, :, None, None] \
        + torch.from_numpy(g["bn_beta"]).double()[None, :, None, None]
    assert np.allclose(yy.numpy(), g["bn_y"], atol=1e-5)
    cnt = x.numel() / x.shape[1]
    assert np.allclose(0.9 * 1.0 + 0.1 * (var * cnt / (cnt - 1)).numpy(), g["bn_running_var"], atol=1e-5)
    assert np.allclose(0.1 * mu.numpy(), g["bn_running_mean"], atol=1e-6)
