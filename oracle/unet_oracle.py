"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

Plain-torch (CPU, fp32, torch.nn / torch.nn.functional) restatement of the hot
path that the reference delegates to the third-party package
`segmentation-models-pytorch` (pinned only as `>=0.3.0`,
/root/reference/requirements.txt:17):

  * smp.Unet(resnet18|resnet34) forward  <- src/models/unet_model.py:64-71,93-120
  * smp.losses.DiceLoss(mode='binary')   <- src/utils/losses.py:18-19
  * nn.BCEWithLogitsLoss / CombinedLoss  <- src/utils/losses.py:22-23,33-52
  * smp.metrics.get_stats + micro scores <- src/utils/metrics.py:11-37
  * the train step order                 <- src/train.py:82-107
  * the predict-time logit threshold     <- src/predict.py:610-625

PARITY UNPINNED at the smp boundary: the smp source is not in /root/reference
and not installed in this image, and the reference has no test / golden vector
for this path.  The graph below follows the published smp
algorithm; every primitive (conv2d, batch_norm, relu,
max_pool2d, nearest interpolate, cat, logsigmoid, autograd, Adam) is stock
torch CPU fp32, which is the numerical ground truth.  Self-checks that stand in
for missing upstream tests live in tests/test_oracle.py (parameter counts,
state_dict key set, Dice closed forms, all-negative batch => 0).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

# resnet block counts: torchvision.models.resnet{18,34} (BasicBlock), resnet50 (Bottleneck, expansion 4)
_RESNET_BLOCKS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3)}
_RESNET_EXPANSION = {"resnet18": 1, "resnet34": 1, "resnet50": 4}
_RESNET_WIDTHS = (64, 128, 256, 512)


class BasicBlock(nn.Module):
    """torchvision BasicBlock: conv3x3-bn-relu-conv3x3-bn (+downsample) -add-relu."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.bn2(self.conv2(o))
        return self.relu(o + idn)


class Bottleneck(nn.Module):
    """torchvision Bottleneck (v1.5: the stride sits on the 3x3): 1x1-bn-relu, 3x3(stride)-bn-relu, 1x1(x4)-bn
    (+downsample) -add-relu.  The encoder of the reference's large config
    (/root/reference/src/configs/unet_watermark_large.yaml:7-10, ENCODER_NAME resnet50)."""

    def __init__(self, cin: int, planes: int, stride: int):
        super().__init__()
        cout = planes * 4
        self.conv1 = nn.Conv2d(cin, planes, 1, 1, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, cout, 1, 1, 0, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride, 0, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        o = self.bn3(self.conv3(o))
        return self.relu(o + idn)


class ResNetEncoder(nn.Module):
    """smp ResNetEncoder (fc/avgpool deleted); features at strides 1,2,4,8,16,32."""

    def __init__(self, name: str, in_channels: int = 3):
        super().__init__()
        blocks = _RESNET_BLOCKS[name]
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        exp = _RESNET_EXPANSION[name]
        for li, (w, nb) in enumerate(zip(_RESNET_WIDTHS, blocks), start=1):
            layer = []
            for b in range(nb):
                stride = 2 if (b == 0 and li > 1) else 1
                layer.append(BasicBlock(cin, w, stride) if exp == 1 else Bottleneck(cin, w, stride))
                cin = w * exp
            setattr(self, f"layer{li}", nn.Sequential(*layer))
        self.out_channels = (in_channels, 64, 64 * exp, 128 * exp, 256 * exp, 512 * exp)
        # torchvision ResNet init (encoder_weights=None)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        f0 = x
        f1 = self.relu(self.bn1(self.conv1(x)))
        f2 = self.layer1(self.maxpool(f1))
        f3 = self.layer2(f2)
        f4 = self.layer3(f3)
        f5 = self.layer4(f4)
        return [f0, f1, f2, f3, f4, f5]


# ----------------------------------------------------------------------------- EfficientNet-b4 encoder (config 4)
# efficientnet_pytorch (vendored by smp) restated: SURVEY.md Appendix A.7.  [smp-knowledge; source not in the container]
_EFFB4_STAGES = [  # (repeats, kernel, stride, expand, in_ch, out_ch) after width 1.4 / depth 1.8 scaling
    (2, 3, 1, 1, 48, 24), (4, 3, 2, 6, 24, 32), (4, 5, 2, 6, 32, 56), (6, 3, 2, 6, 56, 112),
    (6, 5, 1, 6, 112, 160), (8, 5, 2, 6, 160, 272), (2, 3, 1, 6, 272, 448)]
_EFFB4_IMAGE = 380            # global_params.image_size: the STATIC "same" paddings are computed along this size chain


def _static_same_pad(size, k, s):
    """Conv2dStaticSamePadding: total = max((ceil(size/s)-1)*s + k - size, 0) -> (begin, end) = (t//2, t - t//2)."""
    o = -(-size // s)
    t = max((o - 1) * s + k - size, 0)
    return t // 2, t - t // 2, o


class _SamePadConv(nn.Conv2d):
    """Conv2dStaticSamePadding: a Conv2d (state_dict key `<name>.weight`) behind a fixed asymmetric zero pad."""

    def __init__(self, cin, cout, k, stride, groups, pad, bias=False):
        super().__init__(cin, cout, k, stride, 0, groups=groups, bias=bias)
        self.pad = pad                                  # (begin, end), same for H and W

    def forward(self, x):
        b, e = self.pad
        return super().forward(F.pad(x, (b, e, b, e)) if (b or e) else x)


def _swish(x):
    return x * torch.sigmoid(x)


class MBConv(nn.Module):
    def __init__(self, cin, cout, k, stride, expand, size, drop_rate):
        super().__init__()
        self.cin, self.cout, self.stride, self.expand, self.drop_rate = cin, cout, stride, expand, drop_rate
        mid = cin * expand
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, mid, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, eps=1e-3, momentum=0.01)
        b, e, self.out_size = _static_same_pad(size, k, stride)
        self._depthwise_conv = _SamePadConv(mid, mid, k, stride, mid, (b, e))
        self._bn1 = nn.BatchNorm2d(mid, eps=1e-3, momentum=0.01)
        nsq = max(1, int(cin * 0.25))
        self._se_reduce = nn.Conv2d(mid, nsq, 1)
        self._se_expand = nn.Conv2d(nsq, mid, 1)
        self._project_conv = nn.Conv2d(mid, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01)

    def forward(self, inputs, keep=None):
        """keep: per-sample {0,1} tensor (N,) for drop-connect in training (None = no drop)."""
        x = inputs
        if self.expand != 1:
            x = _swish(self._bn0(self._expand_conv(x)))
        x = _swish(self._bn1(self._depthwise_conv(x)))
        sq = F.adaptive_avg_pool2d(x, 1)
        sq = self._se_expand(_swish(self._se_reduce(sq)))
        x = torch.sigmoid(sq) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.cin == self.cout:
            if keep is not None:
                x = x / (1.0 - self.drop_rate) * keep.view(-1, 1, 1, 1).to(x.dtype)
            x = x + inputs
        return x


class EfficientNetB4Encoder(nn.Module):
    """smp EfficientNetEncoder('efficientnet-b4'): features after the stem and after blocks 6, 10, 22, 32."""

    def __init__(self, in_channels: int = 3):
        super().__init__()
        b, e, size = _static_same_pad(_EFFB4_IMAGE, 3, 2)
        self._conv_stem = _SamePadConv(in_channels, 48, 3, 2, 1, (b, e))
        self._bn0 = nn.BatchNorm2d(48, eps=1e-3, momentum=0.01)
        blocks = []
        for rep, k, s, ex, ci, co in _EFFB4_STAGES:
            for r in range(rep):
                idx = len(blocks)
                blk = MBConv(ci if r == 0 else co, co, k, s if r == 0 else 1, ex, size, 0.2 * idx / 32)
                size = blk.out_size
                blocks.append(blk)
        self._blocks = nn.ModuleList(blocks)
        self.stage_idxs = (6, 10, 22, 32)
        self.out_channels = (in_channels, 48, 32, 56, 160, 448)

    def forward(self, x, keeps=None):
        feats = [x]
        x = _swish(self._bn0(self._conv_stem(x)))
        feats.append(x)
        for i, blk in enumerate(self._blocks):
            x = blk(x, None if keeps is None else keeps[i])
            if i + 1 in self.stage_idxs:
                feats.append(x)
        return feats


def _conv2d_relu(cin, cout):
    # smp.base.modules.Conv2dReLU(use_batchnorm=True): Sequential(conv(no bias), bn, relu)
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=False))


class DecoderBlock(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        self.conv1 = _conv2d_relu(cin + cskip, cout)
        self.conv2 = _conv2d_relu(cout, cout)

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)  # upsampled FIRST, skip SECOND
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]          # (512,256,128,64,64)
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.blocks = nn.ModuleList(
            [DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels)])
        for m in self.modules():                         # smp initialize_decoder
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, *features):
        feats = features[1:][::-1]
        x, skips = feats[0], feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class UnetPlusPlusDecoder(nn.Module):
    """smp.decoders.unetplusplus.decoder.UnetPlusPlusDecoder (attention None, center False): dense grid of the
    same DecoderBlocks, keyed x_{depth}_{layer} in a ModuleDict (state_dict keys `decoder.blocks.x_0_0.conv1.0.weight`
    ...).  [smp-knowledge; source not in the container — SURVEY.md Appendix A]  It is what the reference builds by
    default: MODEL.NAME = "UnetPlusPlus" (/root/reference/src/configs/config.py:15, src/models/unet_model.py:19)."""

    def __init__(self, encoder_channels, decoder_channels):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]          # (512,256,128,64,64)
        self.in_channels = [enc[0]] + list(decoder_channels[:-1])
        self.skip_channels = enc[1:] + [0]
        self.out_channels = list(decoder_channels)
        blocks = {}
        for layer_idx in range(len(self.in_channels) - 1):
            for depth_idx in range(layer_idx + 1):
                if depth_idx == 0:
                    in_ch = self.in_channels[layer_idx]
                    skip_ch = self.skip_channels[layer_idx] * (layer_idx + 1)
                    out_ch = self.out_channels[layer_idx]
                else:
                    out_ch = self.skip_channels[layer_idx]
                    skip_ch = self.skip_channels[layer_idx] * (layer_idx + 1 - depth_idx)
                    in_ch = self.skip_channels[layer_idx - 1]
                blocks[f"x_{depth_idx}_{layer_idx}"] = DecoderBlock(in_ch, skip_ch, out_ch)
        blocks[f"x_{0}_{len(self.in_channels) - 1}"] = DecoderBlock(self.in_channels[-1], 0, self.out_channels[-1])
        self.blocks = nn.ModuleDict(blocks)
        self.depth = len(self.in_channels) - 1
        for m in self.modules():                         # smp initialize_decoder
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, *features):
        features = features[1:][::-1]
        dense_x = {}
        for layer_idx in range(len(self.in_channels) - 1):
            for depth_idx in range(self.depth - layer_idx):
                if layer_idx == 0:
                    output = self.blocks[f"x_{depth_idx}_{depth_idx}"](features[depth_idx], features[depth_idx + 1])
                    dense_x[f"x_{depth_idx}_{depth_idx}"] = output
                else:
                    dense_l_i = depth_idx + layer_idx
                    cat_features = [dense_x[f"x_{idx}_{dense_l_i}"] for idx in range(depth_idx + 1, dense_l_i + 1)]
                    cat_features = torch.cat(cat_features + [features[dense_l_i + 1]], dim=1)
                    dense_x[f"x_{depth_idx}_{dense_l_i}"] = self.blocks[f"x_{depth_idx}_{dense_l_i}"](
                        dense_x[f"x_{depth_idx}_{dense_l_i - 1}"], cat_features)
        dense_x[f"x_{0}_{self.depth}"] = self.blocks[f"x_{0}_{self.depth}"](dense_x[f"x_{0}_{self.depth - 1}"])
        return dense_x[f"x_{0}_{self.depth}"]


class OracleUnet(nn.Module):
    """smp.Unet(encoder_name, encoder_depth=5, encoder_weights=None, decoder_channels,
    in_channels, classes, activation=None) — SURVEY.md Appendix A.1-A.4."""

    _DECODER = UnetDecoder

    def __init__(self, encoder_name="resnet34", encoder_depth=5, encoder_weights=None,
                 decoder_use_batchnorm=True, decoder_channels=(256, 128, 64, 32, 16),
                 decoder_attention_type=None, in_channels=3, classes=1, activation=None,
                 aux_params=None):
        super().__init__()
        if encoder_name not in _RESNET_BLOCKS and encoder_name != "efficientnet-b4":
            raise ValueError(f"oracle supports {list(_RESNET_BLOCKS) + ['efficientnet-b4']}; got {encoder_name}")
        if encoder_depth != 5 or len(decoder_channels) != encoder_depth:
            raise ValueError("decoder_channels length must equal encoder_depth (=5)")
        if encoder_weights is not None or decoder_attention_type is not None \
                or activation is not None or aux_params is not None or not decoder_use_batchnorm:
            raise ValueError("oracle: unsupported option")
        self.encoder = EfficientNetB4Encoder(in_channels) if encoder_name == "efficientnet-b4" else ResNetEncoder(encoder_name, in_channels)
        self.decoder = self._DECODER(self.encoder.out_channels, tuple(decoder_channels))
        head = nn.Conv2d(decoder_channels[-1], classes, 3, 1, 1)
        nn.init.xavier_uniform_(head.weight)             # smp initialize_head
        nn.init.constant_(head.bias, 0)
        self.segmentation_head = nn.Sequential(head, nn.Identity(), nn.Identity())

    def forward(self, x, keeps=None):
        """keeps: efficientnet only — per-block per-sample drop-connect keep masks (list of (N,) tensors or None)."""
        h, w = x.shape[-2:]
        if h % 32 != 0 or w % 32 != 0:
            raise RuntimeError(
                f"Wrong input shape height={h}, width={w}. Expected image height and width "
                f"divisible by 32.")
        feats = self.encoder(x, keeps) if keeps is not None else self.encoder(x)
        return self.segmentation_head(self.decoder(*feats))


class OracleUnetPlusPlus(OracleUnet):
    """smp.UnetPlusPlus(...) with the same constructor surface: only the decoder differs."""
    _DECODER = UnetPlusPlusDecoder


# ----------------------------------------------------------------------------- losses
class DiceLoss(nn.Module):
    """smp.losses.DiceLoss(mode='binary', from_logits=True, log_loss=False, eps=1e-7)
    (reference call site src/utils/losses.py:18-19, smooth=cfg.LOSS.SMOOTH)."""

    def __init__(self, mode="binary", smooth=0.0, eps=1e-7):
        super().__init__()
        assert mode == "binary"
        self.smooth, self.eps = float(smooth), float(eps)

    def forward(self, y_pred, y_true):
        bs = y_true.size(0)
        p = F.logsigmoid(y_pred).exp()
        t = y_true.view(bs, 1, -1).type_as(p)
        p = p.view(bs, 1, -1)
        inter = torch.sum(p * t, dim=(0, 2))
        card = torch.sum(p + t, dim=(0, 2))
        score = (2.0 * inter + self.smooth) / (card + self.smooth).clamp_min(self.eps)
        loss = (1.0 - score) * (t.sum((0, 2)) > 0).to(score.dtype)
        return loss.mean()


class BCEWithLogits(nn.Module):
    """nn.BCEWithLogitsLoss() with the float cast the int64 masks need."""

    def forward(self, y_pred, y_true):
        return F.binary_cross_entropy_with_logits(y_pred, y_true.type_as(y_pred))


class CombinedLoss(nn.Module):
    """src/utils/losses.py:33-52 — sum_i w_i * loss_i."""

    def __init__(self, losses, weights=None):
        super().__init__()
        self.losses = list(losses)
        self.weights = list(weights) if weights else [1.0] * len(self.losses)

    def forward(self, pred, target):
        total = 0
        for fn, w in zip(self.losses, self.weights):
            total = total + w * fn(pred, target)
        return total


# ---------------------------------------------------------------------------- metrics
def get_stats(output: torch.Tensor, target: torch.Tensor, threshold=0.5):
    """smp.metrics.get_stats(mode='binary', threshold) -> tp, fp, fn, tn int64 (N,1)."""
    n = output.shape[0]
    o = (output >= threshold).reshape(n, 1, -1).to(torch.int64)
    t = target.reshape(n, 1, -1).to(torch.int64)
    tp = (o * t).sum(2)
    fp = o.sum(2) - tp
    fn = t.sum(2) - tp
    tn = o.shape[2] - tp - fp - fn
    return tp, fp, fn, tn


def _div(a, b):
    a, b = float(a), float(b)
    return 1.0 if b == 0 else a / b           # zero_division=1.0


def micro_metrics(tp, fp, fn, tn):
    """src/utils/metrics.py:22-35 with reduction='micro'."""
    tp, fp, fn, tn = (int(v.sum()) for v in (tp, fp, fn, tn))
    return {
        "iou": _div(tp, tp + fp + fn),
        "f1": _div(2 * tp, 2 * tp + fp + fn),
        "accuracy": _div(tp + tn, tp + fp + fn + tn),
        "recall": _div(tp, tp + fn),
        "precision": _div(tp, tp + fp),
    }


def compute_metrics(output, target):
    return micro_metrics(*get_stats(output, target, 0.5))


def predict_mask(logits: torch.Tensor, threshold=0.5, apply_sigmoid=False):
    """src/predict.py:614-625 thresholds RAW LOGITS (no sigmoid) -> uint8 {0,255}."""
    v = torch.sigmoid(logits) if apply_sigmoid else logits
    return ((v > threshold).to(torch.uint8) * 255)


# ------------------------------------------------------------------------ synthetic IO
def synthetic_batch(n: int, h: int, w: int, seed: int = 42, in_channels: int = 3):
    """BASELINE.md §3 inputs: x = randn(N,C,H,W); int64 masks with one seeded filled
    rectangle of 5-20 % area per image."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, in_channels, h, w, generator=g)
    t = torch.zeros(n, h, w, dtype=torch.int64)
    for i in range(n):
        frac = 0.05 + 0.15 * float(torch.rand((), generator=g))
        asp = 0.5 + float(torch.rand((), generator=g))
        rh = max(1, min(h, int(round((frac * h * w * asp) ** 0.5))))
        rw = max(1, min(w, int(round(frac * h * w / rh))))
        y0 = int(torch.randint(0, h - rh + 1, (), generator=g))
        x0 = int(torch.randint(0, w - rw + 1, (), generator=g))
        t[i, y0:y0 + rh, x0:x0 + rw] = 1
    return x, t


def build(encoder_name="resnet34", seed=42, arch="Unet", **kw) -> OracleUnet:
    torch.manual_seed(seed)
    return {"Unet": OracleUnet, "UnetPlusPlus": OracleUnetPlusPlus}[arch](encoder_name=encoder_name, **kw)


def train_step(model, criterion, optimizer, images, masks):
    """CPU branch of src/train.py:86,100-105: zero_grad, fwd, unsqueeze, loss, bwd, step."""
    optimizer.zero_grad()
    outputs = model(images)
    if masks.dim() == 3:
        masks = masks.unsqueeze(1)
    loss = criterion(outputs, masks)
    loss.backward()
    optimizer.step()
    return outputs.detach(), loss.detach()


def conv_flops(encoder_name="resnet34", h=512, w=512, decoder_channels=(256, 128, 64, 32, 16),
               in_channels=3, classes=1, arch="Unet"):
    """Algorithmic conv FLOPs per image: (fwd, fwd+bwd) — SURVEY.md §8(d)."""
    macs = []          # (macs, needs_dgrad)
    def conv(cin, cout, k, ho, wo, dgrad=True):
        macs.append((cin * cout * k * k * ho * wo, dgrad))
    if encoder_name == "efficientnet-b4":       # dense 1x1 / stem convs, depthwise convs (cin = 1 per group) and the SE FCs
        hh, ww = h // 2, w // 2
        conv(in_channels, 48, 3, hh, ww, dgrad=False)
        for rep, k, st, ex, ci, co in _EFFB4_STAGES:
            for r in range(rep):
                bc, bs = (ci, st) if r == 0 else (co, 1)
                mid = bc * ex
                if ex != 1:
                    conv(bc, mid, 1, hh, ww)
                hh, ww = hh // bs, ww // bs
                conv(1, mid, k, hh, ww)
                nsq = max(1, int(bc * 0.25))
                conv(mid, nsq, 1, 1, 1)
                conv(nsq, mid, 1, 1, 1)
                conv(mid, co, 1, hh, ww)
        enc = [448, 160, 56, 32, 48]
    else:
        conv(in_channels, 64, 7, h // 2, w // 2, dgrad=False)
        cin, hh, ww = 64, h // 4, w // 4
        exp = _RESNET_EXPANSION[encoder_name]
        for li, (wd, nb) in enumerate(zip(_RESNET_WIDTHS, _RESNET_BLOCKS[encoder_name]), start=1):
            for b in range(nb):
                s = 2 if (b == 0 and li > 1) else 1
                if exp == 1:
                    hh, ww = hh // s, ww // s
                    conv(cin, wd, 3, hh, ww)
                    conv(wd, wd, 3, hh, ww)
                else:
                    conv(cin, wd, 1, hh, ww)
                    hh, ww = hh // s, ww // s
                    conv(wd, wd, 3, hh, ww)
                    conv(wd, wd * exp, 1, hh, ww)
                if s != 1 or cin != wd * exp:
                    conv(cin, wd * exp, 1, hh, ww)
                cin = wd * exp
        enc = [512 * exp, 256 * exp, 128 * exp, 64 * exp, 64]
    in_ch = [enc[0]] + list(decoder_channels[:-1])
    skip = enc[1:] + [0]
    if arch == "Unet":
        for i, s, o in zip(in_ch, skip, decoder_channels):
            hh, ww = hh * 2, ww * 2
            conv(i + s, o, 3, hh, ww)
            conv(o, o, 3, hh, ww)
    else:                                   # UnetPlusPlus: block x_{d}_{L} runs at 1/2^(4-L) resolution
        for lay in range(4):
            for dep in range(lay + 1):
                if dep == 0:
                    ci, cs, co = in_ch[lay], skip[lay] * (lay + 1), decoder_channels[lay]
                else:
                    ci, cs, co = skip[lay - 1], skip[lay] * (lay + 1 - dep), skip[lay]
                bh, bw = h >> (4 - lay), w >> (4 - lay)
                conv(ci + cs, co, 3, bh, bw)
                conv(co, co, 3, bh, bw)
        conv(in_ch[4], decoder_channels[4], 3, h, w)
        conv(decoder_channels[4], decoder_channels[4], 3, h, w)
        hh, ww = h, w
    conv(decoder_channels[-1], classes, 3, hh, ww)
    fwd = 2 * sum(m for m, _ in macs)
    bwd = 2 * sum(m * (2 if d else 1) for m, d in macs)
    return fwd, fwd + bwd
