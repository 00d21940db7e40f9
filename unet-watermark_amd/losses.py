"""Dice / BCE-with-logits / CombinedLoss on libuwm's fused HIP loss kernel.

Counterparts of /root/reference/src/utils/losses.py:11-52 and of the smp.losses.DiceLoss /
nn.BCEWithLogitsLoss objects it constructs (SURVEY.md §8 a12,a13, Appendix A.5).  The call protocol
is the reference's: `criterion(outputs (N,1,H,W) float, masks (N,1,H,W) int64|uint8|float) -> 0-d
tensor`, then `.backward()` / `.item()` (/root/reference/src/train.py:94-107).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import model as _model


def _logit_plane(logits: torch.Tensor):
    """(N,1,H,W) or (N,H,W) logits -> (tensor kept alive, data_ptr, element stride ld)."""
    if logits.dim() == 4:
        if logits.shape[1] != 1:
            raise ValueError(f"binary losses take (N,1,H,W) logits, got {tuple(logits.shape)}")
        x = logits[:, 0]
    elif logits.dim() == 3:
        x = logits
    else:
        raise ValueError(f"bad logits shape {tuple(logits.shape)}")
    n, h, w = x.shape
    s = x.stride()
    ld = s[2] if w > 1 else 1
    if not (ld >= 1 and s[1] == w * ld and s[0] == h * w * ld):
        x = x.contiguous()
        ld = 1
    return x, x.data_ptr(), int(ld)


def _target_plane(target: torch.Tensor, n: int, hw: int):
    if target.numel() != n * hw:
        raise ValueError(f"target has {target.numel()} elements, logits have {n * hw}")
    t = target.contiguous()
    return t, t.data_ptr(), L.target_dtype_code(t)


class _DiceBCEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, w_dice, w_bce, smooth, eps):
        if logits.device.type != "cuda":
            raise RuntimeError("uwm losses run only on a HIP device (no CPU fallback)")
        if logits.dtype != torch.float32:
            raise TypeError(f"uwm losses take float32 logits, got {logits.dtype}")
        x, xp, ld = _logit_plane(logits.detach())
        n, h, w = x.shape
        t, tp, tdt = _target_plane(target, n, h * w)
        npix = n * h * w
        scratch = torch.empty(8, dtype=torch.float64, device=x.device)
        out = torch.empty(3, dtype=torch.float32, device=x.device)
        need_grad = logits.requires_grad
        cp = 4
        dl = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device) if need_grad else None
        with L.on_device(x):
            L.check(L.lib().uwm_loss(C.c_void_p(xp), ld, C.c_void_p(tp), tdt, npix, float(w_dice), float(w_bce),
                                     float(smooth), float(eps), C.c_void_p(scratch.data_ptr()),
                                     C.c_void_p(out.data_ptr()), C.c_void_p(dl.data_ptr() if need_grad else 0), cp, 1.0,
                                     C.c_void_p(L.stream_ptr(x.device))))
        ctx.dl = dl
        ctx.in_shape = tuple(logits.shape)
        return out[0].clone()

    @staticmethod
    def backward(ctx, gout):
        dl = ctx.dl
        if dl is None:
            return (None,) * 6
        n, h, w, cp = dl.shape
        g = dl[..., :1].permute(0, 3, 1, 2)          # (N,1,H,W) view of the padded buffer
        if not (gout.numel() == 1 and float(gout) == 1.0):
            g = g * gout
        else:
            _model._PADDED_PTRS.clear()
            _model._PADDED_PTRS.add(dl.data_ptr())
        if len(ctx.in_shape) == 3:
            g = g[:, 0]
        return g, None, None, None, None, None


def _fused(logits, target, w_dice, w_bce, smooth, eps):
    return _DiceBCEFunction.apply(logits, target, w_dice, w_bce, smooth, eps)


class DiceLoss(nn.Module):
    """smp.losses.DiceLoss(mode='binary', smooth, from_logits=True, log_loss=False, eps=1e-7)."""

    def __init__(self, mode: str = "binary", classes=None, log_loss: bool = False, from_logits: bool = True,
                 smooth: float = 0.0, ignore_index=None, eps: float = 1e-7):
        super().__init__()
        if mode != "binary":
            raise ValueError(f"DiceLoss mode={mode!r} is not supported (binary only)")
        if log_loss or not from_logits or ignore_index is not None or classes is not None:
            raise ValueError("DiceLoss: only from_logits=True, log_loss=False, no ignore_index/classes")
        self.mode, self.smooth, self.eps = mode, float(smooth), float(eps)

    def forward(self, y_pred, y_true):
        return _fused(y_pred, y_true, 1.0, 0.0, self.smooth, self.eps)


class BCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss() (mean reduction); integer masks are cast, as SURVEY.md a13 notes."""

    def __init__(self, weight=None, reduction: str = "mean", pos_weight=None):
        super().__init__()
        if weight is not None or pos_weight is not None or reduction != "mean":
            raise ValueError("BCEWithLogitsLoss: only the default (unweighted, mean) form is supported")

    def forward(self, y_pred, y_true):
        return _fused(y_pred, y_true, 0.0, 1.0, 0.0, 1e-7)


class CombinedLoss(nn.Module):
    """/root/reference/src/utils/losses.py:33-52 — sum_i weights[i] * losses[i](pred, target).
    Dice and BCE members are evaluated by ONE fused kernel pass."""

    def __init__(self, losses, weights=None):
        super().__init__()
        self.losses = list(losses)
        self.weights = list(weights) if weights else [1.0] * len(self.losses)

    def forward(self, pred, target):
        w_d = w_b = 0.0
        smooth, eps = 0.0, 1e-7
        rest = []
        n_dice = 0
        for fn, w in zip(self.losses, self.weights):
            if isinstance(fn, DiceLoss) and n_dice == 0:
                w_d += w; smooth, eps = fn.smooth, fn.eps; n_dice += 1
            elif isinstance(fn, BCEWithLogitsLoss):
                w_b += w
            else:
                rest.append((fn, w))
        total = _fused(pred, target, w_d, w_b, smooth, eps) if (w_d != 0.0 or w_b != 0.0) else 0
        for fn, w in rest:
            total = total + w * fn(pred, target)
        return total


def get_loss_function(cfg):
    """Counterpart of get_loss_function (/root/reference/src/utils/losses.py:11-31).  `CombinedLoss`
    (named by the reference's text-watermark YAML but unhandled there) is wired to
    LOSS.BCE_WEIGHT / LOSS.DICE_WEIGHT (/root/reference/src/configs/config.py:61-62)."""
    name = cfg.LOSS.NAME
    mode = getattr(cfg.LOSS, "MODE", "binary")
    smooth = getattr(cfg.LOSS, "SMOOTH", getattr(cfg.LOSS, "DICE_SMOOTH", 1e-5))
    if name == "DiceLoss":
        return DiceLoss(mode=mode, smooth=smooth)
    if name == "BCEWithLogitsLoss":
        return BCEWithLogitsLoss()
    if name == "CombinedLoss":
        return CombinedLoss([BCEWithLogitsLoss(), DiceLoss(mode=mode, smooth=smooth)],
                            [float(getattr(cfg.LOSS, "BCE_WEIGHT", 0.5)), float(getattr(cfg.LOSS, "DICE_WEIGHT", 0.5))])
    raise ValueError(f"不支持的损失函数: {name}")
