"""Binary segmentation metrics on libuwm's confusion-count kernel.

Counterpart of /root/reference/src/utils/metrics.py:11-37 (smp.metrics.get_stats(mode='binary',
threshold=0.5) + micro-reduced iou/f1/accuracy/recall/precision, zero_division=1.0) and of the
predict-time threshold at /root/reference/src/predict.py:614-625.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .losses import _logit_plane, _target_plane


def get_stats(output: torch.Tensor, target: torch.Tensor, threshold: float = 0.5, apply_sigmoid: bool = False):
    """-> tp, fp, fn, tn int64 tensors of shape (N,1) on the device (smp.metrics.get_stats layout)."""
    if output.device.type != "cuda":
        raise RuntimeError("uwm metrics run only on a HIP device (no CPU fallback)")
    if target.dtype.is_floating_point and target.dtype != torch.float32:
        raise TypeError("target must be an integer (or float32) tensor")
    x, xp, ld = _logit_plane(output.detach().float())
    n, h, w = x.shape
    t, tp_, tdt = _target_plane(target, n, h * w)
    out = torch.empty((n, 4), dtype=torch.int64, device=x.device)
    with L.on_device(x):
        L.check(L.lib().uwm_stats(C.c_void_p(xp), ld, C.c_void_p(tp_), tdt, n, h * w, float(threshold),
                                  int(apply_sigmoid), C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr(x.device))))
    return out[:, 0:1], out[:, 1:2], out[:, 2:3], out[:, 3:4]


def _div(a: int, b: int) -> float:
    return 1.0 if b == 0 else a / b


def micro_scores(tp, fp, fn, tn) -> dict:
    s = torch.stack([tp.sum(), fp.sum(), fn.sum(), tn.sum()]).cpu().tolist()     # one D2H
    tp, fp, fn, tn = (int(v) for v in s)
    return {"iou": _div(tp, tp + fp + fn), "f1": _div(2 * tp, 2 * tp + fp + fn),
            "accuracy": _div(tp + tn, tp + fp + fn + tn), "recall": _div(tp, tp + fn),
            "precision": _div(tp, tp + fp)}


def get_metrics():
    """Returns compute_metrics(output_probabilities, target) like the reference's get_metrics()."""
    def compute_metrics(output, target):
        return micro_scores(*get_stats(output, target, 0.5, False))
    return compute_metrics


def logits_metrics(logits, target, threshold: float = 0.5):
    """sigmoid + threshold fused (what train.py:110-115 computes in two steps)."""
    return micro_scores(*get_stats(logits, target, threshold, True))


def _soft_sums(pred: torch.Tensor, target: torch.Tensor):
    """Σ pred·target, Σ pred, Σ target over every element, accumulated in fp64 (one D2H for the three)."""
    p = pred.reshape(-1)
    t = target.reshape(-1).to(p.dtype if p.dtype.is_floating_point else torch.float32)
    p = p.to(t.dtype)
    s = torch.stack([(p * t).sum(dtype=torch.float64), p.sum(dtype=torch.float64), t.sum(dtype=torch.float64)])
    return [float(v) for v in s.cpu()]


def dice_coef(pred, target, smooth: float = 1e-5) -> float:
    """Soft Dice coefficient (2·Σpt + s) / (Σp + Σt + s) over the flattened tensors —
    /root/reference/src/utils/metrics.py:39-45.  `pred` = probabilities (or a hard mask), any device."""
    i, sp, st = _soft_sums(pred, target)
    return (2.0 * i + smooth) / (sp + st + smooth)


def iou_score(pred, target, smooth: float = 1e-5) -> float:
    """Soft IoU (Σpt + s) / (Σp + Σt − Σpt + s) — /root/reference/src/utils/metrics.py:47-54."""
    i, sp, st = _soft_sums(pred, target)
    return (i + smooth) / (sp + st - i + smooth)


def threshold_mask(logits: torch.Tensor, threshold: float = 0.5, apply_sigmoid: bool = False) -> torch.Tensor:
    """(N,1,H,W)|(N,H,W) logits -> uint8 {0,255} (N,H,W).  Default reproduces the reference's quirk of
    thresholding RAW logits (src/predict.py:624); apply_sigmoid=True is watermark_filter.py's form."""
    if logits.device.type != "cuda":
        raise RuntimeError("uwm threshold runs only on a HIP device (no CPU fallback)")
    x, xp, ld = _logit_plane(logits.detach())
    n, h, w = x.shape
    out = torch.empty((n, h, w), dtype=torch.uint8, device=x.device)
    with L.on_device(x):
        L.check(L.lib().uwm_threshold(C.c_void_p(xp), ld, n * h * w, float(threshold), int(apply_sigmoid),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(L.stream_ptr(x.device))))
    return out


def resize_threshold(logits: torch.Tensor, size, threshold: float = 0.5, apply_sigmoid: bool = False, return_resized: bool = False):
    """(N,1,h,w)|(N,h,w) logits -> uint8 {0,255} masks (N,H,W) at `size`=(H,W): the bilinear resize to the original
    image size + threshold of /root/reference/src/predict.py:620-625 (cv2.resize INTER_LINEAR, then `> THRESHOLD`)."""
    if logits.device.type != "cuda":
        raise RuntimeError("uwm resize_threshold runs only on a HIP device (no CPU fallback)")
    x, xp, ld = _logit_plane(logits.detach())
    n, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    out = torch.empty((n, H, W), dtype=torch.uint8, device=x.device)
    rs = torch.empty((n, H, W), dtype=torch.float32, device=x.device) if return_resized else None
    with L.on_device(x):
        L.check(L.lib().uwm_resize_threshold(C.c_void_p(xp), ld, n, h, w, H, W, float(threshold), int(apply_sigmoid),
                                             C.c_void_p(out.data_ptr()), C.c_void_p(rs.data_ptr() if rs is not None else 0),
                                             C.c_void_p(L.stream_ptr(x.device))))
    return (out, rs) if return_resized else out
