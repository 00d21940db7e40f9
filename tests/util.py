"""Shared helpers for the parity tests (layout packing between torch NCHW/OIHW and libuwm NHWC)."""
import ctypes as C

import torch

from unet_watermark_amd import _lib as L


def rup(v, a):
    return (v + a - 1) // a * a


def nhwc(x, cp=None):
    """(N,C,H,W) -> contiguous (N,H,W,CP) with zero channel padding."""
    n, c, h, w = x.shape
    cp = cp or rup(c, 4)
    out = torch.zeros(n, h, w, cp, dtype=x.dtype, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 1)
    return out.contiguous()


def nchw(x, c=None):
    """(N,H,W,CP) -> (N,C,H,W)"""
    c = c or x.shape[-1]
    return x[..., :c].permute(0, 3, 1, 2).contiguous()


def pack_w(w, cinp=None):
    """OIHW -> [O][Kpad] with k = (r*kw+s)*CinP + c"""
    o, i, kh, kw = w.shape
    cinp = cinp or rup(i, 4)
    kpad = rup(kh * kw * cinp, 32)
    out = torch.zeros(o, kpad, dtype=w.dtype, device=w.device)
    tmp = torch.zeros(o, kh, kw, cinp, dtype=w.dtype, device=w.device)
    tmp[..., :i] = w.permute(0, 2, 3, 1)
    out[:, : kh * kw * cinp] = tmp.reshape(o, -1)
    return out.contiguous(), kpad


def unpack_w(wp, o, i, kh, kw, cinp=None):
    cinp = cinp or rup(i, 4)
    return wp[:, : kh * kw * cinp].reshape(o, kh, kw, cinp)[..., :i].permute(0, 3, 1, 2).contiguous()


def src(t, scale=None, shift=None, relu=0, up=0):
    """uwm_src for an NHWC tensor (keeps nothing alive: caller holds the tensors)."""
    n, h, w, c = t.shape
    return L.uwm_src(t.data_ptr(), scale.data_ptr() if scale is not None else None,
                     shift.data_ptr() if shift is not None else None, c, h, w, up, relu)


def P(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
