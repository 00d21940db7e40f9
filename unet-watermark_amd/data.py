"""Minimal data side of the boundary: tensors with the contract of the reference's dataset
(/root/reference/src/utils/dataset.py:113-122,332-333,389-395) — image fp32 NCHW normalised with the
ImageNet mean/std, mask int64 {0,1} (H,W).  The reference's OpenCV/albumentations pipeline is out of scope
(SURVEY.md §2 row 8); a synthetic generator and a plain PIL folder reader are provided."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .predict import IMAGENET_MEAN, IMAGENET_STD


class SyntheticWatermarkDataset(Dataset):
    """Smooth random backgrounds with a semi-transparent rectangular 'watermark'; mask = its footprint."""

    def __init__(self, length=256, img_size=512, seed=42):
        self.length, self.size, self.seed = int(length), int(img_size), int(seed)

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        s = self.size
        base = torch.rand(3, s // 16, s // 16, generator=g)
        img = torch.nn.functional.interpolate(base[None], size=(s, s), mode="bilinear", align_corners=False)[0]
        rh = int(torch.randint(s // 8, s // 2, (), generator=g)); rw = int(torch.randint(s // 8, s // 2, (), generator=g))
        y0 = int(torch.randint(0, s - rh, (), generator=g)); x0 = int(torch.randint(0, s - rw, (), generator=g))
        alpha = 0.3 + 0.4 * float(torch.rand((), generator=g))
        img[:, y0:y0 + rh, x0:x0 + rw] = (1 - alpha) * img[:, y0:y0 + rh, x0:x0 + rw] + alpha
        mask = torch.zeros(s, s, dtype=torch.int64)
        mask[y0:y0 + rh, x0:x0 + rw] = 1
        mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1); std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
        return (img - mean) / std, mask


class FolderDataset(Dataset):
    """<root>/watermarked/*.{png,jpg} + <root>/masks/<same stem>.png (the reference's layout, dataset.py:60-84)."""

    def __init__(self, root, img_size=512):
        from PIL import Image  # noqa: F401
        self.root, self.size = root, int(img_size)
        wd = os.path.join(root, "watermarked")
        self.files = sorted(f for f in os.listdir(wd) if f.lower().endswith((".png", ".jpg", ".jpeg")))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        from PIL import Image
        f = self.files[i]
        img = Image.open(os.path.join(self.root, "watermarked", f)).convert("RGB").resize((self.size, self.size), Image.BILINEAR)
        mpath = os.path.join(self.root, "masks", os.path.splitext(f)[0] + ".png")
        m = Image.open(mpath).convert("L").resize((self.size, self.size), Image.NEAREST)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1); std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
        return (x - mean) / std, torch.from_numpy((np.asarray(m) > 127).astype(np.int64))


# ---------------------------------------------------------------------------- device-side input pipeline
AUG_HFLIP, AUG_VFLIP = 1, 2


def aug_flags(hflip=False, vflip=False, rot90=0) -> int:
    """flag word of one image: HorizontalFlip, VerticalFlip, RandomRotate90(k) — applied in that order."""
    return (AUG_HFLIP if hflip else 0) | (AUG_VFLIP if vflip else 0) | ((int(rot90) & 3) << 2)


def random_aug_flags(n: int, generator=None, p_hflip=0.5, p_vflip=0.2, p_rot90=0.3) -> torch.Tensor:
    """The geometric part of the reference's get_train_transform (dataset.py:378-384 probabilities)."""
    r = torch.rand(n, 3, generator=generator)
    k = torch.randint(1, 4, (n,), generator=generator)
    f = (r[:, 0] < p_hflip).int() | ((r[:, 1] < p_vflip).int() << 1) | (torch.where(r[:, 2] < p_rot90, k, 0).int() << 2)
    return f.to(torch.int32)


def device_preprocess(images_u8: torch.Tensor, masks_u8=None, flags=None, mean=IMAGENET_MEAN, std=IMAGENET_STD,
                      mask_threshold: int = 127):
    """uint8 (N,H,W,C) images [+ uint8 (N,H,W) masks] on the HIP device -> (N,C,H,W) fp32 normalised images
    [+ uint8 {0,1} masks], optional per-image flip/rot90 flags (int32 tensor, see aug_flags): one kernel each —
    the tail of every get_*_transform of the reference (Normalize + ToTensorV2) and its exact geometric
    augmentations, without a host round trip or an fp32 upload."""
    import ctypes as C
    from . import _lib as L
    if images_u8.device.type != "cuda" or images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
        raise RuntimeError("device_preprocess needs a uint8 (N,H,W,C) tensor on a HIP device (no CPU fallback)")
    x = images_u8.contiguous()
    n, h, w, c = x.shape
    fl = None
    if flags is not None:
        fl = flags.to(device=x.device, dtype=torch.int32).contiguous()
        if fl.numel() != n:
            raise ValueError("flags must have one entry per image")
        if h != w and bool(((fl >> 2) & 3).any()):
            raise ValueError("rot90 needs square images")
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    mean_c = (C.c_float * c)(*[float(v) for v in mean[:c]]); std_c = (C.c_float * c)(*[float(v) for v in std[:c]])
    st = C.c_void_p(L.stream_ptr(x.device))
    L.check(L.lib().uwm_preprocess_u8(C.c_void_p(x.data_ptr()), n, h, w, c, mean_c, std_c,
                                      C.c_void_p(fl.data_ptr() if fl is not None else 0), C.c_void_p(out.data_ptr()), st))
    if masks_u8 is None:
        return out
    m = masks_u8.contiguous()
    if m.dtype != torch.uint8 or m.shape != (n, h, w) or m.device != x.device:
        raise ValueError("masks must be uint8 (N,H,W) on the same device")
    mo = torch.empty((n, h, w), dtype=torch.uint8, device=x.device)
    L.check(L.lib().uwm_preprocess_mask_u8(C.c_void_p(m.data_ptr()), n, h, w, int(mask_threshold),
                                           C.c_void_p(fl.data_ptr() if fl is not None else 0), C.c_void_p(mo.data_ptr()), st))
    return out, mo
