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
