"""Host input pipeline of the eval path (reference pit/data.py:74-108).

``SimpleDataset(root, image_size)``: file list = ``root`` as a .txt list, else the sorted recursive
globs ``*.JPEG``, ``*.jpg``, ``*.png`` (in that order); item = ``{"img": float32 [3, S, S] in
[-1, 1], "fpath": str}`` after Resize(S) (shorter edge, bilinear, PIL) -> CenterCrop(S) -> ToTensor
-> Normalize(0.5, 0.5).  torchvision is not required: the three transforms are restated on PIL +
torch with torchvision's size/offset arithmetic (``int(S * long / short)``,
``int(round((h - S) / 2.0))``).  Parity with torchvision itself is unpinned here (not installed).
"""
from __future__ import annotations

from glob import glob
from typing import Dict, List

import numpy as np
import torch
from torch.utils.data import Dataset


def resize_shorter_edge(img, size: int):
    from PIL import Image

    w, h = img.size
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return img
    new_short, new_long = size, int(size * long / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    return img.resize((new_w, new_h), Image.BILINEAR)


def center_crop(img, size: int):
    from PIL import Image

    w, h = img.size
    if w < size or h < size:  # torchvision pads with zeros first
        pad_l, pad_t = max((size - w) // 2, 0), max((size - h) // 2, 0)
        canvas = Image.new(img.mode, (max(w, size), max(h, size)))
        canvas.paste(img, (pad_l, pad_t))
        img, (w, h) = canvas, canvas.size
    top = int(round((h - size) / 2.0))
    left = int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def to_normalized_tensor(img) -> torch.Tensor:
    a = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - 0.5) / 0.5


class SimpleDataset(Dataset):
    def __init__(self, root: str, image_size: int) -> None:
        super().__init__()
        self.image_size = image_size
        if root.endswith(".txt"):
            with open(root) as f:
                self.fpaths: List[str] = [line.strip("\n") for line in f.readlines()]
        else:
            self.fpaths = sorted(glob(root + "/**/*.JPEG", recursive=True))
            self.fpaths += sorted(glob(root + "/**/*.jpg", recursive=True))
            self.fpaths += sorted(glob(root + "/**/*.png", recursive=True))
        assert len(self.fpaths) > 0, "File list is empty. Check the root."

    def __len__(self) -> int:
        return len(self.fpaths)

    def __getitem__(self, index: int) -> Dict[str, object]:
        from PIL import Image

        fpath = self.fpaths[index]
        img = Image.open(fpath).convert("RGB")
        img = center_crop(resize_shorter_edge(img, self.image_size), self.image_size)
        return {"img": to_normalized_tensor(img), "fpath": fpath}
