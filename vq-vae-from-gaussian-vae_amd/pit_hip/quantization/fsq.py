"""FSQQuantizer on the HIP kernel (reference pit/quantization/fsq.py:12-103).

Finite scalar quantisation (FSQ paper, appendix A.1): per channel, bound with a shifted tanh,
round to the nearest of `levels[l]` values, mixed-radix pack (channel 0 most significant).
Indices are int32 like the reference."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib


class FSQQuantizer(nn.Module):
    def __init__(self, levels, format):
        super().__init__()
        self.levels = nn.Parameter(torch.tensor(levels, dtype=torch.int32), requires_grad=False)
        self._levels = [int(v) for v in levels]
        self.dim = self.levels.shape[0]
        self.format = format
        assert self.format in ["bchw", "blc"]

    def forward(self, z):
        z = z.float()
        if self.format == "bchw":
            b, c, h, w = z.shape
            zf = z.reshape(b, c, h * w).transpose(1, 2)
        else:
            b, _, c = z.shape
            zf = z
        l = zf.shape[1]
        ndim = l * c
        with torch.no_grad():
            zq, idx = _lib.fsq_quantize(zf.detach().reshape(-1, c).contiguous(), self._levels)
        zq = zq.reshape(b, l, c)
        if zf.requires_grad and torch.is_grad_enabled():
            # round_ste (fsq.py:6-9, :30-38): the VALUE is the rounded branch (the kernel's zq), the GRADIENT is that of
            # bounded_z / half_width = (tanh(z + shift) * half_l - offset) / half_width, recomputed here in torch
            eps = 1e-3
            lev = self.levels.to(zf.device)
            half_l = (lev - 1) * (1 + eps) / 2
            offset = torch.where(lev % 2 == 0, 0.5, 0.0)
            shift = (offset / half_l).atanh()
            soft = ((zf + shift).tanh() * half_l - offset) / (lev // 2)
            zhat = soft + (zq - soft).detach()
        else:
            zhat = zq
        indices = idx.reshape(b, l, 1)
        if self.format == "bchw":
            zhat = zhat.transpose(1, 2).reshape(b, c, h, w)
            indices = indices.transpose(1, 2).reshape(b, 1, h, w)
        info = {"indices": indices, "bits": torch.sum(torch.log2(self.levels)) * ndim}
        return zhat, info

    def dequant(self, indices):
        if self.format == "bchw":
            b, _, h, w = indices.shape
            ind = indices.reshape(b, 1, h * w).transpose(1, 2)
        else:
            b = indices.shape[0]
            ind = indices
        l = ind.shape[1]
        zhat = _lib.fsq_dequant(ind.contiguous().reshape(-1).to(torch.int32), self._levels).reshape(b, l, self.dim)
        if self.format == "bchw":
            zhat = zhat.transpose(1, 2).reshape(b, self.dim, h, w)
        return zhat
