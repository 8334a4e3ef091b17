"""VQQuantizer on the HIP arg-min path (reference pit/quantization/vq.py:8-129).

Same constructor, same ``embedding.weight`` parameter (so checkpoints load), same
``forward -> (z_q, {"indices", "codebook_loss"})`` / ``dequant`` contracts.  The
``[rows, n]`` distance matrix of vq.py:58-69 is never built: the nearest code is
found by the MFMA filter (``-|e|^2 + 2 z.e``) + an fp64 re-rank of its candidates.
Without autograd (eval under ``torch.no_grad()``) the whole forward -- permutes,
per-sub-codebook slices, embedding lookup, straight-through value, the two MSE
means -- is ONE library call (``vq_quantize_z_f32``); with autograd the arg-min
runs in the library (``vq_argmin_f32``) and the differentiable glue in torch."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


class VQQuantizer(nn.Module):
    def __init__(self, format, n, dim, beta=0.25, codebook_num=1, legacy=True):
        super().__init__()
        self.format, self.n, self.dim = format, n, dim
        self.beta, self.legacy, self.codebook_num = beta, legacy, codebook_num
        self.embedding = nn.Embedding(self.n, self.dim)
        self.embedding.weight.data.uniform_(-1.0 / self.n, 1.0 / self.n)
        self._ws = _lib.Workspace()

    def get_trainable_parameters(self):
        return self.embedding.parameters()

    def _to_bhwc(self, t):
        if self.format == "bchw":
            return t.permute(0, 2, 3, 1).contiguous()
        b, l, c = t.shape
        h = int(np.sqrt(l))
        assert h * h == l, "Input length must be a perfect square for blc format"
        return t.reshape(b, h, h, c).contiguous()

    def _from_bhwc(self, t):
        if self.format == "bchw":
            return t.permute(0, 3, 1, 2).contiguous()
        b, h, w, c = t.shape
        return t.reshape(b, h * w, c).contiguous()

    def _forward_fused(self, z):
        """vq.py:39-96 as one call.  A channels_last z (what the NHWC conv stack hands over) is read, and z_q / indices are written,
        in that memory layout: NHWC memory of [B, c, h, w] IS the "blc" layout."""
        z = z.float()
        w = self.embedding.weight.detach().float()
        if self.format == "bchw":
            b, c, h, wd = z.shape
            assert self.dim * self.codebook_num == c
            if not z.is_contiguous() and z.is_contiguous(memory_format=torch.channels_last):
                zmem = z.permute(0, 2, 3, 1).reshape(b, h * wd, c)          # a view of the same memory
                ind, zq, loss = _lib.vq_quantize_z(zmem, w, self.dim, "blc", self.beta, self.legacy, self._ws)
                as_bchw = lambda t: t.view(b, h, wd, -1).permute(0, 3, 1, 2)
                return as_bchw(zq), {"indices": as_bchw(ind), "codebook_loss": loss[0]}
        else:
            b, l, c = z.shape
            h = int(np.sqrt(l))
            assert h * h == l, "Input length must be a perfect square for blc format"
            assert self.dim * self.codebook_num == c
        ind, zq, loss = _lib.vq_quantize_z(z, w, self.dim, self.format, self.beta, self.legacy, self._ws)
        return zq, {"indices": ind, "codebook_loss": loss[0]}

    def forward(self, z):
        if z.is_cuda and not (torch.is_grad_enabled() and (z.requires_grad or self.embedding.weight.requires_grad)):
            return self._forward_fused(z)
        z = self._to_bhwc(z)
        assert self.dim * self.codebook_num == z.shape[-1]
        zf = z.reshape(-1, self.dim, self.codebook_num)  # channel = d*K + k (vq.py:53)
        w = self.embedding.weight
        z_q, indices = [], []
        for k in range(self.codebook_num):
            with torch.no_grad():
                # a learned codebook moves every optimizer step: no per-codebook cache while training (the dim-4 search index would
                # be rebuilt by one block per call); eval with autograd on keeps it
                idx, _ = _lib.vq_argmin(zf[:, :, k].detach().float().contiguous(), w.detach().float(), ws=self._ws,
                                        use_cache=not self.training)
            z_q.append(self.embedding(idx)[:, :, None])
            indices.append(idx[:, None])
        z_q = torch.cat(z_q, dim=2).view(z.shape)
        indices = torch.cat(indices, dim=1).reshape(z.shape[0], z.shape[1], z.shape[2], self.codebook_num)
        if not self.legacy:
            loss = self.beta * torch.mean((z_q.detach() - z) ** 2) + torch.mean((z_q - z.detach()) ** 2)
        else:
            loss = torch.mean((z_q.detach() - z) ** 2) + self.beta * torch.mean((z_q - z.detach()) ** 2)
        z_q = z + (z_q - z).detach()
        return self._from_bhwc(z_q), {"indices": self._from_bhwc(indices), "codebook_loss": loss}

    def dequant(self, indices):
        ind = self._to_bhwc(indices)
        b, h, w, _ = ind.shape
        flat = ind.reshape(-1, self.codebook_num)
        z_q = torch.cat([self.embedding(flat[:, k])[:, :, None] for k in range(self.codebook_num)], dim=2)
        return self._from_bhwc(z_q.reshape(b, h, w, self.dim * self.codebook_num))
