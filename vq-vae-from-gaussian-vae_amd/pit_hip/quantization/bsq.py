"""BSQQuantizer on the HIP sign/bit-pack kernels (reference pit/quantization/bsq.py:39-156).

Binary spherical quantisation: L2-normalise the channel vector, take signs, scale by
1/sqrt(embed_dim); the index packs the 16 sign bits big-endian (bsq.py:95-99 hard-codes 16).
Eval losses are zeros like the reference (bsq.py:113-118)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import _lib
from .lfq import LFQQuantizer


def bsq_entropy_loss(x, embed_dim, temperature=0.01, sample_minimization_weight=1.0,
                     batch_maximization_weight=1.0, eps=1e-5):
    """bsq.py:14-36."""
    probs = torch.sigmoid(-4 * x / (embed_dim ** 0.5) / temperature)
    probs = torch.stack([probs, 1 - probs], dim=-1)
    log_probs = torch.log(probs + eps)
    avg_probs = probs.reshape(-1, probs.shape[-2], probs.shape[-1]).mean(0)
    avg_entropy = -torch.sum(avg_probs * torch.log(avg_probs + eps))
    sample_entropy = torch.mean(-torch.sum(probs * log_probs, [-2, -1]))
    loss = sample_minimization_weight * sample_entropy - batch_maximization_weight * avg_entropy
    return sample_entropy, avg_entropy, loss


class BSQQuantizer(LFQQuantizer):
    def __init__(self, format, codebook_size, num_codebooks=1, sample_minimization_weight=1.0,
                 batch_maximization_weight=1.0):
        super().__init__(format=format, codebook_size=codebook_size, num_codebooks=num_codebooks,
                         sample_minimization_weight=sample_minimization_weight,
                         batch_maximization_weight=batch_maximization_weight)
        self.embed_dim = self.codebook_dim * num_codebooks

    def forward(self, x):
        if self.format == "bchw":
            b, c, h, w = x.shape
            xf = x.reshape(b, c, h * w).transpose(1, 2)
        else:
            b, _, c = x.shape
            xf = x
        l = xf.shape[1]
        xf = F.normalize(xf, dim=-1)
        q_scale = 1.0 / (self.embed_dim ** 0.5)
        d = self.codebook_dim
        assert self.num_codebooks == 16, "the reference packs exactly 16 codebook bits (bsq.py:97)"
        xs = xf.reshape(b, l, self.num_codebooks, d)
        # pack over the codebook axis for every d: rows = (b, l, d), bits = the 16 codebooks
        rows = xs.detach().float().permute(0, 1, 3, 2).reshape(-1, self.num_codebooks).contiguous()
        with torch.no_grad():
            idx, q = _lib.lfq_pack(rows)
        indices = idx.reshape(b, l, d)
        q = q.reshape(b, l, d, self.num_codebooks).permute(0, 1, 3, 2).to(x.dtype)
        quantized = (xs + (q - xs).detach()) * q_scale
        if self.training:
            per_sample_entropy, codebook_entropy, entropy_aux_loss = bsq_entropy_loss(
                x=xs, embed_dim=self.embed_dim, sample_minimization_weight=self.sample_minimization_weight,
                batch_maximization_weight=self.batch_maximization_weight)
        else:
            per_sample_entropy = codebook_entropy = entropy_aux_loss = self.zero
        quantized = quantized.reshape(b, l, c)
        if self.format == "bchw":
            quantized = quantized.transpose(1, 2).reshape(b, c, h, w)
            indices = indices.transpose(1, 2).reshape(b, d, h, w)
        info = {"indices": indices, "entropy_aux_loss": entropy_aux_loss,
                "per_sample_entropy": per_sample_entropy.detach(), "codebook_entropy": codebook_entropy.detach()}
        return quantized, info

    def dequant(self, indices):
        if self.format == "bchw":
            b, ng, h, w = indices.shape
            ind = indices.reshape(b, ng, h * w).transpose(1, 2)
        else:
            b, _, ng = indices.shape
            ind = indices
        l = ind.shape[1]
        q = _lib.lfq_unpack(ind.contiguous().reshape(-1), 16).reshape(b, l, ng, 16)
        q = q * (1.0 / (self.embed_dim ** 0.5))
        if self.format == "bchw":
            q = q.reshape(b, h, w, ng, 16).permute(0, 3, 4, 1, 2).reshape(b, ng * 16, h, w)
        return q
