"""LFQQuantizer on the HIP sign/bit-pack kernels (reference pit/quantization/lfq.py:79-228).

Lookup-free quantisation is the closed form of the arg-min over the implicit
{-1,+1}^d codebook: bit_i = (x_i > 0), index = big-endian pack (channel 0 = MSB).
Eval returns the same ``info`` keys as the reference with zero losses
(lfq.py:170-183); the training-only entropy/commit losses are plain torch."""
from __future__ import annotations

from math import log2

import torch
import torch.nn.functional as F
from torch import einsum
from torch.nn import Module

from .. import _lib


def lfq_entropy_loss(logits, temperature=0.01, sample_minimization_weight=1.0, batch_maximization_weight=1.0,
                     eps=1e-5):
    """lfq.py:56-76."""
    probs = F.softmax(logits / temperature, -1)
    log_probs = F.log_softmax(logits / temperature + eps, -1)
    avg_probs = probs.reshape(-1, probs.shape[-1]).mean(0)
    avg_entropy = -torch.sum(avg_probs * torch.log(avg_probs + eps))
    sample_entropy = torch.mean(-torch.sum(probs * log_probs, -1))
    loss = sample_minimization_weight * sample_entropy - batch_maximization_weight * avg_entropy
    return sample_entropy, avg_entropy, loss


class LFQQuantizer(Module):
    def __init__(self, format, codebook_size=None, num_codebooks=1, sample_minimization_weight=1.0,
                 batch_maximization_weight=1.0):
        super().__init__()
        assert format in ["bchw", "blc"]
        self.format = format
        self.codebook_size = codebook_size
        self.codebook_dim = int(log2(codebook_size))
        self.num_codebooks = num_codebooks
        self.sample_minimization_weight = sample_minimization_weight
        self.batch_maximization_weight = batch_maximization_weight
        self.register_buffer("mask", 2 ** torch.arange(self.codebook_dim), persistent=False)
        self.register_buffer("zero", torch.tensor(0.0), persistent=False)
        bits = self.indices_to_bits(torch.arange(codebook_size))
        self.register_buffer("codebook", bits * 2.0 - 1.0, persistent=False)

    def indices_to_bits(self, x):
        mask = 2 ** torch.arange(self.codebook_dim, device=x.device, dtype=torch.long)
        return (x.unsqueeze(-1) & mask) != 0

    def forward(self, x):
        if self.format == "bchw":
            b, c, h, w = x.shape
            xf = x.reshape(b, c, h * w).transpose(1, 2)
        else:
            b, _, c = x.shape
            xf = x
        l = xf.shape[1]
        with torch.no_grad():
            idx, q = _lib.lfq_pack(xf.detach().float().reshape(-1, c).contiguous())
        indices = idx.reshape(b, l, 1)
        q = q.reshape(b, l, c).to(x.dtype)
        if self.training:
            xs = xf.reshape(b, l, self.num_codebooks, -1)
            logits = 2 * einsum("... i d, j d -> ... i j", xs, self.codebook)
            per_sample_entropy, codebook_entropy, entropy_aux_loss = lfq_entropy_loss(
                logits=logits, sample_minimization_weight=self.sample_minimization_weight,
                batch_maximization_weight=self.batch_maximization_weight)
            commit_loss = F.mse_loss(xf, q.detach(), reduction="none").mean()
        else:
            per_sample_entropy = codebook_entropy = entropy_aux_loss = commit_loss = self.zero
        quantized = xf + (q - xf).detach()
        if self.format == "bchw":
            quantized = quantized.transpose(1, 2).reshape(b, c, h, w)
            indices = indices.transpose(1, 2).reshape(b, 1, h, w)
        info = {"indices": indices, "entropy_aux_loss": entropy_aux_loss,
                "per_sample_entropy": per_sample_entropy.detach(), "codebook_entropy": codebook_entropy.detach(),
                "commit_loss": commit_loss}
        return quantized, info

    def dequant(self, indices):
        if self.format == "bchw":
            b, ng, h, w = indices.shape
            ind = indices.reshape(b, ng, h * w).transpose(1, 2)
        else:
            b, _, ng = indices.shape
            ind = indices
        l = ind.shape[1]
        c = self.num_codebooks * self.codebook_dim
        assert c == 16, "the reference unpacks exactly 16 bits (lfq.py:220-222)"
        q = _lib.lfq_unpack(ind.contiguous().reshape(-1), c).reshape(b, l, ng, c)
        if self.format == "bchw":
            # "b (h w) c n -> b (c n) h w" with c = ng, n = bits
            q = q.reshape(b, h, w, ng, c).permute(0, 3, 4, 1, 2).reshape(b, ng * c, h, w)
        return q
