"""Torch-CPU restatement of the reference's ``backend="torch"`` arithmetic (TEST INFRASTRUCTURE).

This is the reference's own CPU path -- pit/quantization/gaussian.py:136-150: eight row chunks, per chunk
``Normal(mu, std).log_prob(codebook) - normal_log_prob * beta`` as a ``[chunk, n, dim]`` fp32 tensor, ``sum(dim=2)``,
``argmax(dim=1)`` -- written against the same torch calls, so that ``bench.py``'s ``cpu_baseline`` times what the
reference itself would spend on the host cores.  Like everything under ``oracle/`` it may only be
imported by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py:cpu_baseline``; the product never calls it.

Pinned: ``tests/test_oracle.py`` checks it against the golden vectors captured from the imported reference
(``tests/golden``) and against the C oracle, index for index.
"""
from __future__ import annotations

import torch
from torch.distributions import Normal


def normal_log_prob(cb: torch.Tensor) -> torch.Tensor:
    """gaussian.py:51-52: Normal(0, 1).log_prob(prior_samples), fp32."""
    dim = cb.shape[1]
    return Normal(torch.zeros([1, dim]), torch.ones([1, dim])).log_prob(cb).float()


@torch.no_grad()
def argmax_rows(mu: torch.Tensor, std: torch.Tensor, cb: torch.Tensor, beta: float = 1.0, nlp: torch.Tensor = None):
    """mu, std [rows, dim] fp32, cb [n, dim] fp32 -> (indices int64 [rows], zhat fp32 [rows, dim]).
    gaussian.py:136-150 (the chunk count never changes a per-row result; rows < 8 raise there, one chunk here)."""
    rows = mu.shape[0]
    nlp = normal_log_prob(cb) if nlp is None else nlp
    bs = max(rows // 8, 1)
    zhat = torch.zeros_like(mu)
    indices = torch.zeros([rows], dtype=torch.long)
    for i in range(0, rows, bs):
        q = Normal(mu[i:i + bs][:, None, :], std[i:i + bs][:, None, :])
        perturbed = torch.sum(q.log_prob(cb[None]) - nlp[None] * beta, dim=2)
        arg = torch.argmax(perturbed, dim=1)
        zhat[i:i + bs] = torch.index_select(cb, 0, arg)
        indices[i:i + bs] = arg
    return indices, zhat


@torch.no_grad()
def gq1_forward(z: torch.Tensor, cb: torch.Tensor, group: int, beta: float = 1.0, logvar_range=(-30.0, 20.0)):
    """GaussianQuantRegularizer.forward, eval branch, format bchw (gaussian.py:61-81,120-160) without the
    ``zhat_noquant`` draw: z [B, 2c, h, w] -> (zhat [B, c, h, w], indices [B, K, h, w])."""
    z = z.float()
    b, c2, h, w = z.shape
    c, l = c2 // 2, h * w
    zf = z.reshape(b, c2, l).transpose(1, 2)
    mu, logvar = zf.chunk(2, 2)
    std = torch.exp(0.5 * torch.clamp(logvar, logvar_range[0], logvar_range[1]))
    k = c // group
    mu_r = mu.reshape(b, l, group, k).permute(0, 1, 3, 2).reshape(-1, group)
    std_r = std.reshape(b, l, group, k).permute(0, 1, 3, 2).reshape(-1, group)
    ind, zq = argmax_rows(mu_r, std_r, cb, beta)
    zhat = zq.reshape(b, l, k, group).permute(0, 1, 3, 2).reshape(b, l, c).float()
    zhat = zhat.transpose(1, 2).reshape(b, c, h, w)
    indices = ind.reshape(b, l, k).transpose(1, 2).reshape(b, k, h, w)
    return zhat, indices


@torch.no_grad()
def gq2_forward(z: torch.Tensor, cb: torch.Tensor, dim: int, beta: float = 1.0, logvar_range=(-30.0, 20.0)):
    """GaussianQuantRegularizer2.quant_vq with dim_idx 1 (gaussian.py:273-331): contiguous channel grouping.
    z [B, 2c, h, w] -> (zhat [B, c, h, w], indices [B, K, h, w])."""
    z = torch.movedim(z.float(), 1, -1)
    zs = z.shape
    zf = z.reshape(-1, zs[-1])
    knum = zs[-1] // (dim * 2)
    mu, logvar = zf.chunk(2, -1)
    std = torch.exp(0.5 * torch.clamp(logvar, logvar_range[0], logvar_range[1]))
    ind, zq = argmax_rows(mu.reshape(-1, dim), std.reshape(-1, dim), cb, beta)
    zhat = zq.reshape(-1, knum * dim).float().reshape(*zs[:-1], -1)
    indices = ind.reshape(-1, knum).reshape(*zs[:-1], -1)
    return torch.movedim(zhat, -1, 1).contiguous(), torch.movedim(indices, -1, 1).contiguous()


@torch.no_grad()
def vq_forward(z: torch.Tensor, emb: torch.Tensor, codebook_num: int = 1):
    """VQQuantizer.forward, format bchw, the values (pit/quantization/vq.py:39-96): distance matrix by einsum (the BLAS
    accumulation order is whatever this host's torch picks -- the reference's own property), argmin, embedding lookup, the
    straight-through value z + (z_q - z).  Returns (z_q [B, c, h, w], indices [B, K, h, w])."""
    z = z.float().permute(0, 2, 3, 1).contiguous()
    dim = emb.shape[1]
    zf = z.view(-1, dim, codebook_num)
    zq, inds = [], []
    for i in range(codebook_num):
        d = (torch.sum(zf[:, :, i] ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1)
             - 2 * torch.einsum("bd,dn->bn", zf[:, :, i], emb.t()))
        ind = torch.argmin(d, dim=1)
        zq.append(torch.nn.functional.embedding(ind, emb)[:, :, None])
        inds.append(ind[:, None])
    zq = torch.cat(zq, dim=2).view(z.shape)
    indices = torch.cat(inds, dim=1).reshape(z.shape[0], z.shape[1], z.shape[2], codebook_num)
    zq = z + (zq - z)
    return zq.permute(0, 3, 1, 2).contiguous(), indices.permute(0, 3, 1, 2).contiguous()


@torch.no_grad()
def lfq_forward(x: torch.Tensor):
    """LFQQuantizer.forward in eval, format bchw, the values (pit/quantization/lfq.py:127-158, 196-208): sign quantisation,
    big-endian Horner pack over all channels, straight-through value x + (q - x).  Returns (quantized [B, c, h, w], indices [B, 1, h, w])."""
    x = x.float()
    b, c, h, w = x.shape
    xf = x.reshape(b, c, h * w).transpose(1, 2)
    q = torch.where(xf > 0, torch.ones_like(xf), -torch.ones_like(xf))
    idx = torch.zeros(b, h * w, dtype=torch.long)
    for i in range(c):
        idx = idx * 2 + (xf[..., i] > 0).long()
    quantized = xf + (q - xf)
    return quantized.transpose(1, 2).reshape(b, c, h, w).contiguous(), idx.reshape(b, 1, h, w)
