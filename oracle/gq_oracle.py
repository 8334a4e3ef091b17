"""CPU oracle for the encode->quantize->decode hot path (TEST INFRASTRUCTURE).

This module restates, on the CPU, what the reference computes on its
``backend="torch"`` path.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``pit_hip`` + ``libgqhip.so``) never does.

Reference lines restated (paths relative to /root/reference):
  prior_samples               pit/quantization/gaussian.py:15-19
  normal_log_prob             pit/quantization/gaussian.py:51-52
  GaussianQuantRegularizer    pit/quantization/gaussian.py:61-81,120-178 (eval)
  GaussianQuantRegularizer2   pit/quantization/gaussian.py:273-362 (quant_vq, dequant)
  VQQuantizer                 pit/quantization/vq.py:39-129
  LFQQuantizer                pit/quantization/lfq.py:127-228
  BSQQuantizer                pit/quantization/bsq.py:64-156
  FSQQuantizer                pit/quantization/fsq.py:29-89

The inner (row x code) arithmetic lives in ``gq_oracle.c``; this file holds the
layout glue (numpy) and the two transcendental calls the reference makes
through torch (``exp``/``log``), which are made through torch-CPU here as well
because their last-bit behaviour is libm specific (third
party arithmetic).

Parity pin: ``tests/golden/make_golden.py`` (run in the build container, where
/root/reference is importable) checks every function below bit-for-bit against
the imported reference and writes the vectors ``tests/test_oracle.py`` re-checks.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgq_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force: bool = False) -> str:
    """Compile gq_oracle.c with gcc (make).  Returns the .so path."""
    src = os.path.join(_HERE, "gq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libgq_oracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.gq_oracle_nlp.argtypes = [_f32p, _f32p, ctypes.c_int64, ctypes.c_int64]
        L.gq_oracle_scores.argtypes = [_f32p] * 6 + [ctypes.c_int64] * 3 + [ctypes.c_float]
        L.gq_oracle_argmax.argtypes = (
            [_f32p] * 5 + [_i64p, _f32p, _f32p, _f32p] + [ctypes.c_int64] * 3 + [ctypes.c_float, ctypes.c_int]
        )
        L.gq_oracle_cuda_scores.argtypes = [_f32p] * 4 + [ctypes.c_int64] * 3 + [ctypes.c_double]
        L.vq_oracle_argmin.argtypes = [_f32p, _f32p, _i64p, _f64p, _f64p] + [ctypes.c_int64] * 3 + [ctypes.c_int]
        L.lfq_oracle_pack.argtypes = [_f32p, _i64p, ctypes.c_int64, ctypes.c_int64]
        for fn in (L.gq_oracle_nlp, L.gq_oracle_scores, L.gq_oracle_argmax, L.gq_oracle_cuda_scores,
                   L.vq_oracle_argmin, L.lfq_oracle_pack):
            fn.restype = None
        _lib = L
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- a1
def prior_samples(n_samples: int, n_variable: int, seed_rec: int):
    """gaussian.py:15-19 -- scrambled Sobol points pushed through norm.ppf (fp64)."""
    import torch
    from scipy.stats import norm
    from torch.quasirandom import SobolEngine

    sobol = SobolEngine(n_variable, scramble=True, seed=seed_rec)
    return torch.from_numpy(norm.ppf(sobol.draw(n_samples)))


def codebook(n_samples: int, dim: int, seed: int = 42) -> np.ndarray:
    """The fp32 buffer the reference registers (gaussian.py:50)."""
    return prior_samples(n_samples, dim, seed).float().numpy()


# --------------------------------------------------------------------------- a2
def nlp_table(cb: np.ndarray) -> np.ndarray:
    cb = _f32(cb)
    out = np.empty_like(cb)
    lib().gq_oracle_nlp(_p(cb, _f32p), _p(out, _f32p), cb.shape[0], cb.shape[1])
    return out


def torch_exp_half(logvar: np.ndarray) -> np.ndarray:
    """std = torch.exp(0.5 * logvar) on torch-CPU (gaussian.py:79)."""
    import torch

    return torch.exp(0.5 * torch.from_numpy(_f32(logvar))).numpy()


def torch_log(std: np.ndarray) -> np.ndarray:
    """Normal.log_prob's ``self.scale.log()`` on torch-CPU."""
    import torch

    return torch.from_numpy(_f32(std)).log().numpy()


# --------------------------------------------------------------------------- a4
def score_matrix(mu, std, cb, beta: float = 1.0, logstd=None, nlp=None) -> np.ndarray:
    mu, std, cb = _f32(mu), _f32(std), _f32(cb)
    logstd = torch_log(std) if logstd is None else _f32(logstd)
    nlp = nlp_table(cb) if nlp is None else _f32(nlp)
    rows, dim = mu.shape
    n = cb.shape[0]
    out = np.empty((rows, n), dtype=np.float32)
    lib().gq_oracle_scores(_p(mu, _f32p), _p(std, _f32p), _p(logstd, _f32p), _p(cb, _f32p), _p(nlp, _f32p),
                           _p(out, _f32p), dim, rows, n, float(beta))
    return out


def argmax_rows(mu, std, cb, beta: float = 1.0, logstd=None, nlp=None, threads: int = 0,
                with_gap: bool = False):
    """(mu, std) -> (indices int64 [rows], zhat fp32 [rows, dim]) in the reference's op order.

    ``with_gap=True`` also returns (best, runner-up) scores per row."""
    mu, std, cb = _f32(mu), _f32(std), _f32(cb)
    logstd = torch_log(std) if logstd is None else _f32(logstd)
    nlp = nlp_table(cb) if nlp is None else _f32(nlp)
    rows, dim = mu.shape
    assert dim <= 64 and std.shape == mu.shape and cb.shape[1] == dim
    n = cb.shape[0]
    idx = np.empty(rows, dtype=np.int64)
    zhat = np.empty((rows, dim), dtype=np.float32)
    best = np.empty(rows, dtype=np.float32)
    second = np.empty(rows, dtype=np.float32)
    lib().gq_oracle_argmax(_p(mu, _f32p), _p(std, _f32p), _p(logstd, _f32p), _p(cb, _f32p), _p(nlp, _f32p),
                           _p(idx, _i64p), _p(zhat, _f32p), _p(best, _f32p), _p(second, _f32p),
                           dim, rows, n, float(beta), int(threads))
    if with_gap:
        return idx, zhat, best, second
    return idx, zhat


def cuda_formula_scores(mu, std, cb, beta: float = 1.0) -> np.ndarray:
    """gq_cuda.cu:31-38 score matrix (compat op), small cases."""
    mu, std, cb = _f32(mu), _f32(std), _f32(cb)
    rows, dim = mu.shape
    n = cb.shape[0]
    out = np.empty((rows, n), dtype=np.float32)
    lib().gq_oracle_cuda_scores(_p(mu, _f32p), _p(std, _f32p), _p(cb, _f32p), _p(out, _f32p), dim, rows, n,
                                float(beta))
    return out


# --------------------------------------------------------------------------- a3 / a6
def _split_mu_std(zflat: np.ndarray, logvar_range) -> Tuple[np.ndarray, np.ndarray]:
    c = zflat.shape[-1] // 2
    mu = zflat[..., :c]
    logvar = np.clip(zflat[..., c:], np.float32(logvar_range[0]), np.float32(logvar_range[1]))
    return _f32(mu), torch_exp_half(logvar)


def gq1_forward(z: np.ndarray, cb: np.ndarray, group: int, beta: float = 1.0, fmt: str = "bchw",
                logvar_range=(-30.0, 20.0), threads: int = 0):
    """GaussianQuantRegularizer.forward, eval branch (gaussian.py:61-81,120-160).

    Returns (zhat, indices); ``zhat_noquant`` is RNG dependent and not restated."""
    z = _f32(z)
    if fmt == "bchw":
        b, c2, h, w = z.shape
        zf = z.reshape(b, c2, h * w).transpose(0, 2, 1)  # b (h w) c
    else:
        b, _, c2 = z.shape
        zf = z
    l = zf.shape[1]
    c = c2 // 2
    k = c // group
    mu, std = _split_mu_std(zf, logvar_range)
    # gaussian.py:122-123: row = (b*L + l)*K + k ; column g <- channel g*K + k
    mu_r = mu.reshape(b, l, group, k).transpose(0, 1, 3, 2).reshape(-1, group)
    std_r = std.reshape(b, l, group, k).transpose(0, 1, 3, 2).reshape(-1, group)
    idx, zq = argmax_rows(mu_r, std_r, cb, beta, threads=threads)
    zhat = zq.reshape(b, l, k, group).transpose(0, 1, 3, 2).reshape(b, l, c)
    indices = idx.reshape(b, l, k)
    if fmt == "bchw":
        zhat = zhat.transpose(0, 2, 1).reshape(b, c, h, w)
        indices = indices.transpose(0, 2, 1).reshape(b, k, h, w)
    return np.ascontiguousarray(zhat), np.ascontiguousarray(indices)


def gq1_dequant(indices: np.ndarray, cb: np.ndarray, group: int, fmt: str = "bchw") -> np.ndarray:
    """GaussianQuantRegularizer.dequant (gaussian.py:162-178)."""
    cb = _f32(cb)
    if fmt == "bchw":
        b, ng, h, w = indices.shape
        ind = indices.reshape(b, ng, h * w).transpose(0, 2, 1)
    else:
        b, _, ng = indices.shape
        ind = indices
    l = ind.shape[1]
    zq = cb[ind.reshape(-1)]
    zhat = zq.reshape(b, l, ng, group).transpose(0, 1, 3, 2).reshape(b, l, ng * group)
    if fmt == "bchw":
        zhat = zhat.transpose(0, 2, 1).reshape(b, ng * group, h, w)
    return np.ascontiguousarray(zhat)


# --------------------------------------------------------------------------- a7
def gq2_quant_vq(z: np.ndarray, cb: np.ndarray, dim: int, dim_idx: int = 1, beta: float = 1.0,
                 logvar_range=(-30.0, 20.0), threads: int = 0):
    """GaussianQuantRegularizer2.quant_vq (gaussian.py:273-331): contiguous grouping."""
    z = np.moveaxis(_f32(z), dim_idx, -1)
    zs = z.shape
    zf = z.reshape(-1, zs[-1])
    knum = zs[-1] // (2 * dim)
    mu, std = _split_mu_std(zf, logvar_range)
    idx, zq = argmax_rows(mu.reshape(-1, dim), std.reshape(-1, dim), cb, beta, threads=threads)
    zhat = zq.reshape(-1, knum * dim).reshape(*zs[:-1], -1)
    indices = idx.reshape(-1, knum).reshape(*zs[:-1], -1)
    return (np.ascontiguousarray(np.moveaxis(zhat, -1, dim_idx)),
            np.ascontiguousarray(np.moveaxis(indices, -1, dim_idx)))


def gq2_quant_gaussian_stats(z: np.ndarray, dim: int, n_samples: int, lam_state, dim_idx: int = 1, tolerance: float = 0.5,
                             lam_factor: float = 1.01, lam_range=(1e-7, 1e7), logvar_range=(-30.0, 20.0)):
    """The deterministic part of GaussianQuantRegularizer2.quant_gaussian (gaussian.py:211-257): KL bits per group, their
    mean / min / max, the re-weighted loss with the lambdas as they are on entry, and the lambda update -- incl. the decrease of
    lam_max that the reference spells as an expression without effect (gaussian.py:251).  lam_state = (lam, lam_min, lam_max)
    Python floats; returns ({"kl_loss", "bits-mean", "bits-min", "bits-max"} as float32 scalars, new lam_state).  fp32 element
    arithmetic in the reference's op order; the reductions are numpy's (the reference's are torch's: equal to rounding)."""
    import math

    lam, lam_min, lam_max = (float(v) for v in lam_state)
    log2n = int(math.log(n_samples, 2))
    zf = np.moveaxis(_f32(z), dim_idx, -1)
    zf = np.ascontiguousarray(zf).reshape(-1, zf.shape[-1])
    knum = zf.shape[-1] // (2 * dim)
    mu, logvar = np.split(zf, 2, axis=-1)
    logvar = np.clip(logvar, np.float32(logvar_range[0]), np.float32(logvar_range[1]))
    var = np.exp(logvar.astype(np.float64)).astype(np.float32)
    kl2 = np.float32(1.4426 * 0.5) * (((mu * mu + var) - np.float32(1.0)) - logvar)
    kl2 = kl2.reshape(-1, knum, dim).astype(np.float64).sum(-1).astype(np.float32)
    mean, kmin, kmax = np.float32(kl2.astype(np.float64).mean()), kl2.min(), kl2.max()
    hi, lo = np.float32(log2n + tolerance), np.float32(log2n - tolerance)
    ge = (kl2 > hi).astype(np.float32) * np.float32(lam_max)
    eq = (kl2 <= hi).astype(np.float32) * (kl2 >= lo).astype(np.float32)
    le = (kl2 < lo).astype(np.float32) * np.float32(lam_min)
    w = ge * kl2 + eq * kl2 + le * kl2
    kl_loss = np.float32(w.astype(np.float64).mean()) * np.float32(lam)
    lam = lam * lam_factor if mean > log2n else lam / lam_factor
    if kmax > hi:
        lam_max = lam_max * lam_factor
    lam_max = max(min(lam_max, lam_range[1]), 1.0)
    lam_min = lam_min / lam_factor if kmin < lo else lam_min * lam_factor
    lam_min = max(min(lam_min, 1.0), lam_range[0])
    return ({"kl_loss": np.float32(kl_loss), "bits-mean": mean, "bits-min": np.float32(kmin), "bits-max": np.float32(kmax)},
            (lam, lam_min, lam_max))


def gq2_dequant(indices: np.ndarray, cb: np.ndarray, dim: int, dim_idx: int = 1) -> np.ndarray:
    """GaussianQuantRegularizer2.dequant (gaussian.py:347-362)."""
    ind = np.moveaxis(indices, dim_idx, -1)
    ish = ind.shape
    zq = _f32(cb)[ind.reshape(-1)].reshape(-1, ish[-1] * dim).reshape(*ish[:-1], -1)
    return np.ascontiguousarray(np.moveaxis(zq, -1, dim_idx))


# --------------------------------------------------------------------------- a8
def vq_argmin_rows(z, emb, threads: int = 0, with_gap: bool = False):
    z, emb = _f32(z), _f32(emb)
    rows, dim = z.shape
    idx = np.empty(rows, dtype=np.int64)
    best = np.empty(rows, dtype=np.float64)
    second = np.empty(rows, dtype=np.float64)
    lib().vq_oracle_argmin(_p(z, _f32p), _p(emb, _f32p), _p(idx, _i64p), _p(best, _f64p), _p(second, _f64p),
                           dim, rows, emb.shape[0], int(threads))
    return (idx, best, second) if with_gap else idx


def vq_forward(z: np.ndarray, emb: np.ndarray, codebook_num: int = 1, fmt: str = "bchw", threads: int = 0,
               with_gap: bool = False):
    """VQQuantizer.forward values (vq.py:39-100): z.view(-1, dim, K) -> channel = d*K + k."""
    z = _f32(z)
    emb = _f32(emb)
    dim = emb.shape[1]
    if fmt == "bchw":
        b, c, h, w = z.shape
        zl = z.transpose(0, 2, 3, 1)  # b h w c
    else:
        b, l, c = z.shape
        h = w = int(np.sqrt(l))
        zl = z.reshape(b, h, w, c)
    zf = np.ascontiguousarray(zl).reshape(-1, dim, codebook_num)
    zq = np.empty_like(zf)
    inds, gaps = [], []
    for k in range(codebook_num):
        idx, best, second = vq_argmin_rows(zf[:, :, k], emb, threads, with_gap=True)
        zq[:, :, k] = emb[idx]
        inds.append(idx[:, None])
        gaps.append((second - best)[:, None])
    zq = zq.reshape(b, h, w, c)
    indices = np.concatenate(inds, 1).reshape(b, h, w, codebook_num)
    gap = np.concatenate(gaps, 1).reshape(b, h, w, codebook_num)
    if fmt == "bchw":
        zq, indices, gap = zq.transpose(0, 3, 1, 2), indices.transpose(0, 3, 1, 2), gap.transpose(0, 3, 1, 2)
    else:
        zq, indices, gap = zq.reshape(b, h * w, c), indices.reshape(b, h * w, -1), gap.reshape(b, h * w, -1)
    out = (np.ascontiguousarray(zq), np.ascontiguousarray(indices))
    return out + (np.ascontiguousarray(gap),) if with_gap else out


def vq_forward_eval(z: np.ndarray, emb: np.ndarray, codebook_num: int = 1, fmt: str = "bchw", beta: float = 0.25,
                    legacy: bool = True, threads: int = 0):
    """What VQQuantizer.forward returns (vq.py:76-98): the straight-through VALUE z + (z_q - z) in fp32, the indices, the
    codebook loss (fp32 squares, mean, beta on the reference's side of the sum) and the oracle's top-2 gaps."""
    zq, ind, gap = vq_forward(z, emb, codebook_num, fmt, threads, with_gap=True)
    z = _f32(z)
    d = zq - z
    m = np.float32((d * d).astype(np.float64).mean())
    loss = m + np.float32(beta) * m if legacy else np.float32(beta) * m + m
    return np.ascontiguousarray(z + d), ind, np.float32(loss), gap


def vq_dequant(indices: np.ndarray, emb: np.ndarray, codebook_num: int = 1, fmt: str = "bchw") -> np.ndarray:
    """VQQuantizer.dequant (vq.py:102-129)."""
    emb = _f32(emb)
    dim = emb.shape[1]
    if fmt == "bchw":
        b, _, h, w = indices.shape
        ind = indices.transpose(0, 2, 3, 1)
    else:
        b, l, _ = indices.shape
        h = w = int(np.sqrt(l))
        ind = indices.reshape(b, h, w, -1)
    ind = np.ascontiguousarray(ind).reshape(-1, codebook_num)
    zq = np.stack([emb[ind[:, k]] for k in range(codebook_num)], axis=2)  # rows, dim, K
    zq = zq.reshape(b, h, w, dim * codebook_num)
    if fmt == "bchw":
        return np.ascontiguousarray(zq.transpose(0, 3, 1, 2))
    return np.ascontiguousarray(zq.reshape(b, h * w, -1))


# --------------------------------------------------------------------------- a9
def lfq_forward(x: np.ndarray, fmt: str = "bchw"):
    """LFQQuantizer.forward eval values (lfq.py:127-158, 196-208): sign + Horner bit-pack."""
    x = _f32(x)
    if fmt == "bchw":
        b, c, h, w = x.shape
        xf = x.reshape(b, c, h * w).transpose(0, 2, 1)
    else:
        b, _, c = x.shape
        xf = x
    l = xf.shape[1]
    flat = np.ascontiguousarray(xf).reshape(-1, c)
    idx = np.empty(flat.shape[0], dtype=np.int64)
    lib().lfq_oracle_pack(_p(flat, _f32p), _p(idx, _i64p), flat.shape[0], c)
    q = np.where(flat > 0, np.float32(1.0), np.float32(-1.0)).reshape(b, l, c)
    indices = idx.reshape(b, l, 1)
    if fmt == "bchw":
        q = q.transpose(0, 2, 1).reshape(b, c, h, w)
        indices = indices.transpose(0, 2, 1).reshape(b, 1, h, w)
    return np.ascontiguousarray(q), np.ascontiguousarray(indices)


def lfq_dequant(indices: np.ndarray, nbits: int = 16, fmt: str = "bchw") -> np.ndarray:
    """LFQQuantizer.dequant (lfq.py:210-228); the reference hard-codes 16 bits (``15 - i``)."""
    assert nbits == 16
    if fmt == "bchw":
        b, ng, h, w = indices.shape
        ind = indices.reshape(b, ng, h * w).transpose(0, 2, 1)
    else:
        b, _, ng = indices.shape
        ind = indices
    l = ind.shape[1]
    q = np.zeros((b, l, ng, nbits), dtype=np.float32)
    rem = ind.astype(np.int64).copy()
    for i in range(nbits):
        q[:, :, :, 15 - i] = (rem % 2).astype(np.float32)
        rem = rem // 2
    q = q * 2.0 - 1.0
    if fmt == "bchw":
        # "b (h w) c n -> b (c n) h w"
        q = q.reshape(b, h, w, ng, nbits).transpose(0, 3, 4, 1, 2).reshape(b, ng * nbits, h, w)
    return np.ascontiguousarray(q)


# --------------------------------------------------------------------------- f3: BSQ / FSQ
def bsq_forward(x: np.ndarray, num_codebooks: int = 16, codebook_dim: int = 1, fmt: str = "bchw"):
    """BSQQuantizer.forward eval values (bsq.py:64-131): F.normalize -> sign -> 16-bit pack over the
    codebook axis, output scaled by 1/sqrt(embed_dim).  torch-CPU ops for the normalisation."""
    import torch
    import torch.nn.functional as F

    xt = torch.from_numpy(_f32(x))
    if fmt == "bchw":
        b, c, h, w = xt.shape
        xt = xt.reshape(b, c, h * w).permute(0, 2, 1)
    else:
        b, _, c = xt.shape
    l = xt.shape[1]
    xn = F.normalize(xt, dim=-1).reshape(b, l, num_codebooks, codebook_dim)
    q = torch.where(xn > 0, torch.tensor(1.0), torch.tensor(-1.0))
    bits = ((q + 1.0) / 2.0).to(torch.long)
    idx = torch.zeros_like(bits[:, :, 0, :])
    for i in range(16):
        idx = idx * 2 + bits[:, :, i, :]
    out = (xn + (q - xn)) * (1.0 / (num_codebooks * codebook_dim) ** 0.5)
    out = out.reshape(b, l, c)
    if fmt == "bchw":
        out = out.permute(0, 2, 1).reshape(b, c, h, w)
        idx = idx.permute(0, 2, 1).reshape(b, codebook_dim, h, w)
    return np.ascontiguousarray(out.numpy()), np.ascontiguousarray(idx.numpy())


def bsq_dequant(indices: np.ndarray, embed_dim: int = 16, fmt: str = "bchw") -> np.ndarray:
    """BSQQuantizer.dequant (bsq.py:133-156)."""
    return lfq_dequant(indices, 16, fmt) * np.float32(1.0 / embed_dim ** 0.5)


def fsq_forward(z: np.ndarray, levels, fmt: str = "bchw", with_margin: bool = False):
    """FSQQuantizer.forward (fsq.py:29-68) with torch-CPU tanh/atanh.  Returns (zhat, packed int32
    indices[, distance of the bounded value to the nearest rounding boundary])."""
    import torch

    zt = torch.from_numpy(_f32(z))
    if fmt == "bchw":
        b, c, h, w = zt.shape
        zt = zt.reshape(b, c, h * w).permute(0, 2, 1)
    else:
        b, _, c = zt.shape
    lev = torch.tensor(list(levels), dtype=torch.int32)
    half_l = (lev - 1) * (1 + 1e-3) / 2
    offset = torch.where(lev % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    bounded = (zt + shift).tanh() * half_l - offset
    half_width = lev // 2
    r = bounded.round()
    zhat = r / half_width
    ind = (r + half_width).to(torch.int32)
    packed = torch.zeros_like(ind[:, :, 0:1])
    for i in range(len(levels)):
        packed = packed * lev[i] + ind[:, :, i:i + 1]
    margin = (0.5 - (bounded - bounded.floor() - 0.5).abs()).amin(dim=-1, keepdim=True)
    if fmt == "bchw":
        zhat = zhat.permute(0, 2, 1).reshape(b, c, h, w)
        packed = packed.permute(0, 2, 1).reshape(b, 1, h, w)
        margin = margin.permute(0, 2, 1).reshape(b, 1, h, w)
    out = (np.ascontiguousarray(zhat.numpy()), np.ascontiguousarray(packed.numpy()))
    return out + (np.ascontiguousarray(margin.numpy()),) if with_margin else out


def fsq_dequant(indices: np.ndarray, levels, fmt: str = "bchw") -> np.ndarray:
    """FSQQuantizer.dequant (fsq.py:70-89)."""
    ind = np.asarray(indices).astype(np.int64)
    if fmt == "bchw":
        b, _, h, w = ind.shape
        ind = ind.reshape(b, 1, h * w).transpose(0, 2, 1)
    digits = []
    for lv in reversed(list(levels)):
        digits.append(ind % lv)
        ind = ind // lv
    d = np.concatenate(digits[::-1], axis=2)
    hw = (np.asarray(list(levels)) // 2).astype(np.int64)
    zhat = ((d - hw).astype(np.float32) / hw.astype(np.float32)).astype(np.float32)
    if fmt == "bchw":
        zhat = zhat.transpose(0, 2, 1).reshape(b, len(levels), h, w)
    return np.ascontiguousarray(zhat)


# --------------------------------------------------------------------------- a12 (index logic only)
def distributed_sampler_indices(n: int, world: int, rank: int):
    """torch DistributedSampler(shuffle=False, drop_last=False): pad by wrapping, stride by world."""
    total = -(-n // world) * world
    idx = list(range(n))
    pad = total - n
    if pad:
        idx += (idx * (-(-pad // max(n, 1))))[:pad]
    return idx[rank:total:world]


def eval_batches(n: int, world: int, rank: int, bs: int):
    """DataLoader(batch_size=bs, drop_last=True) over the rank's sampler (eval.py:97-107)."""
    ids = distributed_sampler_indices(n, world, rank)
    return [ids[i:i + bs] for i in range(0, len(ids) - bs + 1, bs)]


def reinterleave(per_rank):
    """eval.py:213-214: out[j] = per_rank[j % W][j // W]."""
    w = len(per_rank)
    total = sum(len(p) for p in per_rank)
    return [per_rank[j % w][j // w] for j in range(total)]
