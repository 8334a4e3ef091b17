"""Gaussian-quantisation regularizers, MI355X path.

Drop-in mirrors of ``pit.quantization.gaussian`` (reference file
pit/quantization/gaussian.py): same constructor keywords, same ``forward(z) ->
(zhat, info)`` / ``dequant(indices) -> zhat`` contracts, same buffer names
(``prior_samples``, ``normal_log_prob``, both non-persistent).  The eval branch
(gaussian.py:120-160 and :273-331) runs as ONE call into libgqhip.so
(``gq_quantize_z_f32``): chunk / clamp / exp, the group permutes, the
rows x 65 536 score matrix, the arg-max and the codeword gather are all inside
the HIP kernels and the score matrix never exists in HBM.  For
``GaussianQuantRegularizer2`` the whole eval forward (gaussian.py:333-345: the
Gaussian branch's sample, KL statistics, re-weighted loss and lambda state
machine, quant_vq, the straight-through mix) is ONE call as well
(``gq_quantize_z_gauss_f32``) whenever autograd is not recording; the lambda
state then advances on the device and is read back only when somebody looks.

backend:
  "hip"   fused path (default here).
  "cuda"  what every shipped GQ YAML says (configs/sd3unet_gq_0.25.yaml:33): in
          the reference that selects its fast path, and it selects the fast path
          here too -- the fused kernels, indices bit-identical to the reference's
          torch backend, no score matrix (``self.perturbed`` stays ``None``).
  "cuda-compat" (or backend "cuda" with GQHIP_COMPAT=1 in the environment)
          the reference's own call sequence -- ``gq_cuda.ops.gq_cuda`` into a
          persistent ``perturbed`` buffer, then ``torch.argmax`` and
          ``index_select`` (gaussian.py:124-133) -- served by our HIP build of
          the ``extension_cpp::gq`` op: for users who want the rows x n score
          matrix.  HBM-bound by construction (4.29 GB written and re-read at
          bs 16), and its arg-max is the op's own rounding of 2 s + const(r):
          equal to the bit-exact indices except at rounding-level ties.
  "torch" accepted for config compatibility; runs the fused HIP path (the
          indices are those of the reference's torch backend, bit for bit).
There is no CPU implementation in this package: tensors must live on a HIP
device (``pit_hip._lib`` raises otherwise).
"""
from __future__ import annotations

import math
import os
from typing import Sequence

import torch
import torch.nn as nn
from torch.distributions import Normal

from .. import _lib


def prior_samples(n_samples: int, n_variable: int, seed_rec: int) -> torch.Tensor:
    """Scrambled-Sobol points mapped through the normal quantile (gaussian.py:15-19).

    Third-party arithmetic (torch SobolEngine + scipy ``norm.ppf``), called the
    same way as the reference so the codebook is identical (sha256 pinned in
    tests/golden)."""
    from scipy.stats import norm
    from torch.quasirandom import SobolEngine

    sobol = SobolEngine(n_variable, scramble=True, seed=seed_rec)
    return torch.from_numpy(norm.ppf(sobol.draw(n_samples)))


def _kl_bits(mu: torch.Tensor, var: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    return 1.4426 * 0.5 * (torch.pow(mu, 2) + var - 1.0 - logvar)


class _GaussianQuantBase(nn.Module):
    def _setup(self, n_samples: int, dim: int, seed: int, beta: float, backend: str,
               logvar_range: Sequence[float], tolerance: float, lam_factor: float, lam_range) -> None:
        self.n_samples = n_samples
        self.log_n_samples = int(math.log(n_samples, 2))
        self.logvar_range = logvar_range
        self.lam_factor, self.tolerance = lam_factor, tolerance
        # lam / lam_min / lam_max (gaussian.py:41-43): Python floats in the reference.  Here properties over a host copy AND a
        # float64 [3] device copy: the fused eval forward of GaussianQuantRegularizer2 advances the device copy inside its launches
        # (no host read per forward); reading an attribute pulls the device copy once, writing one makes the host copy authoritative.
        self._lam_host = [1.0, 1.0, 1.0]
        self._lam_dev = None          # float64 [3] on the module's device, or None
        self._lam_dev_newer = False   # the device copy has advanced past the host copy
        self.lam_range = lam_range
        self.beta, self.seed = beta, seed
        self.register_buffer("prior_samples", prior_samples(n_samples, dim, seed).float(), persistent=False)
        self.normal_dist = Normal(torch.zeros([1, dim]), torch.ones([1, dim]))
        self.register_buffer("normal_log_prob", self.normal_dist.log_prob(self.prior_samples).float(),
                             persistent=False)
        self.perturbed = None
        if backend not in ("hip", "cuda", "cuda-compat", "torch"):
            raise ValueError(f"unknown backend {backend!r}")
        self.backend = backend
        self._ws = _lib.Workspace()   # scratch + the codebook cache (validated by the library against the codebook's content hash on every call)

    # ---- the adaptive lambda state --------------------------------------------------------------------------------------
    def _lam_pull(self) -> None:
        if self._lam_dev_newer:
            self._lam_host = [float(v) for v in self._lam_dev.cpu().tolist()]     # the one host read, only when somebody looks
            self._lam_dev_newer = False

    def _lam_get(self, i: int) -> float:
        self._lam_pull()
        return self._lam_host[i]

    def _lam_set(self, i: int, v) -> None:
        self._lam_pull()
        self._lam_host[i] = float(v)
        self._lam_dev = None          # re-uploaded by the next fused forward

    lam = property(lambda self: self._lam_get(0), lambda self, v: self._lam_set(0, v))
    lam_min = property(lambda self: self._lam_get(1), lambda self, v: self._lam_set(1, v))
    lam_max = property(lambda self: self._lam_get(2), lambda self, v: self._lam_set(2, v))

    def _lam_state_on(self, device) -> torch.Tensor:
        """The device copy of (lam, lam_min, lam_max), current; the caller's launch advances it in place."""
        if self._lam_dev is None or self._lam_dev.device != device:
            self._lam_pull()
            self._lam_dev = torch.tensor(self._lam_host, dtype=torch.float64, device=device)
        self._lam_dev_newer = True
        return self._lam_dev

    def _compat(self) -> bool:
        """True when the caller asked for the score-matrix call sequence instead of the fused kernels."""
        return self.backend == "cuda-compat" or (self.backend == "cuda" and os.environ.get("GQHIP_COMPAT", "0") == "1")

    # the reference's "cuda" call sequence on rows (gaussian.py:124-133 / :289-298)
    def _compat_rows(self, mu: torch.Tensor, std: torch.Tensor, dim: int):
        import gq_cuda

        if self.perturbed is None or self.perturbed.shape[0] != mu.shape[0] or self.perturbed.device != mu.device:
            self.perturbed = torch.zeros([mu.shape[0], self.n_samples], device=mu.device).contiguous()
        gq_cuda.ops.gq_cuda(mu, std, self.prior_samples, self.perturbed, dim, mu.shape[0], self.n_samples, self.beta)
        indices = torch.argmax(self.perturbed, dim=1).clone()
        return torch.index_select(self.prior_samples, 0, indices), indices

    def _update_lambdas(self, kl2_mean, kl2_min, kl2_max, buggy_lam_max: bool) -> None:
        # gaussian.py:103-117 (GQ1) / :238-252 (GQ2, whose lam_max decrease is a no-op: line 251)
        n, tol, f = self.log_n_samples, self.tolerance, self.lam_factor
        self.lam = self.lam * f if kl2_mean > n else self.lam / f
        if kl2_max > n + tol:
            self.lam_max = self.lam_max * f
        elif not buggy_lam_max:
            self.lam_max = self.lam_max / f
        self.lam_max = max(min(self.lam_max, self.lam_range[1]), 1.0)
        self.lam_min = self.lam_min / f if kl2_min < n - tol else self.lam_min * f
        self.lam_min = max(min(self.lam_min, 1.0), self.lam_range[0])

    def _weighted_kl(self, kl2: torch.Tensor) -> torch.Tensor:
        n, tol = self.log_n_samples, self.tolerance
        ge = (kl2 > n + tol).type(kl2.dtype) * self.lam_max
        eq = (kl2 <= n + tol).type(kl2.dtype) * (kl2 >= n - tol).type(kl2.dtype)
        le = (kl2 < n - tol).type(kl2.dtype) * self.lam_min
        return ge * kl2 + eq * kl2 + le * kl2


class GaussianQuantRegularizer(_GaussianQuantBase):
    """train(): Gaussian VAE sample + KL-to-log2(N) loss; eval(): VQ with the fixed
    quasi-random codebook (reference gaussian.py:22-178)."""

    def __init__(self, format, n_samples, group=1, logvar_range=[-30.0, 20.0], tolerance=0.5, lam_factor=1.01,
                 seed=42, beta=1.0, backend="hip"):
        super().__init__()
        assert format in ["bchw", "blc"]
        self.format, self.group = format, group
        self._setup(n_samples, group, seed, beta, backend, logvar_range, tolerance, lam_factor, (1e-3, 1e3))

    def _forward_fused(self, z):
        """Eval branch (gaussian.py:120-160) as ONE call: chunk / clamp / exp, the group permutes, zhat_noquant, the
        rows x n score matrix, arg-max and gather all happen inside libgqhip's three launches.  A channels_last z
        (what the NHWC conv stack hands over) is read, and zhat / indices / zhat_noquant are written, in that memory
        layout directly: NHWC memory of [B, C, h, w] IS the "blc" layout with L = h * w, so nothing is transposed."""
        z = z.float()
        if self.format == "bchw":
            b, c2, h, w = z.shape
            shape_n = (b, c2 // 2, h, w)
            if not z.is_contiguous() and z.is_contiguous(memory_format=torch.channels_last):
                zmem = z.permute(0, 2, 3, 1).reshape(b, h * w, c2)   # a view of the same memory
                # one draw of the size of mu (advances the generator like gaussian.py:121)
                noise = torch.randn((b, h * w, c2 // 2), dtype=torch.float32, device=z.device)
                ind, zhat, noq = _lib.gq_quantize_z(zmem, self.prior_samples, self.group, "blc", _lib.GQHIP_GROUP_STRIDED,
                                                    self.logvar_range, self.beta, self._ws, noise=noise)
                as_bchw = lambda t: t.view(b, h, w, -1).permute(0, 3, 1, 2)   # logical [B, C, h, w], NHWC memory
                return as_bchw(zhat), {"indices": as_bchw(ind), "zhat_noquant": as_bchw(noq)}
        else:
            b, l, c2 = z.shape
            shape_n = (b, l, c2 // 2)
        noise = torch.randn(shape_n, dtype=torch.float32, device=z.device)
        indices, zhat, zhat_noquant = _lib.gq_quantize_z(z, self.prior_samples, self.group, self.format,
                                                         _lib.GQHIP_GROUP_STRIDED, self.logvar_range, self.beta,
                                                         self._ws, noise=noise)
        return zhat, {"indices": indices, "zhat_noquant": zhat_noquant}

    def forward(self, z):
        if not self.training and not self._compat():
            return self._forward_fused(z)
        z = z.float()
        if self.format == "bchw":
            b, c2, h, w = z.shape
            l = h * w
            zf = z.reshape(b, c2, l).transpose(1, 2)  # b (h w) c, a view
        else:
            b, l, c2 = z.shape
            zf = z
        c = c2 // 2
        mu, logvar = zf.chunk(2, 2)
        logvar = torch.clamp(logvar, self.logvar_range[0], self.logvar_range[1])
        std = torch.exp(0.5 * logvar)

        if self.training:
            var = torch.exp(logvar)
            zhat = mu + torch.randn_like(mu) * std
            kl2 = _kl_bits(mu, var, logvar).reshape(b, l, self.group, c // self.group).sum(dim=2)
            kl2_mean, kl2_min, kl2_max = torch.mean(kl2), torch.min(kl2), torch.max(kl2)
            kl_loss = torch.sum(self._weighted_kl(kl2), dim=[1, 2])
            kl_loss = torch.sum(kl_loss) / kl_loss.shape[0] * self.lam
            self._update_lambdas(kl2_mean, kl2_min, kl2_max, buggy_lam_max=False)
            if self.format == "bchw":
                zhat = zhat.transpose(1, 2).reshape(b, c, h, w)
            info = {"kl_loss": torch.mean(kl_loss), "bits-mean": kl2_mean, "bits-min": kl2_min,
                    "bits-max": kl2_max, "lam": torch.zeros_like(kl_loss) + self.lam}
            return zhat, info

        # backend "cuda-compat": the reference's own call sequence (score matrix -> argmax -> index_select)
        zhat_noquant = mu + torch.randn_like(mu) * std  # consumes RNG in eval, like gaussian.py:121
        k = c // self.group
        mu_r = mu.reshape(b, l, self.group, k).permute(0, 1, 3, 2).reshape(-1, self.group)
        std_r = std.reshape(b, l, self.group, k).permute(0, 1, 3, 2).reshape(-1, self.group)
        zq, ind = self._compat_rows(mu_r.contiguous(), std_r.contiguous(), self.group)
        zhat = zq.reshape(b, l, k, self.group).permute(0, 1, 3, 2).reshape(b, l, c).float()
        indices = ind.reshape(b, l, k)
        if self.format == "bchw":
            zhat = zhat.transpose(1, 2).reshape(b, c, h, w)
            indices = indices.transpose(1, 2).reshape(b, k, h, w)
            zhat_noquant = zhat_noquant.transpose(1, 2).reshape(b, c, h, w)
        return zhat, {"indices": indices, "zhat_noquant": zhat_noquant}

    def dequant(self, indices):
        return _lib.gq_dequant(indices, self.prior_samples, self.group, self.format, _lib.GQHIP_GROUP_STRIDED)


class GaussianQuantRegularizer2(_GaussianQuantBase):
    """Contiguous-channel grouping, arbitrary channel axis, Gaussian + VQ branches mixed
    by a straight-through estimator (reference gaussian.py:181-362)."""

    def __init__(self, dim, codebook_size, dim_idx=1, logvar_range=[-30.0, 20.0], tolerance=0.5, lam_factor=1.01,
                 seed=42, beta=1.0, use_ste=True, backend="hip"):
        super().__init__()
        self.dim, self.dim_idx, self.use_ste = dim, dim_idx, use_ste
        self._setup(codebook_size, dim, seed, beta, backend, logvar_range, tolerance, lam_factor, (1e-7, 1e7))

    def _split(self, z):
        z = torch.movedim(z, self.dim_idx, -1)
        assert z.shape[-1] % (self.dim * 2) == 0
        z_shape = z.shape
        z = z.reshape(-1, z_shape[-1])
        mu, logvar = z.chunk(2, -1)
        logvar = torch.clamp(logvar, self.logvar_range[0], self.logvar_range[1])
        return z, z_shape, mu, logvar, torch.exp(0.5 * logvar)

    def _restore(self, t, z_shape):
        return torch.movedim(t.reshape(*z_shape[:-1], -1), -1, self.dim_idx)

    def quant_gaussian(self, z):
        z2, z_shape, mu, logvar, std = self._split(z)
        knum = z2.shape[-1] // (self.dim * 2)
        var = torch.exp(logvar)
        zhat = mu + torch.randn_like(mu) * std
        kl2 = _kl_bits(mu, var, logvar).reshape(-1, knum, self.dim).sum(dim=-1)
        kl2_mean, kl2_min, kl2_max = torch.mean(kl2), torch.min(kl2), torch.max(kl2)
        kl_loss = torch.mean(self._weighted_kl(kl2)) * self.lam
        self._update_lambdas(kl2_mean, kl2_min, kl2_max, buggy_lam_max=True)
        info = {"kl_loss": torch.mean(kl_loss), "bits-mean": kl2_mean, "bits-min": kl2_min, "bits-max": kl2_max,
                "lam-min": self.lam_min, "lam-max": self.lam_max, "lam": self.lam,
                "mu": self._restore(mu, z_shape), "std": self._restore(std, z_shape),
                "zhat_noquant": self._restore(zhat, z_shape)}
        return info["zhat_noquant"], info

    def quant_vq(self, z):
        z2, z_shape, mu, _, std = self._split(z.float())
        knum = z2.shape[-1] // (self.dim * 2)
        if self._compat():
            zq, ind = self._compat_rows(mu.reshape(-1, self.dim).contiguous(), std.reshape(-1, self.dim).contiguous(),
                                        self.dim)
            zhat = zq.reshape(-1, knum * self.dim).float()
            indices = ind.reshape(-1, knum)
        else:
            # rows are already "position-major, channel-last": one BLC image of L = #positions
            ind, zq = _lib.gq_quantize_z(z2.contiguous()[None], self.prior_samples, self.dim, "blc",
                                         _lib.GQHIP_GROUP_CONTIGUOUS, self.logvar_range, self.beta, self._ws)
            zhat, indices = zq[0], ind[0]
        zhat = self._restore(zhat, z_shape)
        indices = self._restore(indices, z_shape)
        return zhat, {"indices": indices, "zhat_quant": zhat}

    def _fused_view(self, z):
        """z as (tensor, layout, restore) for the module-level kernels without a copy where its memory allows: a contiguous z is
        [outer, 2C, inner] = the "bchw" layout for any dim_idx; a channels_last 4-d z with dim_idx 1 is the "blc" layout."""
        d = self.dim_idx % z.dim()
        if z.is_contiguous():
            outer = int(math.prod(z.shape[:d]))
            inner = int(math.prod(z.shape[d + 1:]))
            if inner == 1:
                zv = z.reshape(1, outer, z.shape[d])
                return zv, "blc", lambda t: t.reshape(*z.shape[:d], -1, *z.shape[d + 1:])
            zv = z.reshape(outer, z.shape[d], inner)
            return zv, "bchw", lambda t: t.reshape(*z.shape[:d], -1, *z.shape[d + 1:])
        zl = torch.movedim(z, d, -1)
        if not zl.is_contiguous():
            zl = zl.contiguous()
        lead = zl.shape[:-1]
        zv = zl.reshape(1, -1, zl.shape[-1])
        return zv, "blc", lambda t: torch.movedim(t.reshape(*lead, -1), -1, d)

    def _forward_fused(self, z):
        """Eval forward (gaussian.py:333-345) as ONE library call: quant_gaussian's sample, statistics, re-weighted loss and lambda
        update, quant_vq's arg-max and gather, and the straight-through mix all happen in gq_quantize_z_gauss_f32's launches.  The
        lambda state advances on the device: info["lam"], ["lam-min"], ["lam-max"] are 0-d float64 device tensors (float(...) gives
        the reference's Python floats) and nothing is read back unless an attribute is inspected."""
        z = z.float()
        assert z.shape[self.dim_idx] % (self.dim * 2) == 0
        zv, layout, restore = self._fused_view(z)
        c = zv.shape[1 if layout == "bchw" else 2] // 2
        shape_n = (zv.shape[0], c, zv.shape[2]) if layout == "bchw" else (zv.shape[0], zv.shape[1], c)
        noise = torch.randn(shape_n, dtype=torch.float32, device=z.device)     # one draw of the size of mu (gaussian.py:222)
        ind, zhat, zq, noq, std, sc = _lib.gq_quantize_z_gauss(
            zv, self.prior_samples, self.dim, layout, _lib.GQHIP_GROUP_CONTIGUOUS, noise, self._lam_state_on(z.device),
            self.log_n_samples, self.tolerance, self.lam_factor, self.lam_range, lam_max_decreases=False, use_ste=self.use_ste,
            lv_range=self.logvar_range, beta=self.beta, ws=self._ws)
        f32, f64 = sc[:16].view(torch.float32), sc[32:56].view(torch.float64)
        zhat_o = restore(zhat)                    # with use_ste the kernels stored (zhat_g - zhat_g) + zhat_v here
        d = self.dim_idx % z.dim()
        info = {"kl_loss": f32[0], "bits-mean": f32[1], "bits-min": f32[2], "bits-max": f32[3],
                "lam-min": f64[1], "lam-max": f64[2], "lam": f64[0],
                "mu": z.narrow(d, 0, z.shape[d] // 2), "std": restore(std), "zhat_noquant": restore(noq),
                "indices": restore(ind), "zhat_quant": restore(zq)}
        if not self.use_ste and self.training:
            return info["zhat_noquant"], info
        return zhat_o, info

    def forward(self, z):
        if z.is_cuda and not self._compat() and not (torch.is_grad_enabled() and z.requires_grad):
            return self._forward_fused(z)
        zhat_g, info_g = self.quant_gaussian(z)
        with torch.no_grad():
            zhat_v, info_v = self.quant_vq(z)
        if self.use_ste:
            zhat = zhat_g - zhat_g.detach() + zhat_v
        else:
            zhat = zhat_g if self.training else zhat_v
        return zhat, info_g | info_v

    def dequant(self, indices):
        ind = torch.movedim(indices, self.dim_idx, -1)
        i_shape = ind.shape
        flat = ind.reshape(1, -1, i_shape[-1]).contiguous()
        zhat = _lib.gq_dequant(flat, self.prior_samples, self.dim, "blc", _lib.GQHIP_GROUP_CONTIGUOUS)[0]
        return torch.movedim(zhat.reshape(*i_shape[:-1], -1), -1, self.dim_idx)


class IdentityRegularizer(nn.Module):
    def forward(self, z):
        return z, dict()
