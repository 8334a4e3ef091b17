"""CPU study behind csrc/gq_grid.h (round 5): how many leaves / codes of a 4096-leaf box tree over the 65 536-code book have an upper
bound within the margin of a row's best score -- (a) separable boxes from the normal-quantile cell boundaries, (b) TIGHT bounding
boxes of each leaf's actual members (what the kernel uses) -- at the trained operating point (mu ~ 0.9 N, logvar ~
-1.5 +- 0.3) and in the near-linear regime of seeded-random weights (sigma ~ 1).  Output: profiles/r05/grid_prune_study.txt."""
import os
import sys

import numpy as np
from scipy.stats import norm

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vq-vae-from-gaussian-vae_amd"))
from pit_hip.quantization.gaussian import prior_samples  # noqa: E402


def study(dim, bits_axes, tight, rows=256, seed=0, mu_s=0.9, lv_m=-1.5, lv_s=0.3, margin=1e-3):
    cb = prior_samples(65536, dim, 42).float().numpy().astype(np.float64)
    g = np.random.default_rng(seed)
    mu = mu_s * g.standard_normal((rows, dim))
    sd = np.exp(0.5 * (lv_m + lv_s * g.standard_normal((rows, dim))))
    A, B = 0.5 - 1 / (2 * sd * sd), mu / (sd * sd)
    cell = np.zeros(65536, dtype=np.int64)
    qlo, qhi = [], []
    for i, b in enumerate(bits_axes):
        G = 1 << b
        th = norm.ppf(np.arange(1, G) / G) if G > 1 else np.array([])
        c = np.searchsorted(th, cb[:, i], side="right")
        cell = cell * G + c
        qlo.append(np.concatenate([[-np.inf], th])[c])
        qhi.append(np.concatenate([th, [np.inf]])[c])
    ncell = 1 << sum(bits_axes)
    cnt = np.bincount(cell, minlength=ncell)
    lo, hi = np.full((ncell, dim), np.inf), np.full((ncell, dim), -np.inf)
    if tight:
        np.minimum.at(lo, cell, cb)
        np.maximum.at(hi, cell, cb)
    else:
        amax = np.abs(cb).max()
        np.minimum.at(lo, cell, np.maximum(np.stack(qlo, 1), -amax))
        np.maximum.at(hi, cell, np.minimum(np.stack(qhi, 1), amax))
    F = (A[:, None, :] * cb[None] ** 2 + B[:, None, :] * cb[None]).sum(-1).max(1)
    vc, vk = [], []
    for r in range(rows):
        a, b = A[r], B[r]
        v = np.where(a < 0, np.clip(-b / (2 * np.where(a < 0, a, -1)), lo, hi), 0)
        u = np.where(a < 0, a * v * v + b * v, np.maximum(a * lo * lo + b * lo, a * hi * hi + b * hi)).sum(1)
        m = (u >= F[r] - margin) & (cnt > 0)
        vc.append(m.sum())
        vk.append(cnt[m].sum())
    print(f"dim {dim:2d} leaves {ncell} ({'tight' if tight else 'quantile'} boxes) mu {mu_s} logvar {lv_m}: leaves visited mean "
          f"{np.mean(vc):7.1f} p90 {np.percentile(vc, 90):6.0f} max {np.max(vc):5d}; codes mean {np.mean(vk):7.0f} max {np.max(vk)}; "
          f"largest leaf {cnt.max()}")


if __name__ == "__main__":
    for regime in (dict(), dict(mu_s=0.3, lv_m=0.0, lv_s=0.1)):
        for tight in (False, True):
            study(4, [3] * 4, tight, **regime)
            study(8, [2, 2, 2, 2, 1, 1, 1, 1], tight, **regime)
    study(16, [1] * 12 + [0] * 4, True, rows=64)
    study(16, [1] * 12 + [0] * 4, True, rows=64, mu_s=0.3, lv_m=0.0, lv_s=0.1)
