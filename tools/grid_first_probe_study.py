"""CPU study behind the first round trip of csrc/gq_grid.h (round 5): on the rows the bench's gq_1.00 configuration really produces
(seeded-random encoder: sigma ~ 1, nearly linear scores), how long are the leaf / sub-leaf lists under the threshold that each
way of choosing the first leaves gives -- greedy descent by box bound (one leaf), beams of different widths, and the ideal
threshold (the row's true best score).  The kernel's choice for non-concave rows: 2 L1 nodes -> 4 of their 32 L2 nodes -> 2 of
those nodes' 16 leaves.  Output: profiles/r05/grid_first_probe_study.txt."""
import os
import sys

import numpy as np
import torch
from scipy.stats import norm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip.modules.unet import Encoder  # noqa: E402
from pit_hip.quantization.gaussian import prior_samples  # noqa: E402

FULL = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
            num_res_blocks=2, attn_resolutions=[32], dropout=0.0)


def bench_rows():
    """bench.py's encoder (seed 1234) on one of its images (seed 1000), regrouped as gq_1.00 does: dim 4, K = 4 (pit_hip/quantization)."""
    torch.manual_seed(1234)
    enc = Encoder(**FULL).eval()
    g = torch.Generator().manual_seed(1000)
    x = torch.rand(1, 3, 256, 256, generator=g) * 2 - 1
    with torch.no_grad():
        z = enc(x)
    mu, lv = z[:, :16], z[:, 16:].clamp(-30, 20)
    K, dim = 4, 4
    cut = lambda t: t.reshape(1, 16, -1).permute(0, 2, 1).reshape(-1, dim, K).permute(0, 2, 1).reshape(-1, dim).double().numpy()
    return cut(mu), cut(torch.exp(0.5 * lv))


def main(rows=1024):
    m, s = bench_rows()
    A, B = 0.5 - 1 / (2 * s * s), m / (s * s)
    print(f"rows of bench.py --config gq_1.00: sigma quantiles {np.quantile(s, [.01, .5, .99]).round(3)}, A quantiles "
          f"{np.quantile(A, [.01, .5, .99]).round(3)}, |B| median {np.median(np.abs(B)):.3f}; first {rows} rows")
    cb = prior_samples(65536, 4, 42).float().numpy().astype(np.float64)
    th = norm.ppf(np.arange(1, 8) / 8)
    c = np.stack([np.searchsorted(th, cb[:, i], side="right") for i in range(4)], 1)
    l1 = (c[:, 0] >> 2) | ((c[:, 1] >> 2) << 1) | ((c[:, 2] >> 2) << 2) | ((c[:, 3] >> 2) << 3)
    l2 = ((c[:, 0] >> 1) & 1) | (((c[:, 1] >> 1) & 1) << 1) | (((c[:, 2] >> 1) & 1) << 2) | (((c[:, 3] >> 1) & 1) << 3)
    sid = ((l1 * 16 + l2) * 4 + ((c[:, 0] & 1) | ((c[:, 1] & 1) << 1))) * 4 + ((c[:, 2] & 1) | ((c[:, 3] & 1) << 1))   # gq_grid.h:grid_sub_of
    leaf = sid >> 2

    def boxes(ids, n):
        lo, hi = np.full((n, 4), np.inf), np.full((n, 4), -np.inf)
        np.minimum.at(lo, ids, cb)
        np.maximum.at(hi, ids, cb)
        return lo, hi

    def ub(a, b, box):
        lo, hi = box
        v = np.where(a < 0, np.clip(-b / (2 * np.where(a < 0, a, -1)), lo, hi), 0)
        return np.where(a < 0, a * v * v + b * v, np.maximum(a * lo * lo + b * lo, a * hi * hi + b * hi)).sum(1)

    bs, bl, b2, b1 = boxes(sid, 4096), boxes(leaf, 1024), boxes(sid >> 4, 256), boxes(sid >> 8, 16)
    strategies = {"greedy: 1 L1, 1 L2, 1 leaf": (1, 1, 1), "1 L1, 1 L2, 2 leaves": (1, 1, 2), "1 L1, 4 L2, 2 leaves": (1, 4, 2),
                  "2 L1, 4 L2, 2 leaves (kernel)": (2, 4, 2), "2 L1, 4 L2, 4 leaves": (2, 4, 4), "16 L1, 16 L2, 4 leaves": (16, 16, 4),
                  "ideal threshold": None}
    out = {k: [] for k in strategies}
    for r in range(min(rows, A.shape[0])):
        a, b = A[r], B[r]
        f = (a * cb * cb + b * cb).sum(1)
        u1, u2, u3, us = ub(a, b, b1), ub(a, b, b2), ub(a, b, bl), ub(a, b, bs)
        for k, st in strategies.items():
            if st is None:
                Fx = f.max()
            else:
                t1 = np.argsort(-u1)[:st[0]]
                c2 = np.concatenate([np.arange(q * 16, q * 16 + 16) for q in t1])
                t2 = c2[np.argsort(-u2[c2])[:st[1]]]
                c3 = np.concatenate([np.arange(q * 4, q * 4 + 4) for q in t2])
                Fx = f[np.isin(leaf, c3[np.argsort(-u3[c3])[:st[2]]])].max()
            t = Fx - 5e-4
            out[k].append(((u3 >= t).sum(), ((us >= t) & (u3[np.arange(4096) >> 2] >= t)).sum(), Fx == f.max()))
    print(f"{'first round trip':32s} listed leaves: mean  p90  >48 | listed sub-leaves: mean  p90  >96 | the row's winner is among the first codes")
    for k, v in out.items():
        o = np.array(v, dtype=np.float64)
        print(f"{k:32s} {o[:, 0].mean():19.1f} {np.percentile(o[:, 0], 90):4.0f} {(o[:, 0] > 48).mean():5.3f} | "
              f"{o[:, 1].mean():23.1f} {np.percentile(o[:, 1], 90):4.0f} {(o[:, 1] > 96).mean():5.3f} | {o[:, 2].mean():.2f}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
