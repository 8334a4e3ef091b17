"""The compat op gq_scores_f32 at the size SURVEY.md 8(d) prices: 16 384 rows x 65 536 codes = 4.29 GB of fp32 scores
(VERDICT r2 next #7), beside a plain fill of the same buffer on the same box.

    python tools/scores_bench.py [--dims 16,8,4,32]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dims", default="16")
ap.add_argument("--rows", default="1024,4096,16384")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--accuracy", action="store_true", help="error of the selected kernel against an fp64 evaluation, 512 rows x 8192 codes")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = 65536


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


if a.accuracy:
    for dim in [int(d) for d in a.dims.split(",")]:
        cb = torch.randn(8192, dim, generator=g).clamp(-4.6, 4.6).to(dev)
        mu = (0.9 * torch.randn(512, dim, generator=g)).to(dev)
        sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(512, dim, generator=g))).to(dev)
        out = torch.empty(512, 8192, device=dev)
        _lib.gq_scores(mu, sd, cb, out, 1.0)
        m, s_, c = mu.double()[:, None, :], sd.double()[:, None, :], cb.double()[None, :, :]
        ref = (-((c - m) / s_) ** 2 + c * c).sum(-1)
        terms = (((1.0 - 1.0 / s_ ** 2).abs() * c * c) + (2 * m / s_ ** 2 * c).abs() + (m / s_) ** 2).sum(-1)
        err = (out.double() - ref).abs()
        print(f"accuracy dim {dim} (GQHIP_SCORES={os.environ.get('GQHIP_SCORES', 'default')}): max |out - fp64| = {float(err.max()):.3e}; "
              f"max over elements of |out - fp64| / sum|terms| = {float((err / terms).max()):.3e} (2^-24 = 5.96e-08); "
              f"relative to |score|: {float((err / ref.abs().clamp_min(1e-3)).max()):.3e}")
    sys.exit(0)
for dim in [int(d) for d in a.dims.split(",")]:
    cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6).to(dev)
    for rows in [int(r) for r in a.rows.split(",")]:
        mu = (0.9 * torch.randn(rows, dim, generator=g)).to(dev)
        sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g))).to(dev)
        out = torch.empty(rows, n, device=dev)
        t = timed(lambda: _lib.gq_scores(mu, sd, cb, out, 1.0), a.iters)
        tf = timed(lambda: out.fill_(1.0), a.iters)
        gb = rows * n * 4 / 1e9
        print(f"gq_scores dim {dim} rows {rows} ({gb:.2f} GB): {t:.3f} ms -> {gb / t * 1e3:.0f} GB/s written = {gb / t / 8 * 100:.1f} % of 8 TB/s; "
              f"{4.0 * dim * n * rows / t / 1e9:.1f} TFLOP/s fp32 MFMA; torch fill_ of the same buffer {tf:.3f} ms = {gb / tf * 1e3:.0f} GB/s", flush=True)
        del out
