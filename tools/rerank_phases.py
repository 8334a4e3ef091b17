"""Diagnostic (needs `make -C vq-vae-from-gaussian-vae_amd/csrc stamps` and GQHIP_LIB=.../libgqhip_stamps.so): where a wave of the
re-rank kernel spends its life.  s_memrealtime stamps (100 MHz) of wave 0 of each block at the phase boundaries of gq_rerank.h:
start | records + sums arrived | bound, margins, candidate lists | pass 1 | pass 2 | block barrier | results stored.

    GQHIP_LIB=vq-vae-from-gaussian-vae_amd/csrc/libgqhip_stamps.so python tools/rerank_phases.py [--rows 16384 --dim 16]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
ap.add_argument("--dim", type=int, default=16)
ap.add_argument("--n", type=int, default=65536)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
mu = (0.9 * torch.randn(a.rows, a.dim, generator=g)).to(dev)
sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(a.rows, a.dim, generator=g))).to(dev)
cb = torch.randn(a.n, a.dim, generator=g).clamp(-4.6, 4.6).to(dev)
ws = _lib.Workspace()
for _ in range(20):
    _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
torch.cuda.synchronize()
pl = _lib.debug_plan(a.rows, a.n, a.dim)
a256 = lambda v: (v + 255) // 256 * 256
off = 4096 + a256(pl["nsplit"] * a.rows * 32) + a256(a.rows * 4) + 65536      # csrc/gqhip.hip:ws_layout: hdr | rec | fb | dbg (2nd half)
nblk = min((a.rows + 15) // 16, 1024)
raw = ws.buf[off:off + nblk * 64].cpu().numpy().view(np.uint64).reshape(nblk, 8).astype(np.int64)
t0 = raw[:, 0].min()
names = ["start (after launch)", "records + sums arrived", "bound, margins, candidates", "pass 1 (gathers, fp32)", "pass 2 (exact) + reduce",
         "block barrier", "results stored"]
print(f"rows {a.rows} dim {a.dim}: {nblk} blocks stamped; 100 MHz ticks -> us")
print(f"  {'phase':32s} {'median':>8s} {'p10':>8s} {'p90':>8s}   (end of phase, us after the first block's start: median)")
prev = raw[:, 0]
for k in range(7):
    cur = raw[:, k]
    d = (cur - (t0 if k == 0 else prev)) / 100.0
    end = (cur - t0) / 100.0
    print(f"  {names[k]:32s} {np.median(d):8.2f} {np.percentile(d, 10):8.2f} {np.percentile(d, 90):8.2f}   {np.median(end):8.2f}")
    prev = cur
print(f"  last block's results stored at {((raw[:, 6] - t0).max()) / 100.0:.2f} us")
