"""Diagnostic (needs `make -C vq-vae-from-gaussian-vae_amd/csrc stamps` and GQHIP_LIB=.../libgqhip_stamps.so): where a wave of the grid
search (csrc/gq_grid.h) spends a row set.  s_memrealtime stamps (100 MHz; a full s_waitcnt in front of each, so the phases do not
overlap as they would in the product build) of wave 0 of six blocks: row set starts | operands + margins | greedy descent + first
leaf | leaf list (LDS) | sub-leaf boxes fetched, sub-list | listed sub-leaves | exact pass | reduce + stores.

    GQHIP_LIB=vq-vae-from-gaussian-vae_amd/csrc/libgqhip_stamps.so python tools/grid_phases.py [--rows 65536 --dim 4] [--flat]
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
ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--dim", type=int, default=4)
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--flat", action="store_true", help="sigma ~ 1 rows (nearly linear scores: the bench's seeded-random encoder)")
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
mu = (0.9 * torch.randn(a.rows, a.dim, generator=g)).to(dev)
sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(a.rows, a.dim, generator=g))).to(dev)
if a.flat:
    mu = (0.6 * torch.randn(a.rows, a.dim, generator=g)).to(dev)
    sd = torch.exp(0.5 * (0.1 * torch.randn(a.rows, a.dim, generator=g))).to(dev)
cb = torch.randn(a.n, a.dim, generator=g).clamp(-4.6, 4.6).to(dev)
ws = _lib.Workspace()
for _ in range(20):
    _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
torch.cuda.synchronize()
raw = ws.buf[1152:1152 + 48 * 8].cpu().numpy().view(np.uint64).reshape(6, 8).astype(np.int64)   # gq_common.h:WsHeader.stamps
names = ["operands, bounds, margins", "greedy descent + first leaf", "leaf list (LDS)", "sub-leaf boxes -> sub-list", "listed sub-leaves",
         "exact pass", "reduce, stores"]
print(f"rows {a.rows} dim {a.dim}: wave 0 of blocks 0, 100, ..., 500, first row set; us per phase")
for b in range(6):
    d = np.diff(raw[b]) / 100.0
    print(f"  block {100 * b:5d}: " + "  ".join(f"{names[k]} {d[k]:.2f}" for k in range(7)) + f"   total {(raw[b, 7] - raw[b, 0]) / 100.0:.2f}")
