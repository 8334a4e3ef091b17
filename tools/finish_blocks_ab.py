"""Diagnostic: wall time per call of the dim-4 quantiser call at the trained operating point and on sigma ~ 1 rows that hand a few
hundred rows to gq_grid_finish_kernel, for GQHIP_FINISH_BLOCKS = the finish kernel's grid (measured, round 5: trained 75-77 us
whatever the grid; flat 193 / 178 / 168 / 167 / 166 us at 64 / 128 / 256 / 512 / 768 blocks: 256 stays).  Batches of ten calls
between synchronisations: a hundred calls (400 launches) enqueued at once -- the host needs 29 us per call, the GPU 170 -- ran
into the HIP runtime's handling of deep queues in every other process (averages of 430 ... 850 us with unchanged kernel times,
one 65 ms host stall), an artefact of the harness, not of the call."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
sys.path.insert(0, ROOT)
from pit_hip import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rows, dim, n = 65536, 4, 65536


def rowsof(kind):
    if kind == "trained":
        mu = 0.9 * torch.randn(rows, dim, generator=g)
        sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g)))
    else:
        mu = 0.3 * torch.randn(rows, dim, generator=g)
        sd = torch.exp(0.5 * (0.25 * torch.randn(rows, dim, generator=g)))
    return mu.to(dev), sd.to(dev)


cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6).to(dev)
for kind in ("trained", "flat"):
    mu, sd = rowsof(kind)
    ws = _lib.Workspace()
    for _ in range(5):
        _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
    torch.cuda.synchronize()
    per = []
    t00 = time.perf_counter()
    for k in range(100):
        t0 = time.perf_counter()
        _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
        per.append(time.perf_counter() - t0)
        if k % 10 == 9:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    w = (time.perf_counter() - t00) / 100
    per = np.array(per) * 1e6
    print(os.environ.get("GQHIP_FINISH_BLOCKS", "256"), kind, f"call {w*1e6:.1f} us; host time per call: median {np.median(per):.1f} p90 "
          f"{np.percentile(per, 90):.1f} max {per.max():.1f} (at call {per.argmax()})", _lib.debug_grid(ws)["scanned_rows"], "rows handed on")
