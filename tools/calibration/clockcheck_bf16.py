"""Diagnostic: in-kernel shader clock and main-loop duration of the split-bf16 filter (needs
`make -C .../csrc stamps` and GQHIP_LIB=.../libgqhip_stamps.so)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rows, dim, n = 16384, 16, 65536
mu = (0.9 * torch.randn(rows, dim, generator=g)).to(dev)
sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g))).to(dev)
cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6).to(dev)
ws = _lib.Workspace()
for reps in (1, 100):
    for _ in range(reps):
        _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
    torch.cuda.synchronize()
    st = ws.buf[128:128 + 48 * 8].cpu().numpy().view(np.uint64).reshape(24, 2).astype(np.float64)
    ghz = st[:, 0] / st[:, 1] * 0.1
    print(f"after {reps} back-to-back call(s): main loop {np.median(st[:, 1]) / 100:.1f} us (median of 24 blocks), "
          f"shader clock {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f}), "
          f"{np.median(st[:, 0]) / (256 * 12):.1f} clocks per MFMA per wave (2 waves/SIMD)")
