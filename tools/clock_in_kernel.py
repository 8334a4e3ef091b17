"""Diagnostic (needs `make -C vq-vae-from-gaussian-vae_amd/csrc stamps` and GQHIP_LIB=.../libgqhip_stamps.so): the shader clock
the chip holds INSIDE the filter kernel's main loop = delta s_memtime / delta s_memrealtime x 100 MHz, per block (first 24
blocks), after two seconds of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    _lib.set_filter(sys.argv[2])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
rows, n = 16384, 65536
mu = (0.9 * torch.randn(rows, dim, generator=g)).to(dev)
sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g))).to(dev)
cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6).to(dev)
ws = _lib.Workspace()
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50):
        _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
    torch.cuda.synchronize()
st = ws.buf[1152:1152 + 48 * 8].cpu().numpy().view(np.uint64).reshape(24, 2).astype(np.float64)
clk = st[:, 0] / st[:, 1] * 0.1          # cycles per 10 ns tick -> GHz
print(f"dim {dim} filter {_lib.get_filter()}: plan {_lib.debug_plan(rows, n, dim)}")
print(f"main loop: {np.median(st[:, 1]) / 100.0:.1f} us (median of 24 blocks), {np.median(st[:, 0]):.0f} shader cycles -> in-kernel clock "
      f"min/med/max {clk.min():.2f}/{np.median(clk):.2f}/{clk.max():.2f} GHz")
