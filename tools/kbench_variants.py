"""Timings of the other entry points at BASELINE shapes: compat score op (+argmax), VQ / LFQ at 512x512,
module-level fused quantiser, dequant."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


n, dim = 65536, 16
cb = torch.randn(n, dim, generator=g).clamp(-4.6, 4.6).to(dev)
for rows in (1024, 4096):
    mu = (0.9 * torch.randn(rows, dim, generator=g)).to(dev)
    sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(rows, dim, generator=g))).to(dev)
    out = torch.zeros(rows, n, device=dev)
    t = timed(lambda: _lib.gq_scores(mu, sd, cb, out, 1.0))
    t2 = timed(lambda: torch.argmax(out, dim=1))
    gb = rows * n * 4 / 1e9
    print(f"compat gq_scores rows={rows}: {t:.3f} ms -> {gb / t * 1e3:.0f} GB/s written ({gb / t * 1e3 / 8000 * 100:.1f}% of 8 TB/s); "
          f"torch.argmax re-read {t2:.3f} ms")
rows = 16 * 4096  # config 5: bs 16, 512x512
z = torch.randn(rows, 16, generator=g).to(dev)
emb = torch.randn(n, 16, generator=g).to(dev)
ws = _lib.Workspace()
t = timed(lambda: _lib.vq_argmin(z, emb, ws=ws), 10)
print(f"vq_argmin rows={rows} (bs16 @512^2): {t:.3f} ms -> {2 * 2 * 16 * n * rows / t / 1e9:.1f} TFLOP/s (4*dim*N convention)")
t = timed(lambda: _lib.lfq_pack(z))
print(f"lfq_pack rows={rows}: {t * 1e3:.1f} us -> {(rows * 16 * 4 * 2 + rows * 8) / t / 1e6:.0f} GB/s")
zz = torch.cat([0.9 * torch.randn(16, 16, 32, 32, generator=g), -1.5 + 0.3 * torch.randn(16, 16, 32, 32, generator=g)], 1).to(dev)
t = timed(lambda: _lib.gq_quantize_z(zz, cb, 16, "bchw", 0, ws=ws))
_lib.debug_enable(True)
_lib.gq_quantize_z(zz, cb, 16, "bchw", 0, ws=ws)
torch.cuda.synchronize()
fb, rr = _lib.debug_counters(ws)
_lib.debug_enable(False)
print(f"gq_quantize_z bs16 256^2 (prep + filter + re-rank): {t:.3f} ms; undecided rows (in-block scan) {fb}, candidates/row {rr / 16384:.3f}")
idx, _ = _lib.gq_quantize_z(zz, cb, 16, "bchw", 0, ws=ws)
t = timed(lambda: _lib.gq_dequant(idx, cb, 16, "bchw", 0))
print(f"gq_dequant bs16: {t * 1e3:.1f} us")
