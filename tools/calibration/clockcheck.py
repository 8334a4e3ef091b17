"""Diagnostic: per-block timeline / placement of the filter kernel (needs libgqhip_stamps.so via GQHIP_LIB)."""
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
for _ in range(100):
    _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
torch.cuda.synchronize()
L = _lib.lib()
pl = _lib.debug_plan(rows, n, dim)
rpb = 32 * pl['waves'] * pl['rt']
nblk = ((rows + rpb - 1) // rpb) * pl['nsplit']
print('plan:', pl, 'blocks:', nblk)
# workspace layout (csrc/gqhip.hip:ws_layout): hdr (4096) | rec (record sets * rows * 32) | fb (rows * 4) | dbg (64 KiB: the stamps) | ...
a256 = lambda v: (v + 255) // 256 * 256
off = 4096 + a256(pl['nsplit'] * rows * 32) + a256(rows * 4)
raw = ws.buf[off:off + nblk * 32].cpu().numpy().view(np.uint64).reshape(nblk, 4)
t0 = raw[:, 0].min()
start = (raw[:, 0] - t0) / 100.0   # us
end = (raw[:, 1] - t0) / 100.0
pro = (raw[:, 2] >> 32) / 100.0
loop_end = (raw[:, 2] & 0xFFFFFFFF) / 100.0
hw = raw[:, 3] & 0xFFFFFFFF
xcc = (raw[:, 3] >> 32) & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
key = xcc * 10000 + se * 100 + sh * 50 + cu
print(f"kernel span {end.max():.1f} us; block duration min/med/max {np.min(end-start):.1f}/{np.median(end-start):.1f}/{np.max(end-start):.1f} us")
print("start-time histogram (us):", np.histogram(start, bins=[0, 1, 5, 50, 150, 250, 350, 450, 600])[0].tolist())
uniq, cnt = np.unique(key, return_counts=True)
print(f"distinct (xcc,se,sh,cu) = {len(uniq)}; blocks per CU histogram: {np.bincount(cnt).tolist()}")
print("blocks per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
late = start > 50
print(f"late starters: {late.sum()}, their median start {np.median(start[late]) if late.any() else 0:.1f} us")
dur = end - start
print(f"prologue (block start -> first barrier) min/med/max {pro.min():.1f}/{np.median(pro):.1f}/{pro.max():.1f} us")
print(f"main loop min/med/max {np.min(loop_end-pro):.1f}/{np.median(loop_end-pro):.1f}/{np.max(loop_end-pro):.1f} us")
print(f"tail (loop end -> block end) min/med/max {np.min(dur-loop_end):.1f}/{np.median(dur-loop_end):.1f}/{np.max(dur-loop_end):.1f} us")
print(f"last loop end at {np.max(start+loop_end):.1f} us, first block start {start.min():.2f}, last block start {start.max():.2f} us")
