"""Diagnostic: per-block timeline / placement of conv3x3_gn_f16x3_kernel at 16 x 256 x 256 x 128 -> 128 (needs the stamps build:
`make -C vq-vae-from-gaussian-vae_amd/csrc stamps`, GQHIP_LIB=.../libgqhip_stamps.so).  Prints, per phase (prologue = statistics
fold + first chunk staged, main loop, epilogue), the durations and -- per CU -- how the phases of the two resident blocks overlap:
the share of the kernel's time in which 0 / 1 / 2 of a CU's blocks are inside their main loop."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib  # noqa: E402
from pit_hip.modules import unet as U  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cout, H = 128, 256
conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(dev)
with torch.no_grad():
    x = torch.randn(16, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(16, cout, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    wf, us = _lib.conv3_weights_f16(conv.weight)
    stats = _lib.gn_stats(x, 32)
    gn = (norm.weight, norm.bias, 32, 1e-6, True, stats, None)
    bound = U._gn_act_bound(norm, x)
    for _ in range(5):
        _lib.conv3x3_direct(x, wf, us, bound, gn=gn, residual=res, bias=conv.bias, stats_groups=32)
    torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == "nostamps":     # only the launches (tools/convstack/pmc_conv3_abl.sh)
    sys.exit(0)
nblk = 16 * (H // 8) * (H // 32)
raw = np.zeros(4 * nblk, dtype=np.uint64)
L = _lib.lib()
L.gqhip_debug_c3_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert L.gqhip_debug_c3_stamps(raw.ctypes.data, 4 * nblk) == 0
raw = raw.reshape(nblk, 4)
t0 = raw[:, 0].min()
start = (raw[:, 0] - t0) / 100.0
end = (raw[:, 1] - t0) / 100.0
pro = (raw[:, 2] >> 32) / 100.0
loop_end = (raw[:, 2] & 0xFFFFFFFF) / 100.0
hw = raw[:, 3] & 0xFFFFFFFF
xcc = (raw[:, 3] >> 32) & 0xF
key = (xcc * 10000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)).astype(np.int64)
dur = end - start
print(f"{cin}->{cout} {H}^2: {nblk} blocks; kernel span {end.max():.1f} us; block duration min/med/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us")
print(f"prologue min/med/max {pro.min():.1f}/{np.median(pro):.1f}/{pro.max():.1f} us")
ml = loop_end - pro
print(f"main loop min/med/max {ml.min():.1f}/{np.median(ml):.1f}/{ml.max():.1f} us")
ep = dur - loop_end
print(f"epilogue min/med/max {ep.min():.1f}/{np.median(ep):.1f}/{ep.max():.1f} us")
uniq, cnt = np.unique(key, return_counts=True)
print(f"distinct CUs {len(uniq)}; blocks per CU min/max {cnt.min()}/{cnt.max()}")
# per CU: time with 0 / 1 / 2 blocks inside the main loop, on a 0.5 us grid
grid = np.arange(0.0, end.max(), 0.5)
share = np.zeros(4)
resident = np.zeros(4)
for k in uniq[:64]:
    m = key == k
    inside = np.zeros(len(grid), dtype=int)
    res_n = np.zeros(len(grid), dtype=int)
    for s, p0, l1, e in zip(start[m], pro[m], loop_end[m], end[m]):
        inside += (grid >= s + p0) & (grid < s + l1)
        res_n += (grid >= s) & (grid < e)
    share += np.bincount(np.minimum(inside, 3), minlength=4)
    resident += np.bincount(np.minimum(res_n, 3), minlength=4)
share /= share.sum()
resident /= resident.sum()
print("share of time with 0 / 1 / 2 / 3+ blocks of a CU inside the main loop: " + " / ".join(f"{100 * v:.1f} %" for v in share))
print("share of time with 0 / 1 / 2 / 3+ blocks resident on a CU:             " + " / ".join(f"{100 * v:.1f} %" for v in resident))
first = np.sort(start)[:520]
print(f"start times of the first 512 blocks: max {first[511]:.1f} us; block 513 starts at {np.sort(start)[512]:.1f} us")
