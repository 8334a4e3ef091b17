"""The Upsample layers of the decoder alone (nearest x2 + conv 3x3 as libgqhip's direct sub-pixel convolution): time per call
at the step's three shapes, executed PFLOP/s (4 phases x 4 taps x 3 fp16 products)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
from pit_hip.modules import unet as U
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for ch, H in ((512, 32), (512, 64), (256, 128)):
    up = U.Upsample(ch).to(dev).eval().to(memory_format=torch.channels_last)
    x = torch.randn(16, ch, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    x._gn_stats = (_lib.gn_stats(x, 32), 32)
    with torch.no_grad():
        t = timed(lambda: up(x))
    fl = 4 * 4 * 2.0 * 16 * H * H * ch * ch * 3
    print(f"upsample {ch} ch {H}x{H} -> {2*H}x{2*H}: {t*1e3:7.1f} us, {fl/t/1e12:5.2f} PFLOP/s executed", flush=True)
