"""Quantiser call time, undecided rows and candidates per row for three kinds of encoder output (random z,
trained-model-like, tiny z of a random-init encoder) with both filter kernels."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib


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


dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
cb = torch.randn(65536, 16, generator=g).clamp(-4.6, 4.6).to(dev)
for name, z in (("randn z", torch.randn(16, 32, 32, 32, generator=g)),
                ("trained-like", torch.cat([0.9 * torch.randn(16, 16, 32, 32, generator=g), -1.5 + 0.3 * torch.randn(16, 16, 32, 32, generator=g)], 1)),
                ("small z (random-init encoder)", 0.1 * torch.randn(16, 32, 32, 32, generator=g))):
    z = z.to(dev)
    for filt in ("auto", "fp32"):
        _lib.set_filter(filt)
        ws = _lib.Workspace()
        _lib.debug_enable(True)
        _lib.gq_quantize_z(z, cb, 16, "bchw", _lib.GQHIP_GROUP_STRIDED, (-30.0, 20.0), 1.0, ws=ws)
        torch.cuda.synchronize()
        fb, rr = _lib.debug_counters(ws)
        _lib.debug_enable(False)
        t = timed(lambda: _lib.gq_quantize_z(z, cb, 16, "bchw", _lib.GQHIP_GROUP_STRIDED, (-30.0, 20.0), 1.0, ws=ws))
        print(f"{name:32s} filter={filt:5s}: {t:.3f} ms, undecided rows (in-block scan) {fb}, candidates/row {rr/16384:.3f}")
_lib.set_filter("auto")
