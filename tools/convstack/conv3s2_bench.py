"""The stride-2 (Downsample) fp16 x 3 convolution alone: error vs fp64, time vs F.pad + MIOpen's fp32 implicit GEMM."""
import os, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, C, H) in ((2, 128, 64), (16, 128, 256), (16, 256, 128), (16, 512, 64)):
    conv = torch.nn.Conv2d(C, C, 3, 2, 0).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        x = (3 * torch.randn(B, C, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
        wf, us = _lib.conv3s2_weights_f16(conv.weight)
        sc = _lib.f16_scales(_lib.gn_stats(x, 32), 1.0, us)
        y, st = _lib.conv3x3s2_direct(x, wf, us, sc, bias=conv.bias, stats_groups=32)
        sl = slice(0, 2)
        xp = F.pad(x[sl].double(), (0, 1, 0, 1))
        ref = F.conv2d(xp, conv.weight.double(), conv.bias.double(), 2, 0)
        scl = F.conv2d(xp.abs(), conv.weight.double().abs(), None, 2, 0)
        err = float(((y[sl].double() - ref).abs() / scl).max())
        t_own = timed(lambda: _lib.conv3x3s2_direct(x, wf, us, sc, bias=conv.bias, stats_groups=32))
        t_lib = timed(lambda: F.conv2d(F.pad(x, (0, 1, 0, 1)), conv.weight, None, 2, 0))
        fl = 2.0 * B * (H // 2) ** 2 * 9 * C * C * 3
        print(f"B{B} C{C} {H}^2 -> {H//2}^2: own {t_own:.0f} us = {fl/t_own/1e6:.0f} TFLOP/s executed (err {err:.1e}); F.pad + MIOpen fp32 {t_lib:.0f} us", flush=True)
