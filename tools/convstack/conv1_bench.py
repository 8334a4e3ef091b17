"""The 1x1 fp16 x 3 convolution (libgqhip conv1x1_f16x3) alone: error vs fp64, time vs MIOpen's fp32 implicit GEMM."""
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
for (cin, cout, H) in ((256, 128, 256), (512, 256, 128), (128, 256, 128), (512, 256, 32)):
    conv = torch.nn.Conv2d(cin, cout, 1).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        x = (3 * torch.randn(16, cin, H, H, device=dev)).contiguous(memory_format=torch.channels_last)
        res = torch.randn(16, cout, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        wf, us = _lib.conv3_weights_f16(conv.weight)
        stats = _lib.gn_stats(x, 32)
        sc = _lib.f16_scales(stats, 1.0, us)
        y, st = _lib.conv1x1_direct(x, wf, us, sc, residual=res, bias=conv.bias, stats_groups=32)
        y2 = _lib.conv1x1_direct(x, wf, us, float(x.abs().max()))
        sl = slice(0, 2)
        ref = F.conv2d(x[sl].double(), conv.weight.double(), conv.bias.double()) + res[sl].double()
        scl = F.conv2d(x[sl].double().abs(), conv.weight.double().abs())
        e1 = float(((y[sl].double() - ref).abs() / scl).max())
        e2 = float(((y2[sl].double() - F.conv2d(x[sl].double(), conv.weight.double())).abs() / scl).max())
        t_own = timed(lambda: _lib.conv1x1_direct(x, wf, us, sc))
        t_lib = timed(lambda: F.conv2d(x, conv.weight))
        by = (x.numel() + y.numel()) * 4
        print(f"{cin}->{cout} {H}^2: own {t_own:.0f} us = {by/t_own/1e6:.2f} TB/s, {2*3*x.numel()*cout/t_own/1e6:.0f} TFLOP/s executed "
              f"(err {e1:.1e} device scale, {e2:.1e} host scale); MIOpen fp32 {t_lib:.0f} us", flush=True)
