"""The direct fp16 x 3 3x3 convolution (libgqhip conv3x3_gn_f16x3 / conv3x3_f16x3) alone: correctness against fp64 and against the
Winograd route, time of the split pass and of the convolution kernel at the 256 x 256 level's shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
from pit_hip.modules import unet as U
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
# ---- correctness on a small ragged-ish case (borders, two images, residual, statistics) ----
for cin in (128, 256):
    conv = torch.nn.Conv2d(cin, 128, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
    conv._gq_wino = conv._gq_wino4 = True
    norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(dev)
    with torch.no_grad():
        norm.weight.normal_(); norm.bias.normal_()
        x = (2 * torch.randn(2, cin, 16, 64, device=dev)).contiguous(memory_format=torch.channels_last)
        res = torch.randn(2, 128, 16, 64, device=dev).contiguous(memory_format=torch.channels_last)
        wf, us = _lib.conv3_weights_f16(conv.weight)
        stats = _lib.gn_stats(x, 32)
        gn = (norm.weight, norm.bias, 32, 1e-6, True, stats, None)
        y, st = _lib.conv3x3_direct(x, wf, us, U._gn_act_bound(norm, x), gn=gn, residual=res, bias=conv.bias, stats_groups=32)
        xn = _lib.gn_apply(x, norm.weight, norm.bias, 32, 1e-6, True, stats)
        ref = torch.nn.functional.conv2d(xn.double(), conv.weight.double(), conv.bias.double(), 1, 1) + res.double()
        sc = torch.nn.functional.conv2d(xn.double().abs(), conv.weight.double().abs(), None, 1, 1)
        err = float(((y.double() - ref).abs() / sc).max())
        yw, _ = _lib.wino_conv3x3(x, U._wino_weights(conv, False), gn=gn, residual=res, bias=conv.bias, stats_groups=32,
                                  f16=U._f16_args_gn(conv, norm, x, False))
        errw = float(((yw.double() - ref).abs() / sc).max())
        st_ref = torch.stack([ref.reshape(2, 32, 4, -1).sum((2, 3)), (ref ** 2).reshape(2, 32, 4, -1).sum((2, 3))], -1).flatten()
        print(f"Cin {cin}: direct max err {err:.2e} of sum|x||w| (Winograd F2 route {errw:.2e}); max abs diff {float((y.double()-ref).abs().max()):.2e}; "
              f"stats rel err {float(((_lib.gn_stats_values(st) - st_ref).abs() / st_ref.abs().clamp_min(1)).max()):.1e}", flush=True)
# ---- timing at the bench shapes ----
for (cin, cout, H) in ((128, 128, 256), (256, 128, 256), (256, 256, 128), (128, 256, 128)):
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
    norm = torch.nn.GroupNorm(32, cin, eps=1e-6).to(dev)
    with torch.no_grad():
        x = torch.randn(16, cin, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        res = torch.randn(16, cout, H, H, device=dev).contiguous(memory_format=torch.channels_last)
        wf, us = _lib.conv3_weights_f16(conv.weight)
        stats = _lib.gn_stats(x, 32)
        gn = (norm.weight, norm.bias, 32, 1e-6, True, stats, None)
        bound = U._gn_act_bound(norm, x)
        flops = 2.0 * 16 * H * H * 9 * cin * cout * 3
        t_fused = timed(lambda: _lib.conv3x3_direct(x, wf, us, bound, gn=gn, residual=res, bias=conv.bias, stats_groups=32))
        t_nores = timed(lambda: _lib.conv3x3_direct(x, wf, us, bound, gn=gn, bias=conv.bias, stats_groups=32))
        t_bare = timed(lambda: _lib.conv3x3_direct(x, wf, us, bound, gn=gn))
        print(f"{cin}->{cout} {H}^2: GroupNorm + split + conv + residual + stats {t_fused:.0f} us = {flops/t_fused/1e6:.0f} TFLOP/s executed; "
              f"without residual {t_nores:.0f}; without statistics too {t_bare:.0f}", flush=True)
