"""Does running a Winograd convolution on sub-batches keep V / M inside the 256 MB Infinity Cache?  Time of one
GroupNorm-fused 3x3 convolution (+ bias + residual + statistics) at 16 x C x H x H as ONE call vs image chunks."""
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
for (C, Co, H) in ((128, 128, 256), (256, 256, 128), (256, 128, 256), (512, 512, 64)):
    conv = torch.nn.Conv2d(C, Co, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
    conv._gq_wino = conv._gq_wino4 = True
    norm = torch.nn.GroupNorm(32, C, eps=1e-6).to(dev)
    x = torch.randn(16, C, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn(16, Co, H, H, device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for f4 in (False, True):
            Uw = U._wino_weights(conv, f4); f16 = U._f16_args_gn(conv, norm, x, f4)
            stats = _lib.gn_stats(x, 32)
            line = f"C{C}->{Co} {H}^2 F{4 if f4 else 2}:"
            for chunk in (16, 8, 4, 2, 1):
                def run():
                    for b0 in range(0, 16, chunk):
                        st = stats[2 * 32 * b0: 2 * 32 * (b0 + chunk)]
                        gn = (norm.weight, norm.bias, 32, 1e-6, True, st, None)
                        _lib.wino_conv3x3(x[b0:b0 + chunk], Uw, gn=gn, residual=res[b0:b0 + chunk], bias=conv.bias,
                                          stats_groups=32, f16=f16)
                line += f"  chunk {chunk}: {timed(run):.0f} us"
            print(line, flush=True)
