"""Runs one kernel in a loop for ~N seconds (for tools/convstack/power_probe.sh): conv3 (the direct 3x3 convolution at 16 x 256 x 256 x 128 -> 128),
filter (the fused quantiser at config 2), scores (the compat op at 16 384 x 65 536 x dim 16), fill (torch fill_ of 4.29 GB), idle."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
from pit_hip.modules import unet as U
what, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
if what == "conv3":
    conv = torch.nn.Conv2d(128, 128, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
    norm = torch.nn.GroupNorm(32, 128, eps=1e-6).to(dev)
    with torch.no_grad():
        x = torch.randn(16, 128, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
        res = torch.randn(16, 128, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
        wf, us = _lib.conv3_weights_f16(conv.weight)
        stats = _lib.gn_stats(x, 32)
        gn = (norm.weight, norm.bias, 32, 1e-6, True, stats, None)
        bound = U._gn_act_bound(norm, x)
    fn = lambda: _lib.conv3x3_direct(x, wf, us, bound, gn=gn, residual=res, bias=conv.bias, stats_groups=32)
elif what == "filter":
    mu = (0.9 * torch.randn(16384, 16, generator=g)).to(dev)
    sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(16384, 16, generator=g))).to(dev)
    cb = torch.randn(65536, 16, generator=g).clamp(-4.6, 4.6).to(dev)
    ws = _lib.Workspace()
    fn = lambda: _lib.gq_argmax(mu, sd, cb, 1.0, ws=ws)
elif what in ("scores", "fill"):
    mu = (0.9 * torch.randn(16384, 16, generator=g)).to(dev)
    sd = torch.exp(0.5 * (-1.5 + 0.3 * torch.randn(16384, 16, generator=g))).to(dev)
    cb = torch.randn(65536, 16, generator=g).clamp(-4.6, 4.6).to(dev)
    out = torch.empty(16384, 65536, device=dev)
    fn = (lambda: _lib.gq_scores(mu, sd, cb, out, 1.0)) if what == "scores" else (lambda: out.fill_(1.0))
else:
    fn = lambda: time.sleep(0.01)
t0 = time.time()
with torch.no_grad():
    while time.time() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
