"""Per-stage device times (encoder / quantiser / decoder) for the bench workload."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench  # noqa: E402


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        out = fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--channels-last", type=int, default=0)
    ap.add_argument("--fused-gn", type=int, default=1)
    ap.add_argument("--defer-bias", type=int, default=1)
    ap.add_argument("--attn-math", type=int, default=1)
    ap.add_argument("--add-stats", type=int, default=1)
    ap.add_argument("--winograd", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from pit_hip.modules import unet
    unet.FUSED_GN = bool(a.fused_gn)
    unet.DEFER_BIAS = bool(a.defer_bias)
    unet.ATTN_MATH = bool(a.attn_math)
    unet.FUSED_ADD_STATS = bool(a.add_stats)
    unet.WINOGRAD = bool(a.winograd)
    torch.backends.cudnn.benchmark = False
    vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"])
    x = (torch.rand(a.batch, 3, 256, 256) * 2 - 1).to(dev)
    if a.channels_last:
        vae = vae.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(2):
            z = vae.encoder(x)
            zh, info = vae.regularization(z)
            vae.decoder(zh)
        te, z = timed(lambda: vae.encoder(x), a.iters)
        tq, (zh, info) = timed(lambda: vae.regularization(z), a.iters * 4)
        td, _ = timed(lambda: vae.decoder(zh), a.iters)
    print(f"channels_last={a.channels_last} fused_gn={a.fused_gn} defer_bias={a.defer_bias} attn_math={a.attn_math} add_stats={a.add_stats} winograd={a.winograd} batch={a.batch}: encoder {te:.1f} ms, quantiser {tq:.3f} ms, "
          f"decoder {td:.1f} ms -> {a.batch / (te + tq + td) * 1e3:.1f} img/s (stage sum)")


if __name__ == "__main__":
    main()
