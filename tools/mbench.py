"""Module-level microbenchmark: eval forwards of the quantiser MODULES back to back (what `regularization(z)` costs a caller),
for rocprofv3 --kernel-trace --stats breakdowns.  usage: python tools/mbench.py [--module gq|gq2|vq|lfq] [--dim 16] [--bs 16]
[--size 256] [--iters 50] [--nchw]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--module", default="gq2", choices=["gq", "gq2", "vq", "lfq"])
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--nchw", action="store_true", help="NCHW-contiguous z instead of channels_last")
    a = ap.parse_args()
    from pit_hip.quantization.gaussian import GaussianQuantRegularizer, GaussianQuantRegularizer2
    from pit_hip.quantization.lfq import LFQQuantizer
    from pit_hip.quantization.vq import VQQuantizer

    dev = torch.device("cuda:0")
    hw = a.size // 8
    g = torch.Generator().manual_seed(0)
    if a.module in ("gq", "gq2"):
        z = torch.cat([0.9 * torch.randn(a.bs, 16, hw, hw, generator=g), -1.5 + 0.3 * torch.randn(a.bs, 16, hw, hw, generator=g)], 1)
        m = (GaussianQuantRegularizer("bchw", 65536, group=a.dim) if a.module == "gq" else GaussianQuantRegularizer2(a.dim, 65536))
    elif a.module == "vq":
        z = torch.randn(a.bs, 16, hw, hw, generator=g)
        m = VQQuantizer("bchw", 65536, a.dim, codebook_num=16 // a.dim)
        torch.manual_seed(7)
        m.embedding.weight.data.normal_()
    else:
        z = torch.randn(a.bs, 16, hw, hw, generator=g)
        m = LFQQuantizer("bchw", codebook_size=256, num_codebooks=2)
    m = m.eval().to(dev)
    z = z.to(dev)
    if not a.nchw:
        z = z.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(5):
            m(z)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.iters):
            m(z)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters
    print(f"{a.module} dim {a.dim} bs {a.bs} size {a.size} ({'nchw' if a.nchw else 'channels_last'}): "
          f"{e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per forward on the device, {wall * 1e6:.1f} us wall")


if __name__ == "__main__":
    main()
