"""Which call of the conv stack is not bitwise reproducible?

Wraps every tensor-returning function of pit_hip._lib and the ATen entry points the modules use (conv2d, mm, bmm, matmul,
addmm, softmax), runs encoder / decoder N times on identical inputs and compares, call by call, a checksum of every output
with the first run.  A call is reported as a SOURCE when its outputs differ while all its tensor inputs had the checksums
of the first run.

    python tools/convstack/determinism_trace.py [--size 256] [--batches 1,4,16] [--runs 6]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))

import bench  # noqa: E402
from pit_hip import _lib  # noqa: E402
from pit_hip.modules import unet  # noqa: E402
import torch.nn.functional as F  # noqa: E402

TRACE = []


def _sum(t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.numel() == 0:
        return None
    c = t.detach().contiguous()
    if c.dtype in (torch.float32, torch.int32):
        v = c.view(torch.int32).to(torch.int64)
    elif c.dtype in (torch.float16, torch.bfloat16, torch.int16, torch.uint16):
        v = c.view(torch.int16).to(torch.int64)
    elif c.dtype in (torch.float64, torch.int64):
        v = c.view(torch.int64)
    else:
        v = c.view(torch.uint8).to(torch.int64)
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64).reshape(v.shape) % 8191 + 1
    return int((v * w).sum())


def _flat(x):
    if isinstance(x, torch.Tensor):
        yield x
    elif isinstance(x, (tuple, list)):
        for y in x:
            yield from _flat(y)
    elif isinstance(x, dict):
        for y in x.values():
            yield from _flat(y)


def wrap(owner, name, label):
    fn = getattr(owner, name)

    def w(*a, **k):
        ins = tuple(_sum(t) for t in _flat((a, k)))
        shapes = tuple(tuple(t.shape) for t in _flat((a, k)))
        out = fn(*a, **k)
        outs = tuple(_sum(t) for t in _flat(out))
        TRACE.append((label, shapes, ins, outs))
        return out

    setattr(owner, name, w)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batches", default="1,4,16")
    ap.add_argument("--runs", type=int, default=6)
    ap.add_argument("--config", default="gq_0.25")
    ap.add_argument("--smoke", action="store_true", help="the toy model of __graft_entry__.smoke (resolution 64, one res block)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.smoke:
        from pit_hip.models.autoencoder import AutoencodingEngine

        unet_cfg = dict(attn_type="vanilla", double_z=True, z_channels=16, resolution=64, in_channels=3, out_ch=3, ch=128,
                        ch_mult=[1, 2, 4, 4], num_res_blocks=1, attn_resolutions=[8], dropout=0.0)
        torch.manual_seed(1234)
        vae = AutoencodingEngine(
            encoder_config={"target": "pit.modules.unet.Encoder", "params": unet_cfg},
            decoder_config={"target": "pit.modules.unet.Decoder", "params": unet_cfg},
            regularizer_config={"target": "pit.quantization.gaussian.GaussianQuantRegularizer",
                                "params": {"format": "bchw", "group": 16, "n_samples": 4096, "backend": "hip"}},
        ).eval().to(dev).to(memory_format=torch.channels_last)
    else:
        vae = bench.build_model(dev, bench.CONFIGS[args.config]).to(memory_format=torch.channels_last)
    for n in ("gn_silu", "add_bias", "add_bias_stats", "gn_apply", "gn_stats", "wino_conv3x3", "attention_f16x3",
              "conv3x3_direct", "conv1x1_direct", "conv3x3s2_direct", "upconv2x_direct", "conv3x3_gn_small", "conv3x3_f32",
              "f16_scales", "upsample2x_nhwc"):
        wrap(_lib, n, "_lib." + n)
    wrap(F, "conv2d", "F.conv2d")
    for n in ("mm", "bmm", "matmul", "addmm", "softmax"):
        wrap(torch, n, "torch." + n)
    # nn.Conv2d.forward goes through F.conv2d via _conv_forward (module attribute lookup at call time)
    total_bad = 0
    for B in [int(b) for b in args.batches.split(",")]:
        g = torch.Generator().manual_seed(7 + B)
        x = (torch.rand(B, 3, args.size, args.size, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            z0 = vae.encoder(x)
            zh, _ = vae.regularization(z0)
        for part, fn, inp in (("encoder", vae.encoder, x), ("decoder", vae.decode, zh)):
            runs = []
            outs = []
            with torch.no_grad():
                for _ in range(args.runs):
                    TRACE.clear()
                    o = fn(inp)
                    torch.cuda.synchronize()
                    runs.append(list(TRACE))
                    outs.append(o.clone())
            eq = [torch.equal(outs[0], o) for o in outs[1:]]
            md = max(float((outs[0] - o).abs().max()) for o in outs[1:])
            print(f"B={B} {args.size}^2 {part}: {len(runs[0])} traced calls; output equal to run 0 in {sum(eq)}/{len(eq)} runs, max |diff| {md:.2e}")
            sources = {}
            for r in runs[1:]:
                assert len(r) == len(runs[0])
                for i, (a, b) in enumerate(zip(runs[0], r)):
                    if a[3] != b[3] and a[2] == b[2]:
                        sources[i] = sources.get(i, 0) + 1
            for i, cnt in sorted(sources.items()):
                lab, shapes, _, _ = runs[0][i]
                print(f"   SOURCE call #{i} {lab} shapes {shapes}: differed in {cnt}/{len(runs) - 1} runs with identical inputs")
            total_bad += len(sources)
    print("non-reproducible call sites:", total_bad)


if __name__ == "__main__":
    main()
