"""Which kernel serves which layer at the BASELINE shapes?

Runs encoder + quantiser + decoder of every BASELINE config (gq_0.25 / gq_0.50 / gq_1.00 / gq2_0.25 at 256^2, vq_16 / lfq_16
and gq_0.25 at 512^2; batch 16 and batch 1) with every C-ABI entry point of libgqhip and every library call the modules can
fall back to (F.conv2d -> MIOpen, torch.mm / bmm / matmul / addmm -> hipBLASLt, F.scaled_dot_product_attention, F.group_norm,
F.silu, F.interpolate, F.pad) counted, and prints one table: entry point -> calls per forward at each config.  An entry
point (or fallback) with zero calls everywhere is a route no BASELINE shape selects.

    python tools/route_table.py > profiles/r03/route_table.txt
"""
import collections
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))

import bench  # noqa: E402
from pit_hip import _lib  # noqa: E402

COUNTS = collections.Counter()
SHAPES = collections.defaultdict(set)


class _Proxy:
    def __init__(self, real):
        self._real = real

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if name.startswith("gqhip_debug") or name.startswith("gqhip_profile") or name in ("gqhip_abi_version", "gqhip_workspace_bytes",
                                                                                          "gqhip_status_string", "gqhip_last_hip_error"):
            return fn

        def w(*a):
            COUNTS["libgqhip." + name] += 1
            return fn(*a)

        return w


def wrap(owner, name, label):
    fn = getattr(owner, name)

    def w(*a, **k):
        COUNTS[label] += 1
        ts = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:2]
        SHAPES[label].add(str(ts))
        return fn(*a, **k)

    setattr(owner, name, w)


def main():
    dev = torch.device("cuda:0")
    _lib._lib = _Proxy(_lib.lib())
    wrap(F, "conv2d", "MIOpen: F.conv2d")
    wrap(F, "group_norm", "ATen: F.group_norm")
    wrap(F, "silu", "ATen: F.silu")
    wrap(F, "interpolate", "ATen: F.interpolate")
    wrap(F, "pad", "ATen: F.pad")
    wrap(F, "scaled_dot_product_attention", "ATen: SDPA")
    for n in ("mm", "bmm", "matmul", "addmm"):
        wrap(torch, n, "hipBLASLt: torch." + n)
    cases = [("gq_0.25", 256, 16), ("gq_0.25", 256, 1), ("gq_0.50", 256, 16), ("gq_1.00", 256, 16), ("gq2_0.25", 256, 16),
             ("gq_0.25", 512, 16), ("vq_16", 512, 16), ("lfq_16", 512, 16), ("vq_16", 512, 1)]
    table = {}
    for cfg_name, size, B in cases:
        vae = bench.build_model(dev, bench.CONFIGS[cfg_name]).to(memory_format=torch.channels_last)
        x = (torch.rand(B, 3, size, size) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            zhat, info = vae.encode(x, return_reg_log=True)      # warm-up: caches, workspace
            vae.decode(zhat)
            torch.cuda.synchronize()
            COUNTS.clear()
            zhat, info = vae.encode(x, return_reg_log=True)
            vae.decode(zhat)
            torch.cuda.synchronize()
        table[(cfg_name, size, B)] = dict(COUNTS)
        del vae
    names = sorted({k for t in table.values() for k in t} | {"libgqhip." + n for n in _lib.EXPORTED_SYMBOLS
                                                            if not n.startswith(("gqhip_", "fsq_", "lfq_unpack", "gq_dequant", "gq_scores",
                                                                                 "gq_argmax", "vq_argmin", "gq_index", "gq_indices"))})
    hdr = "  ".join(f"{c}@{s}/b{b}" for c, s, b in cases)
    print(f"{'entry point (calls per encode + decode)':52s} {hdr}")
    for n in names:
        row = "  ".join(f"{table[c].get(n, 0):>{len(f'{c[0]}@{c[1]}/b{c[2]}')}d}" for c in cases)
        tag = "   <-- no BASELINE shape selects it" if all(table[c].get(n, 0) == 0 for c in cases) else ""
        print(f"{n:52s} {row}{tag}")
    print()
    for lab in sorted(SHAPES):
        print(f"{lab}: operand shapes seen: {sorted(SHAPES[lab])[:12]}")


if __name__ == "__main__":
    main()
