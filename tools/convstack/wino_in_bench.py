"""Winograd input transforms alone at the step's two biggest shapes: GroupNorm+SiLU apply followed by the plain
transform (two passes) vs the transform with the normalisation fused in (one pass).  Prints time and bytes moved."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

L = _lib.lib()
S = lambda: torch.cuda.current_stream().cuda_stream
for (B, C, H) in ((16, 128, 256), (16, 256, 128)):
    x = torch.randn(B, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev); beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    stats = _lib.gn_stats(x, 32)
    act = x.numel() * 4 / 1e9
    for t in (2, 4):
        P = (t + 2) ** 2; tiles = B * (H // t) ** 2
        for mode, width in (("f16x3", 3), ("f16x2", 2)):
            V = torch.empty(P, tiles, width * C, dtype=torch.float16, device=dev)
            vb = V.numel() * 2 / 1e9
            fn_plain = L.wino_in_nhwc_f16x3 if width == 3 else L.wino_in_nhwc_f16x2
            def two_pass():
                xn = _lib.gn_apply(x, gamma, beta, 32, 1e-6, True, stats)
                _lib._check(fn_plain(xn.data_ptr(), V.data_ptr(), B, H, H, C, t, 64.0, S()), "in")
            def plain_only():
                _lib._check(fn_plain(x.data_ptr(), V.data_ptr(), B, H, H, C, t, 64.0, S()), "in")
            t2 = timed(two_pass); t1 = timed(plain_only)
            line = f"B{B} C{C} {H}^2 F{t} {mode}: apply+transform {t2:.0f} us (transform alone {t1:.0f} us = {(act+vb)/t1*1e3:.2f} TB/s)"
            fused = getattr(L, "wino_in_gn_nhwc_" + mode, None)
            if fused is not None:
                def one_pass():
                    _lib._check(fused(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, stats.data_ptr(), V.data_ptr(),
                                      B, H, H, C, 32, 1e-6, 1, t, 64.0, S()), "in_gn")
                Vref = V.clone(); two_pass(); Vref.copy_(V); one_pass()
                d = float((V.float() - Vref.float()).abs().max())
                t3 = timed(one_pass)
                line += f"; fused {t3:.0f} us = {(act+vb)/t3*1e3:.2f} TB/s (max |dV| {d:.2e})"
            print(line, flush=True)
