"""Winograd output transforms (+ bias + residual + GroupNorm statistics) and the remaining elementwise passes alone, at
the step's biggest shapes: time and bytes moved."""
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
for (B, C, H) in ((16, 128, 256), (16, 256, 128), (16, 512, 64)):
    res = torch.randn(B, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g).to(dev)
    y = torch.empty_like(res)
    act = res.numel() * 4 / 1e9
    for t in (2, 4):
        P = (t + 2) ** 2; tiles = B * (H // t) ** 2
        M = torch.randn(P, tiles, C, device=dev)
        mb = M.numel() * 4 / 1e9
        stats = torch.empty(2 * B * 32, dtype=torch.float64, device=dev)
        def tail():
            _lib._check(L.wino_out_res_nhwc_f32(M.data_ptr(), res.data_ptr(), bias.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                                B, H, H, C, 32, t, 1.0, S()), "out_res")
        def tail_nores():
            _lib._check(L.wino_out_res_nhwc_f32(M.data_ptr(), None, bias.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                                B, H, H, C, 32, t, 1.0, S()), "out_res")
        fn_plain = L.wino4_out_nhwc_f32 if t == 4 else L.wino_out_nhwc_f32
        def plain():
            _lib._check(fn_plain(M.data_ptr(), y.data_ptr(), B, H, H, C, 1.0, S()), "out")
        t1, t2, t3 = timed(tail), timed(tail_nores), timed(plain)
        print(f"B{B} C{C} {H}^2 F{t}: out+bias+res+stats {t1:.0f} us = {(mb+2*act)/t1*1e3:.2f} TB/s; without residual {t2:.0f} us = "
              f"{(mb+act)/t2*1e3:.2f} TB/s; plain {t3:.0f} us = {(mb+act)/t3*1e3:.2f} TB/s", flush=True)
    x = res
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    st = _lib.gn_stats(x, 32)
    t_stats = timed(lambda: _lib.gn_stats(x, 32)); t_apply = timed(lambda: _lib.gn_apply(x, gamma, beta, 32, 1e-6, True, st))
    t_add = timed(lambda: _lib.add_bias_stats(x, y, bias, 32))
    print(f"B{B} C{C} {H}^2: gn_stats {t_stats:.0f} us = {act/t_stats*1e3:.2f} TB/s; gn_apply {t_apply:.0f} us = {2*act/t_apply*1e3:.2f} TB/s; "
          f"add_bias_stats {t_add:.0f} us = {3*act/t_add*1e3:.2f} TB/s", flush=True)
