"""The Winograd GEMMs of the 256- / 512-channel levels alone, at the step's shapes: libgqhip's wino_gemm_f16x2 on the [h | l]
operand vs the library route (ONE hipBLASLt fp16 GEMM over K' = 3 Cin of [h | h | l]); error of both against fp64 on a slice."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

L = _lib.lib()
for name, P, tiles, cin, cout in (("dec L1 F4 256->256", 36, 16384, 256, 256), ("dec L2 F4 512->512", 36, 4096, 512, 512),
                                  ("enc L2 F2 512->512", 16, 16384, 512, 512), ("enc L3 F2 512->512", 16, 4096, 512, 512),
                                  ("dec L3 F4 512->512", 36, 1024, 512, 512), ("dec L1 F4 512->256", 36, 16384, 512, 256),
                                  ("enc L2 F2 256->512", 16, 16384, 256, 512)):
    V = torch.randn(P, tiles, cin, generator=g).to(dev)
    U = (torch.randn(P, cin, cout, generator=g) / 11.3).to(dev)
    vh = V.half(); vl = (V - vh.float()).half()
    uh = U.half(); ul = (U - uh.float()).half()
    V2 = torch.cat([vh, vl], 2).contiguous()
    Wf = _lib.wino_weights_operand_order(uh, ul)
    V3 = torch.cat([vh, vh, vl], 2).contiguous(); U3 = torch.cat([uh, ul, uh], 1).contiguous()
    M = torch.empty(P, tiles, cout, device=dev)
    def own():
        _lib._check(L.wino_gemm_f16x2(V2.data_ptr(), Wf.data_ptr(), M.data_ptr(), P, tiles, cin, cout,
                                      torch.cuda.current_stream().cuda_stream), "wg2")
    t_own = timed(own); t_lib = timed(lambda: torch.bmm(V3, U3, out_dtype=torch.float32))
    own(); ref = torch.bmm(V3, U3, out_dtype=torch.float32)
    r64 = torch.bmm(V[:1, :512].double(), U[:1].double()); sc = torch.bmm(V[:1, :512].abs().double(), U[:1].abs().double())
    e_own = float(((M[:1, :512].double() - r64).abs() / sc).max()); e_lib = float(((ref[:1, :512].double() - r64).abs() / sc).max())
    fl = 6.0 * P * tiles * cin * cout
    gb_own = (V2.numel() * 2 + M.numel() * 4) / 1e9; gb_lib = (V3.numel() * 2 + M.numel() * 4) / 1e9
    print(f"{name:20s} own {t_own*1e3:6.0f} us = {fl/t_own/1e9:5.0f} TFLOP/s executed, {gb_own/t_own:4.2f} TB/s (err {e_own:.1e}, fits {_lib.own_gemm_fits(P, tiles, cout)}) | "
          f"library {t_lib*1e3:6.0f} us = {fl/t_lib/1e9:5.0f} TFLOP/s, {gb_lib/t_lib:4.2f} TB/s (err {e_lib:.1e}) | {t_lib/t_own:4.2f}x", flush=True)
    del V, U, V2, V3, M, ref
