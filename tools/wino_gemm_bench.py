"""The 128-channel Winograd GEMM alone: libgqhip's kernel on the [h | l] operand vs the library route (one fp16 GEMM over
K' = 384), both shapes of the 256 x 256 level; checks the result against fp64 on a slice."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
from pit_hip import _lib
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

for name, P, tiles in (("dec F4 256^2", 36, 65536), ("enc F2 256^2", 16, 262144), ("ragged", 36, 65536 - 77)):
    V = torch.randn(P, tiles, 128, generator=g).to(dev)
    U = (torch.randn(P, 128, 128, generator=g) / 11.3).to(dev)
    vh = V.half(); vl = (V - vh.float()).half()
    uh = U.half(); ul = (U - uh.float()).half()
    V2 = torch.cat([vh, vl], 2).contiguous()
    U2t = torch.stack([uh, ul], 1).transpose(2, 3).contiguous()
    V3 = torch.cat([vh, vh, vl], 2).contiguous(); U3 = torch.cat([uh, ul, uh], 1).contiguous()
    M = torch.empty(P, tiles, 128, device=dev)
    L = _lib.lib()
    def own():
        _lib._check(L.wino_gemm_c128_f16x2(V2.data_ptr(), U2t.data_ptr(), M.data_ptr(), P, tiles, torch.cuda.current_stream().cuda_stream), "wg")
    t_own = timed(own); t_lib = timed(lambda: torch.bmm(V3, U3, out_dtype=torch.float32))
    own(); ref = torch.bmm(V3, U3, out_dtype=torch.float32)
    r64 = torch.bmm(V[:2, :512].double(), U[:2].double()); sc = torch.bmm(V[:2, :512].abs().double(), U[:2].abs().double())
    e_own = float(((M[:2, :512].double() - r64).abs() / sc).max()); e_lib = float(((ref[:2, :512].double() - r64).abs() / sc).max())
    gb = (V2.numel() * 2 + M.numel() * 4) / 1e9
    print(f"{name}: own {t_own*1e3:.0f} us = {gb/t_own:.2f} TB/s (err {e_own:.1e}, max diff vs library {float((M-ref).abs().max()):.1e}); "
          f"library f16x3 {t_lib*1e3:.0f} us (err {e_lib:.1e})", flush=True)
