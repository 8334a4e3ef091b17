"""A/B of the Winograd GEMM route (fp16 x 3 over K vs the library's fp32 GEMM): stage times at the bench shape and the
difference of z / reconstruction between the two."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
from pit_hip.modules import unet as U
dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(1000)
x = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
def run():
    with torch.no_grad():
        z = vae.encoder(x); zh, info = vae.regularization(z); rec = vae.decode(zh)
    return z, info["indices"], rec
def timed(n=10):
    for _ in range(3): run()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    te = td = 0.0
    for _ in range(n):
        with torch.no_grad():
            ev[0].record(); z = vae.encoder(x); ev[1].record(); zh, info = vae.regularization(z); ev[2].record(); rec = vae.decode(zh); ev[3].record()
        torch.cuda.synchronize(); te += ev[0].elapsed_time(ev[1]); td += ev[2].elapsed_time(ev[3])
    return te / n, td / n
res = {}
import itertools
variants = [dict(WINOGRAD_F16X3=False), dict(WINOGRAD_F16X3=True, WINOGRAD_C128_GEMM=False), dict(WINOGRAD_F16X3=True, WINOGRAD_C128_GEMM=True)]
base = dict(WINOGRAD_F16X3=True, FUSED_WINO_GN=False, FUSED_WINO_GN_F4=False, WINOGRAD_C128_GEMM=True)
for rep in range(2):
    for v in variants:
        for k, val in {**base, **v}.items():
            setattr(U, k, val)
        from pit_hip.modules.unet import invalidate_caches
        invalidate_caches(vae)
        te, td = timed()
        res[tuple(sorted(v.items()))] = run()
        print(f"{v}: encoder {te:.2f} ms, decoder {td:.2f} ms, sum {te + td:.2f} ms -> {16 / (te + td) * 1e3:.1f} img/s", flush=True)
za, ia, ra = res[tuple(sorted(variants[0].items()))]; zb, ib, rb = res[tuple(sorted(variants[-1].items()))]
print(f"z max abs diff {float((za - zb).abs().max()):.2e}; indices differing {int((ia != ib).sum())} of {ia.numel()}; "
      f"recon max abs diff {float((ra - rb).abs().max()):.2e}")
