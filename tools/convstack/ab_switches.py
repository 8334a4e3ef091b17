"""A/B of pit_hip.modules.unet's module-level switches at the bench shape: stage times per variant, and z / indices /
reconstruction of every variant against the FIRST one.
usage: python tools/convstack/ab_switches.py "FUSED_WINO_GN=0,FUSED_WINO_GN_F4=0" "FUSED_WINO_GN=1,FUSED_WINO_GN_F4=1" ..."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vq-vae-from-gaussian-vae_amd"))
import bench
from pit_hip.modules import unet as U
from pit_hip.modules.unet import invalidate_caches
dev = torch.device("cuda:0")
vae = bench.build_model(dev, bench.CONFIGS["gq_0.25"]).to(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(1000)
x = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).to(dev).contiguous(memory_format=torch.channels_last)
def parse(spec):
    out = {}
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("="); cur = getattr(U, k)
        out[k] = (v not in ("0", "False")) if isinstance(cur, bool) else type(cur)(v)
    return out
variants = [parse(a) for a in sys.argv[1:]] or [{}]
defaults = {k: getattr(U, k) for v in variants for k in v}
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
res = []
for rep in range(2):
    for i, v in enumerate(variants):
        for k, val in {**defaults, **v}.items(): setattr(U, k, val)
        invalidate_caches(vae)
        te, td = timed()
        if rep == 0: res.append(run())
        print(f"{v}: encoder {te:.2f} ms, decoder {td:.2f} ms, sum {te + td:.2f} ms -> {16 / (te + td) * 1e3:.1f} img/s", flush=True)
za, ia, ra = res[0]
for v, (zb, ib, rb) in zip(variants[1:], res[1:]):
    print(f"{v} vs first: z max abs diff {float((za - zb).abs().max()):.2e}; indices differing {int((ia != ib).sum())} of "
          f"{ia.numel()}; recon max abs diff {float((ra - rb).abs().max()):.2e}")
